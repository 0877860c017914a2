"""In-memory perturbation dataset and a device-resident batch source.

``TensorPerturbationDataset`` / ``SCRNASeqTensorPerturbationDataset`` mirror the reference's
classes (cradle_vae/data/utils/perturbation_dataset.py:44-89).  ``DeviceBatchLoader`` replaces
the ``torch.utils.data.DataLoader`` the reference puts in front of them
(norman/data_module.py train_dataloader): the whole dataset lives in HBM (a 4 M-cell x 8000-gene
fp32 matrix is 128 GB, X + cfX need bf16 or two GPUs; Norman's 111 k x 5000 is 2.2 GB), every
batch is one on-device row gather of a shuffled index slice, and with N ranks each rank
walks its own disjoint slice of the same permutation.  No host->device copy is left in the
training loop.
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, Optional

import numpy as np
import torch


class TensorPerturbationDataset:
    """perturbation_dataset.py:44-82."""

    def __init__(self, X: torch.Tensor, D: torch.Tensor, qc: torch.Tensor, cfX: Optional[torch.Tensor] = None,
                 ids: Optional[Iterable] = None):
        self.X, self.D, self.qc, self.cfX = X, D, qc, cfX
        self.ids = np.arange(len(X)) if ids is None else np.array(ids)
        self.D_obs_per_dim = (self.D != 0).sum(0)
        self.qc_obs_per_dim = (self.qc != 0).sum(0)
        self.library_size = self.X.sum(1)

    def __getitem__(self, idx) -> Dict[str, torch.Tensor]:
        return dict(idx=idx, X=self.X[idx], D=self.D[idx], qc=self.qc[idx], cfX=self.cfX[idx])

    def __len__(self):
        return len(self.X)

    def get_dosage_obs_per_dim(self):
        return self.D_obs_per_dim

    def get_qc_obs_per_dim(self):
        return self.qc_obs_per_dim

    def convert_idx_to_ids(self, idx: np.ndarray) -> np.ndarray:
        return self.ids[idx]


class SCRNASeqTensorPerturbationDataset(TensorPerturbationDataset):
    """perturbation_dataset.py:85-89: adds ``library_size`` to every sample."""

    def __getitem__(self, idx):
        out = super().__getitem__(idx)
        out["library_size"] = self.library_size[idx]
        return out


class DeviceBatchLoader:
    """Iterates ``{idx, X, D, qc, cfX, library_size}`` batches gathered on ``device``.

    shuffle: a fresh permutation per epoch from ``seed + epoch`` -- the same on every rank, so
    the ranks' slices (``perm[rank::world_size]``, truncated to a common length) are disjoint.
    """

    def __init__(self, dataset: TensorPerturbationDataset, batch_size: int, shuffle: bool = False,
                 drop_last: bool = False, seed: int = 0, rank: int = 0, world_size: int = 1, device=None):
        self.dataset = dataset
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.seed, self.rank, self.world_size = seed, rank, world_size
        self.device = torch.device(device) if device is not None else dataset.X.device
        self.epoch = 0
        f32 = dict(device=self.device, dtype=torch.float32)
        self._t = dict(X=dataset.X.to(**f32), D=dataset.D.to(**f32), qc=dataset.qc.to(**f32),
                       cfX=dataset.cfX.to(**f32), library_size=dataset.library_size.to(**f32))

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def _indices(self) -> torch.Tensor:
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            perm = torch.randperm(n, generator=g)
        else:
            perm = torch.arange(n)
        if self.world_size > 1:
            per_rank = n // self.world_size if self.shuffle else (n + self.world_size - 1) // self.world_size
            perm = perm[self.rank::self.world_size][:per_rank]
        return perm

    def __len__(self):
        n = len(self._indices())
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        idx_all = self._indices()
        idx_dev = idx_all.to(self.device)
        n = len(idx_all)
        stop = n - n % self.batch_size if self.drop_last else n
        for s in range(0, stop, self.batch_size):
            sl = idx_dev[s:s + self.batch_size]
            batch = {k: v.index_select(0, sl) for k, v in self._t.items()}
            batch["idx"] = idx_all[s:s + self.batch_size]
            yield batch
        if self.shuffle:
            self.epoch += 1
