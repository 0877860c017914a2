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


class SparseDosage:
    """A dosage matrix D (n_cells, n_treatments) kept as its non-zeros: per cell ``kmax`` (treatment index,
    dosage) pairs, index -1 = unused slot, indices ascending.  The reference keeps D dense
    (perturbation_dataset.py:47-67), which is 160 GB for 4 M cells x 10 k treatments; every cell carries one or
    two guides (data/norman/data_module.py:100-102).  Rows are densified only for the batch that needs them."""

    def __init__(self, idx: torch.Tensor, val: torch.Tensor, n_treatments: int):
        assert idx.shape == val.shape and idx.dim() == 2
        self.idx, self.val, self.n_treatments = idx.to(torch.int32), val.to(torch.float32), int(n_treatments)

    @classmethod
    def from_dense(cls, D: torch.Tensor, kmax: Optional[int] = None, chunk: int = 1 << 16) -> "SparseDosage":
        n, T = D.shape
        if kmax is None:
            kmax = 1
            for i in range(0, n, chunk):
                kmax = max(kmax, int((D[i:i + chunk] != 0).sum(1).max().item())) if n else 1
        idx = torch.full((n, kmax), -1, dtype=torch.int32, device=D.device)
        val = torch.zeros((n, kmax), dtype=torch.float32, device=D.device)
        for i in range(0, n, chunk):
            blk = D[i:i + chunk]
            nz = blk != 0
            if int(nz.sum(1).max().item()) > kmax:
                raise ValueError(f"a cell has more than kmax={kmax} non-zero dosages")
            # stable sort puts the non-zero columns first, in ascending column order
            order = torch.sort((~nz).to(torch.int8), dim=1, stable=True)[1][:, :kmax]
            v = torch.gather(blk, 1, order).float()
            keep = v != 0
            idx[i:i + chunk] = torch.where(keep, order.to(torch.int32), torch.full_like(order, -1, dtype=torch.int32))
            val[i:i + chunk] = v
        return cls(idx, val, T)

    @property
    def shape(self):
        return (self.idx.shape[0], self.n_treatments)

    @property
    def device(self):
        return self.idx.device

    def __len__(self):
        return self.idx.shape[0]

    def to(self, device=None, **_):
        return SparseDosage(self.idx.to(device), self.val.to(device), self.n_treatments)

    def rows(self, sel: torch.Tensor) -> "SparseDosage":
        return SparseDosage(self.idx.index_select(0, sel), self.val.index_select(0, sel), self.n_treatments)

    def dense(self, sel: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Dense fp32 rows (all, or those of the index tensor ``sel``) -- the batch contract's D (N, T)."""
        idx, val = (self.idx, self.val) if sel is None else (self.idx.index_select(0, sel), self.val.index_select(0, sel))
        out = torch.zeros((idx.shape[0], self.n_treatments + 1), dtype=torch.float32, device=idx.device)
        col = torch.where(idx >= 0, idx, torch.full_like(idx, self.n_treatments)).long()  # unused slots -> a spare column
        out.scatter_(1, col, val)
        return out[:, :self.n_treatments].contiguous()

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            return self.dense(torch.tensor([int(i)], device=self.device))[0]
        if isinstance(i, slice):
            return SparseDosage(self.idx[i], self.val[i], self.n_treatments).dense()
        return self.dense(torch.as_tensor(i, device=self.device).long())

    def obs_per_dim(self) -> torch.Tensor:
        """(D != 0).sum(0) (perturbation_dataset.py:64)."""
        if self.idx.is_cuda:
            from . import ops

            return ops.obs_counts_pairs(self.idx, self.val, self.n_treatments)
        hit = (self.idx >= 0) & (self.val != 0)
        return torch.bincount(self.idx[hit].long(), minlength=self.n_treatments)

    def max_per_row(self) -> int:
        return max(1, int(((self.idx >= 0) & (self.val != 0)).sum(1).max().item())) if len(self) else 1


def pack_counts(X: torch.Tensor, dtype="auto") -> torch.Tensor:
    """Count matrices are integer-valued (the reference validates that: pyro GammaPoisson's support): stored as
    uint8 / int16 / int32 they take a quarter / half of the fp32 footprint and convert back exactly."""
    if dtype is None or dtype == "fp32" or X.dtype in (torch.uint8, torch.int16, torch.int32):
        return X
    if dtype == "auto":
        if not bool((X == X.round()).all()) or bool((X < 0).any()):
            return X
        mx = float(X.max()) if X.numel() else 0.0
        dtype = torch.uint8 if mx <= 255 else (torch.int16 if mx <= 32767 else (torch.int32 if mx < 2**31 else None))
        if dtype is None:
            return X
    return X.to(dtype)


def _obs_counts(M: torch.Tensor) -> torch.Tensor:
    """(M != 0).sum(0) (perturbation_dataset.py:64-65): a HIP reduction for a matrix in HBM, torch for one on the host."""
    if M.is_cuda and M.dtype == torch.float32:
        from . import ops

        return ops.obs_counts_dense(M)
    return (M != 0).sum(0)


class TensorPerturbationDataset:
    """perturbation_dataset.py:44-82.  ``D`` may be dense (as in the reference) or a ``SparseDosage``."""

    def __init__(self, X: torch.Tensor, D, qc: torch.Tensor, cfX: Optional[torch.Tensor] = None,
                 ids: Optional[Iterable] = None):
        self.X, self.D, self.qc, self.cfX = X, D, qc, cfX
        self.ids = np.arange(len(X)) if ids is None else np.array(ids)
        self.D_obs_per_dim = self.D.obs_per_dim() if isinstance(D, SparseDosage) else _obs_counts(self.D)
        self.qc_obs_per_dim = _obs_counts(self.qc)
        self.library_size = self.X.sum(1, dtype=torch.float32) if not self.X.is_floating_point() else self.X.sum(1)

    def __getitem__(self, idx) -> Dict[str, torch.Tensor]:
        return dict(idx=idx, X=self.X[idx].float(), D=self.D[idx], qc=self.qc[idx], cfX=self.cfX[idx].float())

    def __len__(self):
        return len(self.X)

    def get_dosage_obs_per_dim(self):
        return self.D_obs_per_dim

    def max_dosages_per_cell(self) -> int:
        """Largest number of non-zero dosages any cell of this data set carries (Norman: 2, Replogle: 1).
        The fused step sizes its per-cell dosage lists and its compact treatment axis with it."""
        if getattr(self, "_max_dosages", None) is None and isinstance(self.D, SparseDosage):
            self._max_dosages = self.D.max_per_row()
        if getattr(self, "_max_dosages", None) is None:
            m = 0
            for i in range(0, self.D.shape[0], 1 << 16):
                m = max(m, int((self.D[i:i + (1 << 16)] != 0).sum(1).max().item()))
            self._max_dosages = max(m, 1)
        return self._max_dosages

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

    Storage: counts (``X``, ``cfX``) as the narrowest integer type that holds them (``count_dtype="auto"``;
    converted to fp32 by the batch gather, exactly), ``D`` as ``SparseDosage`` pairs (densified per batch: the
    batch contract's dense (N, T)), and -- with ``shard_storage`` (default for more than one rank) -- only this
    rank's cells: rank r of w keeps cells r, r + w, r + 2w, ... of a seeded permutation and reshuffles within
    them every epoch, so that 4 M cells x 8 k genes (64 GB per count matrix at 16 bit) cost every one of 8 ranks
    16 GB rather than 256 GB.  Without ``shard_storage`` every rank holds the whole set and walks its slice of a
    fresh global permutation per epoch (``perm[rank::world_size]``, truncated to a common length: disjoint).
    """

    def __init__(self, dataset: TensorPerturbationDataset, batch_size: int, shuffle: bool = False,
                 drop_last: bool = False, seed: int = 0, rank: int = 0, world_size: int = 1, device=None,
                 shard_storage: Optional[bool] = None, count_dtype="auto", sparse_dosage: bool = True,
                 prefetch: bool = True):
        self.dataset = dataset
        self.prefetch, self._side = prefetch, None
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.seed, self.rank, self.world_size = seed, rank, world_size
        self.device = torch.device(device) if device is not None else dataset.X.device
        self.epoch = 0
        self.shard_storage = (world_size > 1) if shard_storage is None else bool(shard_storage)
        n = len(dataset)
        # cells this rank stores (global indices, ascending within the rank's share of a seeded permutation)
        if self.shard_storage and world_size > 1:
            g = torch.Generator().manual_seed(seed)
            perm = torch.randperm(n, generator=g) if shuffle else torch.arange(n)
            per_rank = n // world_size if shuffle else (n + world_size - 1) // world_size
            self.owned = perm[rank::world_size][:per_rank].sort()[0] if shuffle else perm[rank::world_size][:per_rank]
        else:
            self.owned = None
        sel = None if self.owned is None else self.owned.to(dataset.X.device)

        def take(t):
            return t if sel is None else t.index_select(0, sel.to(t.device))

        f32 = dict(device=self.device, dtype=torch.float32)
        D = dataset.D
        if isinstance(D, SparseDosage):
            Dst = (D if sel is None else D.rows(sel.to(D.device))).to(self.device)
        elif sparse_dosage:
            Dst = SparseDosage.from_dense(take(D)).to(self.device)
        else:
            Dst = take(D).to(**f32)
        self._t = dict(X=pack_counts(take(dataset.X), count_dtype).to(self.device), D=Dst,
                       qc=take(dataset.qc).to(**f32), cfX=pack_counts(take(dataset.cfX), count_dtype).to(self.device),
                       library_size=take(dataset.library_size).to(**f32))

    _COUNT_TYPES = {torch.float32: 0, torch.uint8: 1, torch.int16: 2, torch.int32: 3}

    @property
    def _hip_ok(self) -> bool:
        t = self._t
        T = t["D"].n_treatments if isinstance(t["D"], SparseDosage) else t["D"].shape[1]
        return (t["X"].dtype in self._COUNT_TYPES and t["cfX"].dtype in self._COUNT_TYPES and T <= 60 * 1024 * 8
                and all(v.is_contiguous() for v in (t["X"], t["cfX"], t["qc"], t["library_size"]))
                and (isinstance(t["D"], SparseDosage) or (t["D"].is_contiguous() and t["D"].dtype == torch.float32))
                and t["qc"].dtype == torch.float32 and t["library_size"].dtype == torch.float32)

    def _gather_hip(self, sl: torch.Tensor) -> Dict[str, torch.Tensor]:
        """The batch of stored rows ``sl`` through ``cradle_gather_batch`` (csrc/batch.hip)."""
        import ctypes

        from . import hip

        t, dev = self._t, self.device
        N, G, Q = sl.numel(), t["X"].shape[1], t["qc"].shape[1]
        sparse = isinstance(t["D"], SparseDosage)
        T = t["D"].n_treatments if sparse else t["D"].shape[1]
        f32 = dict(device=dev, dtype=torch.float32)
        out = dict(X=torch.empty((N, G), **f32), D=torch.empty((N, T), **f32), qc=torch.empty((N, Q), **f32),
                   cfX=torch.empty((N, G), **f32), library_size=torch.empty((N,), **f32),
                   treatments_present=torch.empty((2,), device=dev, dtype=torch.int32))  # [count, out-of-range flag]
        d = hip.BatchDesc()
        d.N, d.G, d.T, d.Q = N, G, T, Q
        d.x_type, d.cfx_type = self._COUNT_TYPES[t["X"].dtype], self._COUNT_TYPES[t["cfX"].dtype]
        d.pos, d.X, d.cfX = sl.data_ptr(), t["X"].data_ptr(), t["cfX"].data_ptr()
        d.qc, d.library_size = t["qc"].data_ptr(), t["library_size"].data_ptr()
        if sparse:
            d.kmax, d.d_idx, d.d_val = t["D"].idx.shape[1], t["D"].idx.data_ptr(), t["D"].val.data_ptr()
        else:
            d.d_dense = t["D"].data_ptr()
        d.X_out, d.cfX_out, d.qc_out = out["X"].data_ptr(), out["cfX"].data_ptr(), out["qc"].data_ptr()
        d.lib_out, d.D_out, d.present_out = (out["library_size"].data_ptr(), out["D"].data_ptr(),
                                             out["treatments_present"].data_ptr())
        d.rows, d.flags = t["X"].shape[0], out["treatments_present"].data_ptr() + 4
        hip.check(hip.load().cradle_gather_batch(ctypes.byref(d), hip.stream()), "cradle_gather_batch")
        return out

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def stored_bytes(self) -> int:
        """Bytes this rank keeps resident for the data set."""
        tot = 0
        for v in self._t.values():
            tot += (v.idx.numel() * 4 + v.val.numel() * 4) if isinstance(v, SparseDosage) else v.numel() * v.element_size()
        return tot

    def _indices(self):
        """-> (positions into this rank's storage, global cell indices), in iteration order."""
        if self.owned is not None:
            m = len(self.owned)
            if self.shuffle:
                g = torch.Generator().manual_seed(self.seed + 1000003 * (self.epoch + 1) + self.rank)
                pos = torch.randperm(m, generator=g)
            else:
                pos = torch.arange(m)
            return pos, self.owned[pos]
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            perm = torch.randperm(n, generator=g)
        else:
            perm = torch.arange(n)
        if self.world_size > 1:
            per_rank = n // self.world_size if self.shuffle else (n + self.world_size - 1) // self.world_size
            perm = perm[self.rank::self.world_size][:per_rank]
        return perm, perm

    def __len__(self):
        n = len(self._indices()[0])
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        pos_all, idx_all = self._indices()
        pos_dev = pos_all.to(self.device)
        n = len(pos_all)
        stop = n - n % self.batch_size if self.drop_last else n

        def gather(s):
            sl = pos_dev[s:s + self.batch_size]
            if self.device.type == "cuda" and self._hip_ok:
                batch = self._gather_hip(sl)  # K0: one launch for the five tensors, one for the count
                batch["idx"] = idx_all[s:s + self.batch_size]
                return batch
            batch = {}
            for k, v in self._t.items():
                if isinstance(v, SparseDosage):
                    batch[k] = v.dense(sl)
                else:
                    r = v.index_select(0, sl)
                    batch[k] = r if r.dtype == torch.float32 else r.to(torch.float32)
            batch["idx"] = idx_all[s:s + self.batch_size]
            # distinct treatments in the batch: lets the fused step size its treatment axis by what is there
            # (cradle_step_desc.max_present) instead of by N * kmax; a device scalar here, a host int at hand-over
            batch["treatments_present"] = (batch["D"] != 0).any(dim=0).sum(dtype=torch.int32).reshape(1)
            return batch

        def resolve(batch, pinned=None):
            words = (batch["treatments_present"] if pinned is None else pinned).tolist()
            if len(words) > 1 and words[1]:
                raise IndexError("DeviceBatchLoader: a batch position lies outside the stored data set")
            batch["treatments_present"] = int(words[0])
            return batch

        starts = list(range(0, stop, self.batch_size))
        if self.device.type != "cuda" or not self.prefetch or not starts:
            for s in starts:
                yield resolve(gather(s))
        else:
            # the gathers of batch i+1 (index_select, integer -> fp32, densified D: ~0.1 ms of small kernels) run on a
            # side stream while the consumer's stream works on batch i; the hand-over is one event per batch
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            side = self._side
            side.wait_stream(torch.cuda.current_stream(self.device))  # storage and pos_dev were made there

            if getattr(self, "_pinned", None) is None:  # two batches are in flight at most: a ring of four host words
                self._pinned = [torch.empty(2, dtype=torch.int32, pin_memory=True) for _ in range(4)]
            ring = [0]

            def launch(s):
                with torch.cuda.stream(side):
                    b = gather(s)
                    pinned = self._pinned[ring[0] % 4]
                    ring[0] += 1
                    words = b["treatments_present"]
                    pinned[:words.numel()].copy_(words, non_blocking=True)
                    if words.numel() < 2:
                        pinned[1] = 0
                    ev = torch.cuda.Event()
                    ev.record(side)
                return b, ev, pinned

            nxt = launch(starts[0])
            for i in range(len(starts)):
                batch, ev, pinned = nxt
                nxt = launch(starts[i + 1]) if i + 1 < len(starts) else None
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for t in batch.values():
                    if t.is_cuda:
                        t.record_stream(cur)  # allocated on the side stream, read on the consumer's
                ev.synchronize()  # (enqueued one batch ago: done long before the consumer gets here)
                yield resolve(batch, pinned)
        if self.shuffle:
            self.epoch += 1


def _on_device(X: torch.Tensor) -> bool:
    """A count matrix resident in HBM in one of the loader's storage types goes through the HIP kernels
    (csrc/dataprep.hip); a matrix on the host is handled on the host, as the reference's data modules do."""
    from . import ops

    return X.is_cuda and X.dtype in ops.COUNT_TYPES and X.dim() == 2 and X.stride(1) == 1


def batch_log_mean(X: torch.Tensor, step_size: int = 1 << 16) -> torch.Tensor:
    """mean over cells of log(X + 1) per gene (cradle_vae/data/utils/batch_statistics.py:4-10), in row
    slabs so that a data set resident in HBM needs no second copy."""
    if _on_device(X):
        from . import ops

        return (ops.log_stats(X)[0] / X.shape[0]).float()
    total = torch.zeros(X.shape[1], device=X.device, dtype=torch.float32)
    for i in range(0, X.shape[0], step_size):
        total += torch.log(X[i:i + step_size].float() + 1).sum(0)
    return total / X.shape[0]


def batch_log_std(X: torch.Tensor, log_mean: torch.Tensor, correction: int = 1, step_size: int = 1 << 16) -> torch.Tensor:
    """std over cells of log(X + 1) per gene around ``log_mean`` (batch_statistics.py:13-23)."""
    if _on_device(X):
        from . import ops

        centred = ops.log_stats(X, shift=log_mean.to(device=X.device, dtype=torch.float32).contiguous())[1]
        return torch.sqrt(centred / (X.shape[0] - correction)).float()
    total = torch.zeros(X.shape[1], device=X.device, dtype=torch.float32)
    for i in range(0, X.shape[0], step_size):
        total += ((torch.log(X[i:i + step_size].float() + 1) - log_mean) ** 2).sum(0)
    return torch.sqrt(total / (X.shape[0] - correction))


def log_mean_std(X: torch.Tensor, correction: int = 1):
    """(batch_log_mean, batch_log_std) of a count matrix in HBM in ONE streaming pass (``cradle_log_stats``): sums of
    y - y0 and (y - y0)^2, y = log(x + 1), around the first row's y0 -- the shift keeps the variance free of
    cancellation (a gene that is constant over all cells gets std 0 exactly, as the reference's two passes give it)."""
    from . import ops

    n = X.shape[0]
    shift = torch.log(X[0].float() + 1)
    s = ops.log_stats(X, shift=shift)
    mean = shift.double() + s[0] / n
    var = torch.clamp(s[1] - s[0] * s[0] / n, min=0.0) / (n - correction)
    return mean.float(), torch.sqrt(var).float()


# ---------------------------------------------------------------------------------------------
# Counterfactual inputs and an in-memory data module
# ---------------------------------------------------------------------------------------------
def build_counterfactual_inputs(X: torch.Tensor, treatment_codes: torch.Tensor, qc_fail: torch.Tensor,
                                n_peers: int = 50, generator: Optional[torch.Generator] = None,
                                peer_index: Optional[Dict[int, torch.Tensor]] = None,
                                max_rows_per_gather: int = 1 << 22,
                                reference_source: Optional[torch.Tensor] = None) -> torch.Tensor:
    """cfX of one split (reference ``_get_cfX``, cradle_vae/data/norman/data_module.py:175-191): for every
    cell, the per-gene (lower) median over ``min(n_peers, #fail)`` QC-failing cells of the same treatment,
    drawn with replacement; zeros when the treatment has no QC-failing cell.  Runs on X's device, one
    gather + median per treatment instead of a pandas ``apply`` over cells.

    ``peer_index`` ({treatment code: (n_cells_of_t, k) positions into that treatment's fail list}) injects
    the draws (tests); otherwise they come from ``generator``.

    ``reference_source``: the reference groups the split's QC-failing cells with ``groupby("treatment").indices``
    -- positions within that *filtered* frame -- and then indexes the un-split count matrix of the whole data set
    with them (:177-190: ``X`` is the full matrix there).  The rows whose median it takes are therefore
    ``X_all[j]`` for ``j`` = the peer's rank among the split's QC-failing cells, not the peer's own counts.  Passing
    the whole data set's matrix here reproduces exactly that (pinned by tests/golden/cfx_tiny.npz, the output of
    the reference's own function); without it the peers' own rows are used -- what the procedure means to do.
    """
    dev = X.device
    n, G = X.shape
    codes = treatment_codes.to(dev)
    fail = qc_fail.to(dev).bool()
    source = X if reference_source is None else reference_source.to(dev)
    rank_among_fails = torch.cumsum(fail.long(), 0) - 1  # position of a failing cell within the split's failing cells
    if _on_device(X) and source.dtype == X.dtype and source.is_contiguous() and n_peers <= 128:
        # HBM-resident counts: the peers of every cell as one (n, n_peers) table of source rows (-1 = unused), then ONE
        # launch of the segmented per-gene median kernel for the whole split (cradle_peer_median)
        from . import ops

        peers = torch.full((n, max(1, min(n_peers, 128))), -1, device=dev, dtype=torch.int32)
        for t in torch.unique(codes).tolist():
            members = torch.nonzero(codes == t).squeeze(1)
            fails = torch.nonzero((codes == t) & fail).squeeze(1)
            if fails.numel() == 0:
                continue
            k = min(n_peers, fails.numel())
            if peer_index is not None:
                pos = peer_index[t].to(dev)
            else:
                pos = torch.randint(0, fails.numel(), (members.numel(), k), device=dev, generator=generator)
            sel = fails[pos]
            if reference_source is not None:
                sel = rank_among_fails[sel]
            peers[members, :k] = sel.to(torch.int32)
        return ops.peer_median(source, peers)
    cfX = torch.zeros((n, G), device=dev, dtype=X.dtype)
    for t in torch.unique(codes).tolist():
        members = torch.nonzero(codes == t).squeeze(1)
        fails = torch.nonzero((codes == t) & fail).squeeze(1)
        if fails.numel() == 0:
            continue
        k = min(n_peers, fails.numel())
        if peer_index is not None:
            pos = peer_index[t].to(dev)
        else:
            pos = torch.randint(0, fails.numel(), (members.numel(), k), device=dev, generator=generator)
        rows_per = max(1, max_rows_per_gather // (k * G))
        for s in range(0, members.numel(), rows_per):
            sel = fails[pos[s:s + rows_per]]  # (m, k) rows of the split
            if reference_source is not None:
                sel = rank_among_fails[sel]  # the reference's positions, read from the un-split matrix
            cfX[members[s:s + rows_per]] = source[sel].median(dim=1)[0].to(cfX.dtype)
    return cfX


class TensorPerturbationDataModule:
    """In-memory data module with the interface the training / evaluation code uses
    (cradle_vae/data/utils/perturbation_datamodule.py:18-158; concrete behaviour as
    norman/data_module.py:120-286): splits, obs counts, cfX, train-set statistics, unique-intervention
    table and dosage map, data-side ATE.  Everything stays on ``device``."""

    control_label = "non-targeting"

    def __init__(self, X: torch.Tensor, D: torch.Tensor, qc: torch.Tensor, treatments: Iterable[str], splits: Iterable[str],
                 x_var_names: Optional[Iterable[str]] = None, d_var_names: Optional[Iterable[str]] = None,
                 qc_var_names: Optional[Iterable[str]] = None, batch_size: int = 512, device=None, seed: int = 0,
                 control_label: Optional[str] = None, rank: int = 0, world_size: int = 1,
                 cf_indexing: str = "reference"):
        """``cf_indexing="reference"`` builds cfX exactly as the reference's ``_get_cfX`` does, peers' rows read from
        the un-split matrix by their rank among the split's QC-failing cells (see build_counterfactual_inputs);
        ``"same_treatment"`` takes the median over the peers' own counts."""
        import pandas as pd

        self.device = torch.device(device) if device is not None else X.device
        self.batch_size, self.seed, self.rank, self.world_size = batch_size, seed, rank, world_size
        if control_label is not None:
            self.control_label = control_label
        f32 = dict(device=self.device, dtype=torch.float32)
        self.X, self.qc = X.to(**f32), qc.to(**f32)
        # dosages stay (index, dosage) pairs from here on: dense only per batch (DeviceBatchLoader)
        self.D = (D if isinstance(D, SparseDosage) else SparseDosage.from_dense(D.to(self.device))).to(self.device)
        n = self.X.shape[0]
        self.obs = pd.DataFrame({"treatment": np.asarray(list(treatments)), "split": np.asarray(list(splits))},
                                index=[f"cell{i}" for i in range(n)])
        self.obs["total_qc"] = (self.qc.max(dim=1)[0] != 0).cpu().numpy().astype(int)
        self.x_var_info = pd.DataFrame(index=list(x_var_names) if x_var_names is not None else [f"gene{i}" for i in range(self.X.shape[1])])
        self.d_var_info = pd.DataFrame(index=list(d_var_names) if d_var_names is not None else [f"guide{i}" for i in range(self.D.shape[1])])  # noqa: E501
        self.qc_var_info = pd.DataFrame(index=list(qc_var_names) if qc_var_names is not None else [f"qc{i}" for i in range(self.qc.shape[1])])
        self.n_qc_fail = int(self.obs["total_qc"].sum())
        self.n_qc_pass = n - self.n_qc_fail
        codes = torch.from_numpy(pd.Categorical(self.obs["treatment"]).codes.astype(np.int64))
        gen = torch.Generator(device=self.device).manual_seed(seed)
        ids = np.asarray(self.obs.index)
        self.datasets = {}
        for split in ("train", "val", "test"):
            idx = torch.from_numpy(np.nonzero((self.obs["split"] == split).to_numpy())[0])
            di = idx.to(self.device)
            Xs = self.X[di]
            cfX = build_counterfactual_inputs(Xs, codes[idx], self.qc[di].max(dim=1)[0] != 0, generator=gen,
                                              reference_source=self.X if cf_indexing == "reference" else None)
            self.datasets[split] = SCRNASeqTensorPerturbationDataset(Xs, self.D.rows(di), self.qc[di], cfX, ids=ids[idx.numpy()])
        Xtr = self.datasets["train"].X
        if _on_device(Xtr):
            log_mean, log_std = log_mean_std(Xtr)  # one pass over the train set
        else:
            log_mean = batch_log_mean(Xtr)
            log_std = batch_log_std(Xtr, log_mean)
        self.x_train_statistics = _stats_cls()(x_mean=Xtr.mean(0), x_std=Xtr.std(0), log_x_mean=log_mean, log_x_std=log_std)
        df = self.obs.groupby("treatment")["split"].agg(set).reset_index()
        for split in ("train", "val", "test"):
            df[split] = df["split"].apply(lambda s, split=split: split in s)
        self.unique_observed_intervention_df = df.set_index("treatment").drop(columns=["split"])
        first = self.obs.assign(i=np.arange(n)).drop_duplicates("treatment").set_index("treatment")["i"].to_dict()
        self.unique_intervention_dosage_map = {k: self.D[v].cpu() for k, v in first.items()}

    # ---- loaders
    def _loader(self, split, shuffle):
        return DeviceBatchLoader(self.datasets[split], self.batch_size, shuffle=shuffle, drop_last=shuffle, seed=self.seed,
                                 rank=self.rank, world_size=self.world_size, device=self.device)

    def train_dataloader(self):
        return self._loader("train", True)

    def val_dataloader(self):
        return self._loader("val", False)

    def test_dataloader(self):
        return self._loader("test", False)

    # ---- counts / info (perturbation_datamodule.py:19-135)
    def get_train_perturbation_obs_counts(self):
        return self.datasets["train"].get_dosage_obs_per_dim()

    def get_val_perturbation_obs_counts(self):
        return self.datasets["val"].get_dosage_obs_per_dim()

    def get_test_perturbation_obs_counts(self):
        return self.datasets["test"].get_dosage_obs_per_dim()

    def get_train_qc_obs_counts(self):
        return self.datasets["train"].get_qc_obs_per_dim()

    def get_val_qc_obs_counts(self):
        return self.datasets["val"].get_qc_obs_per_dim()

    def get_test_qc_obs_counts(self):
        return self.datasets["test"].get_qc_obs_per_dim()

    def get_x_var_info(self):
        return self.x_var_info.copy()

    def get_d_var_info(self):
        return self.d_var_info.copy()

    def get_qc_var_info(self):
        return self.qc_var_info.copy()

    def get_qc_n(self):
        return {"n_qc_pass": self.n_qc_pass, "n_qc_fail": self.n_qc_fail}

    def get_obs_info(self):
        return self.obs.copy()

    def get_x_train_statistics(self):
        return self.x_train_statistics

    def get_unique_observed_intervention_info(self):
        return self.unique_observed_intervention_df.copy()

    def get_unique_observed_intervention_dosages(self, pert_names) -> torch.Tensor:
        D = torch.zeros((len(pert_names), self.d_var_info.shape[0]))
        for i, name in enumerate(pert_names):
            D[i] = self.unique_intervention_dosage_map[name]
        return D

    def get_estimated_average_treatment_effects(self, method, split: Optional[str] = None, qc_pass: bool = False):
        from .analysis.average_treatment_effects import estimate_data_average_treatment_effects
        from .predictor import make_anndata

        keep = np.ones(len(self.obs), dtype=bool)
        if qc_pass:
            keep &= (self.obs["total_qc"] == 0).to_numpy()
        if split is not None:
            keep &= (self.obs["split"] == split).to_numpy()
        adata = make_anndata(self.X.cpu().numpy()[keep], obs=self.obs[keep], var=self.x_var_info.copy())
        return estimate_data_average_treatment_effects(adata, label_col="treatment", control_label=self.control_label,
                                                       method=method)

    def get_simulated_latent_effects(self):
        return None


def _stats_cls():
    from .guide import ObservationNormalizationStatistics

    return ObservationNormalizationStatistics


class SyntheticPerturbationDataModule(TensorPerturbationDataModule):
    """Synthetic Perturb-seq-like data of a given shape (no data set can be fetched here): a low-rank
    ground-truth effect per guide on top of gene base rates, Poisson counts, ~20 % QC-failing cells with
    damped counts, control + single + (optionally) double perturbations, 80/10/10 split by cell."""

    def __init__(self, n_cells: int = 4096, n_genes: int = 200, n_guides: int = 10, frac_combinations: float = 0.2,
                 batch_size: int = 256, device=None, seed: int = 0, n_qc_pass: int = 0, n_qc_fail: int = 0, **kw):
        g = torch.Generator().manual_seed(seed)
        base = torch.exp(-1.0 + 1.2 * torch.randn(n_genes, generator=g))
        base = base * (1e4 * n_genes / 5000) / base.sum() * 1.0
        effect = 0.6 * torch.randn(n_guides, 4, generator=g) @ torch.randn(4, n_genes, generator=g) / 2.0
        kind = torch.rand(n_cells, generator=g)
        D = torch.zeros(n_cells, n_guides)
        a = torch.randint(0, n_guides, (n_cells,), generator=g)
        b = torch.randint(0, n_guides, (n_cells,), generator=g)
        single = kind >= 0.15
        D[single, a[single]] = 1.0
        double = (kind >= 1.0 - frac_combinations) & (a != b)
        D[double, b[double]] = 1.0
        names = []
        for i in range(n_cells):
            nz = torch.nonzero(D[i]).squeeze(1).tolist()
            names.append(self.control_label if not nz else "+".join(f"guide{j}" for j in nz))
        qc = (torch.rand(n_cells, 1, generator=g) < 0.2).float()
        lib = torch.exp(0.3 * torch.randn(n_cells, 1, generator=g))
        rate = base * torch.exp(D @ effect) * lib * (1.0 - 0.5 * qc)
        X = torch.poisson(rate, generator=g)
        X[:, 0] += 1
        u = torch.rand(n_cells, generator=g)
        splits = np.where(u.numpy() < 0.8, "train", np.where(u.numpy() < 0.9, "val", "test"))
        super().__init__(X, D, qc, names, splits, batch_size=batch_size, device=device, seed=seed, **kw)
        self.true_effect = effect
