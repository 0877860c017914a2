"""Host-side glue of the multi-GPU path (one process per GPU, ``torch.distributed``; the
"nccl" backend is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference trains with Lightning's DDP strategy (train_norman.py:89-97): every rank
computes the ELBO of its own cells and the gradients are averaged.  The path shards by
cells (training) and by particle chunks (ATE estimation); the only data-path collective is
one all-reduce of the flat gradient arena (or of the ATE sums).
"""
from __future__ import annotations

from typing import List, Tuple

import torch

PLATED_KEYS = ("u_mask", "eps_E", "eps_qc_E")  # perturbation-plated draws: global, must agree across ranks
LOCAL_KEYS = ("eps_z",)  # per-cell draws: must differ across ranks


def world_and_rank() -> Tuple[int, int]:
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def make_generators(seed: int, rank: int, device) -> Tuple[torch.Generator, torch.Generator]:
    """(plated, local): the plated generator is seeded identically on every rank so that all
    ranks draw the same mask / E / qc_E noise (they are shared latent variables whose
    gradients are averaged); the local generator is rank-specific."""
    gen_plated = torch.Generator(device=device).manual_seed(seed)
    gen_local = torch.Generator(device=device).manual_seed(seed + 7919 * (rank + 1))
    return gen_plated, gen_local


def draw_noise(gen_plated, gen_local, P: int, N: int, T: int, L: int, Q: int, device) -> dict:
    return dict(
        u_mask=torch.rand((P, T, L), device=device, generator=gen_plated),
        eps_E=torch.randn((P, T, L), device=device, generator=gen_plated),
        eps_qc_E=torch.randn((P, Q, L), device=device, generator=gen_plated),
        eps_z=torch.randn((P, N, L), device=device, generator=gen_local),
    )


def allreduce_sum_(flat: torch.Tensor, world_size: int) -> torch.Tensor:
    """In-place sum of ``flat`` over ranks (no-op for one rank).  The mean is taken by the
    consumer (``cradle_adam_step``'s grad_scale = 1/world_size) so that no extra pass is needed."""
    if world_size > 1:
        _all_reduce(flat)
    return flat


def _all_reduce(t: torch.Tensor, op=None) -> None:
    """RCCL reduces device tensors in place; with the gloo backend (rehearsals of the N>1 path on a
    box with fewer GPUs than ranks, CPU tests) device tensors are staged through host memory."""
    dist = torch.distributed
    kw = {} if op is None else {"op": op}
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, **kw)
        t.copy_(h)
    else:
        dist.all_reduce(t, **kw)


def allreduce_mean_grads_(params: List[torch.nn.Parameter], world_size: int) -> None:
    """Averages ``p.grad`` over ranks through one flat bucket (modular path)."""
    if world_size == 1:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    _all_reduce(flat)
    flat.div_(world_size)
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p))
        off += n


def my_chunks(n_chunks: int, world_size: int, rank: int) -> List[int]:
    """Round-robin deal of independent work items (ATE particle chunks) to ranks."""
    return [i for i in range(n_chunks) if i % world_size == rank]
