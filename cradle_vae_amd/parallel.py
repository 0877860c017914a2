"""Host-side glue of the multi-GPU path (one process per GPU, ``torch.distributed``; the
"nccl" backend is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference trains with Lightning's DDP strategy (train_norman.py:89-97): every rank
computes the ELBO of its own cells and the gradients are averaged.  The path shards by
cells (training) and by particle chunks (ATE estimation); the only data-path collective is
one all-reduce of the flat gradient arena (or of the ATE sums).
"""
from __future__ import annotations

import os
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


def group_size() -> int:
    dist = torch.distributed
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def reduce_scatter_inplace_(bucket: torch.Tensor, rank: int, world: int) -> None:
    """Sum of ``bucket`` (a contiguous 1-D view, length divisible by ``world``) over ranks, delivered into this rank's
    piece ``bucket[rank * n/world : (rank + 1) * n/world]`` -- RCCL's in-place reduce-scatter.  gloo has no
    reduce-scatter: there the whole bucket is all-reduced through host memory (rehearsals and tests; same sums)."""
    dist = torch.distributed
    if world == 1 and not (dist.is_initialized() and dist.get_backend() == "nccl"):
        return  # (a one-rank RCCL group still makes the call: tests exercise the library's stream ordering with it)
    n = bucket.numel() // world
    if dist.get_backend() == "gloo":
        _all_reduce(bucket)
        return
    dist.reduce_scatter_tensor(bucket[rank * n:(rank + 1) * n], bucket)


def all_gather_inplace_(bucket: torch.Tensor, rank: int, world: int) -> None:
    """Every rank's piece of ``bucket`` (as in reduce_scatter_inplace_) to all ranks, in place."""
    dist = torch.distributed
    if world == 1 and not (dist.is_initialized() and dist.get_backend() == "nccl"):
        return
    n = bucket.numel() // world
    if dist.get_backend() == "gloo":
        mine = bucket[rank * n:(rank + 1) * n].detach().cpu()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        bucket.copy_(torch.cat(parts))
        return
    dist.all_gather_into_tensor(bucket, bucket[rank * n:(rank + 1) * n])


_COALESCE = {"ok": os.environ.get("CRADLE_COALESCE_ALLGATHER", "1") != "0"}


def all_gather_inplace_many_(buckets: List[torch.Tensor], rank: int, world: int) -> None:
    """all_gather_inplace_ of several buckets as ONE RCCL group launch (ncclGroupStart/End through c10d's coalescing
    manager: a collective on this path costs tens of microseconds of launch and handshake whatever its size, and the
    parameter all-gather of the step is four of them, exposed after the optimiser).  Falls back to one call per
    bucket when the backend or this torch build has no coalesced all-gather."""
    dist = torch.distributed
    if _COALESCE["ok"] and len(buckets) > 1 and dist.is_initialized() and dist.get_backend() == "nccl" and \
            hasattr(dist, "_coalescing_manager"):
        try:
            with dist._coalescing_manager():
                for b in buckets:
                    n = b.numel() // world
                    dist.all_gather_into_tensor(b, b[rank * n:(rank + 1) * n])
            return
        except (RuntimeError, ValueError, AssertionError, AttributeError, TypeError) as e:  # not supported here
            _COALESCE["ok"] = False
            import warnings

            warnings.warn(f"coalesced all-gather unavailable ({type(e).__name__}: {e}); using one call per bucket")
            dist.distributed_c10d._world.pg_coalesce_state.pop(dist.distributed_c10d._get_default_group(), None)
    for b in buckets:
        all_gather_inplace_(b, rank, world)


def all_gather_small(mine: torch.Tensor, world: int) -> torch.Tensor:
    """(k,) per rank -> (world * k,), rank-major."""
    dist = torch.distributed
    if world == 1:
        return mine.clone()
    if dist.get_backend() == "gloo":
        h = mine.detach().cpu()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h)
        return torch.cat(parts).to(mine.device)
    out = torch.empty(world * mine.numel(), device=mine.device, dtype=mine.dtype)
    dist.all_gather_into_tensor(out, mine)
    return out


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
