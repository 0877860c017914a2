"""Thin tensor-level wrappers over the C ABI (one function per entry point).

These take/return torch CUDA tensors purely as device-memory handles; all arithmetic is in
the HIP kernels.  Autograd wiring lives in ``cradle_vae_amd.functional``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import hip


def nb_gene_consts(log_concentration: torch.Tensor) -> torch.Tensor:
    """(G,) log-concentration -> (3,G) [c, lgamma(c), digamma(c)]  (mlp.py:160)."""
    G = log_concentration.numel()
    out = torch.empty((3, G), device=log_concentration.device, dtype=torch.float32)
    hip.check(hip.load().cradle_nb_gene_consts(hip.ptr(log_concentration), hip.ptr(out), G, hip.stream()), "nb_gene_consts")
    return out


def _nb_check(logits, X, lib, P, N, G):
    if logits.shape != (P * N, G) or X.shape != (N, G) or lib.numel() != N:
        raise hip.CradleHipError(
            f"nb_logprob: shape mismatch logits {tuple(logits.shape)} X {tuple(X.shape)} lib {tuple(lib.shape)} "
            f"for P={P} N={N} G={G}"
        )


def nb_logprob_fwd(logits: torch.Tensor, X: torch.Tensor, lib: torch.Tensor, gene_consts: torch.Tensor, P: int) -> torch.Tensor:
    """sum_g log GammaPoisson(X | softmax(logits), lib, c) per (particle, cell) row -> (P*N,)."""
    N, G = X.shape
    _nb_check(logits, X, lib, P, N, G)
    out = torch.empty((P * N,), device=X.device, dtype=torch.float32)
    hip.check(
        hip.load().cradle_nb_logprob_fwd(
            hip.ptr(logits), logits.stride(0), hip.ptr(X), X.stride(0), hip.ptr(lib), hip.ptr(gene_consts),
            hip.ptr(out), P, N, G, hip.stream()),
        "nb_logprob_fwd",
    )
    return out


def nb_logprob_fwd_bwd(
    logits: torch.Tensor, X: torch.Tensor, lib: torch.Tensor, gene_consts: torch.Tensor, grow: torch.Tensor, P: int,
    dlog_c: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None, inplace: bool = False,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (out (P*N,), dlogits (P*N,G), dlog_concentration (G,)); ``inplace`` overwrites logits."""
    N, G = X.shape
    _nb_check(logits, X, lib, P, N, G)
    lib_ = hip.load()
    out = torch.empty((P * N,), device=X.device, dtype=torch.float32)
    dlogits = logits if inplace else torch.empty_like(logits)
    accumulate = dlog_c is not None
    if dlog_c is None:
        dlog_c = torch.empty((G,), device=X.device, dtype=torch.float32)
    need = lib_.cradle_nb_workspace_floats(P, N, G)
    if workspace is None:
        workspace = torch.empty((need,), device=X.device, dtype=torch.float32)
    hip.check(
        lib_.cradle_nb_logprob_fwd_bwd(
            hip.ptr(logits), logits.stride(0), hip.ptr(dlogits), dlogits.stride(0), hip.ptr(X), X.stride(0),
            hip.ptr(lib), hip.ptr(gene_consts), hip.ptr(grow), hip.ptr(out), hip.ptr(dlog_c), int(accumulate),
            hip.ptr(workspace), workspace.numel(), P, N, G, hip.stream()),
        "nb_logprob_fwd_bwd",
    )
    return out, dlogits, dlog_c
