"""Autograd wiring of the HIP kernels (``torch.autograd.Function`` per kernel group).

Used by the module-level API (``guide.forward`` / ``model.forward`` / predictor), where the
caller composes the pieces itself.  The training step ``loss_module.loss`` does not go
through autograd at all: it calls the fused C step (``cradle_vae_amd.step``).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

LEAKY_SLOPE = 0.01


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) (+ residual) on the fp32 MFMA GEMM (mlp.py:76-80,111-119)."""

    @staticmethod
    def forward(ctx, x, W, b, act, residual):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, N = x2.shape[0], W.shape[0]
        res2 = residual.reshape(M, N) if residual is not None else None
        act_out = torch.empty((M, N), device=x.device, dtype=torch.float32) if (act and residual is not None) else None
        y = ops.gemm(ops.NT, x2, W, bias=b, act=act, act_out=act_out, residual=res2)
        ctx.act = act
        ctx.has_res = residual is not None
        ctx.has_bias = b is not None
        ctx.save_for_backward(x2, W, act_out if act_out is not None else (y if act else None))
        ctx.x_shape = x.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W, r = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        da = dy2 * torch.where(r > 0, 1.0, LEAKY_SLOPE) if ctx.act else dy2
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(ops.NN, da, W).reshape(ctx.x_shape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            db_buf = torch.empty((W.shape[0],), device=dy.device, dtype=torch.float32) if ctx.has_bias else None
            dW = ops.gemm(ops.TN, da, x2, extra_col=db_buf)
            db = db_buf
        dres = dy if ctx.has_res else None
        return dx, dW, db, None, dres


def linear(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], act: bool = False,
           residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    return _Linear.apply(x, W, b, act, residual)


class _NBLogProb(torch.autograd.Function):
    """sum_g log GammaPoisson(X | softmax(logits), library, exp(log_c)) per (particle, cell)."""

    @staticmethod
    def forward(ctx, logits, X, lib, log_c, P):
        gc = ops.nb_gene_consts(log_c)
        out = ops.nb_logprob_fwd(logits, X, lib, gc, P)
        ctx.save_for_backward(logits, X, lib, gc)
        ctx.P = P
        return out

    @staticmethod
    def backward(ctx, grow):
        logits, X, lib, gc = ctx.saved_tensors
        _, dlogits, dlog_c = ops.nb_logprob_fwd_bwd(logits, X, lib, gc, grow.contiguous(), ctx.P)
        return dlogits, None, None, dlog_c, None


def nb_logprob(logits: torch.Tensor, X: torch.Tensor, lib: torch.Tensor, log_c: torch.Tensor, P: int) -> torch.Tensor:
    """logits (P*N, G) -> (P*N,) summed log-likelihood (loss_modules.py:134-135,143)."""
    return _NBLogProb.apply(logits.contiguous(), X.contiguous(), lib.contiguous(), log_c.contiguous(), P)
