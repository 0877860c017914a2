"""Model-based average treatment effects -- mirror of the reference's
``estimate_model_average_treatment_effect`` (cradle_vae/analysis/average_treatment_effects.py:78-203).

The reference walks the alternative perturbations one by one, each with its own decoder call
and count draw (:178-195).  Here the control and every alternative perturbation of a particle
chunk go through ONE decoder GEMM chain, and one launch of the sampler kernel
(``cradle_nb_sample_accumulate``, K9) draws the counts, applies the perturb-seq
normalisation and adds them to the per-perturbation sums without materialising the draws.

Particle chunks are independent given the seed, so with ``torch.distributed`` initialised
they are dealt round-robin to the ranks and the sums meet in one all-reduce.
"""
from __future__ import annotations

import contextlib
import os

from typing import Dict, List, Literal, Optional, Sequence

import numpy as np
import torch

from . import functional as F_hip
from . import ops, parallel

MAX_ROWS_PER_LAUNCH = 1 << 16  # decoder rows (particles x conditions) per GEMM chain / sampler launch ...
MAX_LOGIT_FLOATS = 1 << 27     # ... and at most 512 MB of decoder logits at a time


_DRAW_STREAMS = {}


def _draw_stream(device) -> "torch.cuda.Stream":
    """One side stream per device for the sampler launches of every call (created once: a stream is a hardware queue)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _DRAW_STREAMS:
        _DRAW_STREAMS[key] = torch.cuda.Stream(device=device)
    return _DRAW_STREAMS[key]


def _chunk_generator(device, seed: int, chunk_idx: int) -> torch.Generator:
    g = torch.Generator(device=device)
    g.manual_seed((seed * 1000003 + chunk_idx) & (2**63 - 1))
    return g


@torch.no_grad()
def estimate_model_average_treatment_effect(
    model: torch.nn.Module,
    guide: torch.nn.Module,
    dosages_alt: torch.Tensor,
    dosages_control: torch.Tensor,
    quality: torch.Tensor,
    n_particles: int,
    method: Literal["mean", "perturbseq"],
    condition_values: Optional[Dict[str, torch.Tensor]] = None,
    batch_size: int = 500,
    dosage_independent_variables: Optional[List[str]] = None,
    seed: int = 0,
    *,
    sampler: Literal["nb", "mean"] = "nb",
    noise_chunks: Optional[Sequence[Dict[str, torch.Tensor]]] = None,
) -> np.ndarray:
    """-> (n_alt, n_phenos) numpy array: mean over particles of (transformed) x_good under
    each alternative dosage minus the same under the control dosage.

    ``sampler="mean"`` replaces the count draw by the distribution mean and ``noise_chunks``
    (per chunk: u_mask, eps_E, eps_qc_E, z_control (c,1,L), z_alt (n_alt,c,1,L)) injects the
    latent draws: the deterministic form the golden fixture pins against the reference.
    """
    assert method in ("mean", "perturbseq"), "Method must be one of ['mean', 'perturbseq']"  # :94-95
    if model.sum_mode:
        raise NotImplementedError("sum_mode=True decoders are not on the HIP path (no shipped config uses them)")
    cv = dict(condition_values) if condition_values is not None else {}
    device = next(model.parameters()).device
    D_alt = dosages_alt.to(device=device, dtype=torch.float32)
    D_ctl = dosages_control.to(device=device, dtype=torch.float32)
    if D_ctl.shape[0] != 1:
        raise ValueError("dosages_control must have exactly one row")
    if "library_size" not in cv:
        raise KeyError("condition_values['library_size'] is required by the library_nb decoder (model.py:128)")
    lib = cv["library_size"].to(device=device, dtype=torch.float32).reshape(-1).contiguous()
    if lib.numel() != 1:
        raise ValueError("the ATE estimator takes one library size shared by all conditions (eval.py:95)")
    D_all = torch.cat([D_ctl, D_alt], dim=0)
    C, L, G = D_all.shape[0], model.n_latent, model.n_phenos
    # the conditions' dosages as lists of their non-zeros, built once: every D @ (E * mask) below is a gather of
    # <= kmax rows per condition (the reference runs one dense model call per perturbation, :178-195)
    kmax = max(1, int((D_all != 0).sum(dim=1).max().item()))
    plan = F_hip.DosagePlan(D_all, kmax) if kmax <= 64 else None
    share_z = dosage_independent_variables is not None and "z_basal" in dosage_independent_variables
    # Gene order of the decoder's output layer for this call.  K9 searches a gene's count upwards from 0, a wave's 64 lanes
    # side by side, each loop as long as its slowest lane: the genes a wave holds should be alike in expected count.  Their
    # order in the data is arbitrary, so the output layer's rows (and log_concentration) are permuted by the mean logit
    # over a fixed probe set of basal states -- seeded by `seed`, not by the chunk: every rank and every split of the
    # particles uses the same order, and with it the same draws -- and the sums are put back in the data's order at the end.
    dec = model.decoder
    lin = dec.normalized_mean_decoder[0]
    perm = None
    # (a wave's vector slot covers 256 consecutive genes: below that there is nothing to sort apart)
    sort_env = os.environ.get("CRADLE_ATE_SORT_GENES", "1")  # "0": off; "2": always, for the mean "sampler" too (tests)
    if (sampler == "nb" and sort_env != "0" and G > 256) or sort_env == "2":
        pg = torch.Generator(device=device).manual_seed((seed * 1000003 + 999983) & (2**63 - 1))
        zp = torch.cat([torch.randn((256, L), device=device, generator=pg), torch.zeros((256, 2 * L), device=device)], dim=-1)
        score = F_hip.linear(dec.mlp(zp), lin.weight, lin.bias)[:, :G].mean(dim=0)
        perm = torch.argsort(score)
    W_out = lin.weight.detach() if perm is None else lin.weight.detach()[perm].contiguous()
    b_out = lin.bias.detach() if perm is None else lin.bias.detach()[perm].contiguous()
    log_c = dec.log_concentration.detach()
    gene_consts = ops.nb_gene_consts(log_c if perm is None else log_c[perm].contiguous())
    sums_rows = torch.zeros((C, (G + 3) & ~3), device=device, dtype=torch.float32)  # the sampler's row length
    sums = sums_rows[:, :G]

    world, rank = parallel.world_and_rank()
    starts = list(range(0, n_particles, batch_size))
    # draws beside the next decoder pass (CRADLE_ATE_OVERLAP=0: one stream; also when every launch is being timed on its
    # own, bench.py's kernel figures)
    draw_stream, in_flight = None, []
    if device.type == "cuda" and os.environ.get("CRADLE_ATE_OVERLAP", "1") != "0" and ops.KERNEL_TIMING != "all":
        draw_stream = _draw_stream(device)
    # q(qc_E)'s mean and scale once for the whole call, not once per particle chunk (guide.frozen_qc_dist)
    with guide.frozen_qc_dist() if hasattr(guide, "frozen_qc_dist") else contextlib.nullcontext():
        for ci in parallel.my_chunks(len(starts), world, rank):
            i = starts[ci]
            c = min(batch_size, n_particles - i)
            gen = _chunk_generator(device, seed, ci)
            nz = {}
            if noise_chunks is not None:
                nz = {k: v.to(device=device, dtype=torch.float32) for k, v in noise_chunks[ci].items()}
            plated_cv = {k: v for k, v in cv.items() if k in ("mask", "E", "qc_E")}
            # :143-145; draws that are not injected come from this chunk's generator inside the guide (in slabs of
            # treatments when there are thousands: no (c, T, L) noise tensors up front)
            _, gs = guide(n_particles=c, condition_values=plated_cv, noise=nz, generator=gen)
            EM = gs["E"] * gs["mask"] if "E" in plated_cv else gs["E"].mul_(gs["mask"])  # in place: E is this call's own
            del gs
            # conditions in slabs so that one decoder pass stays within MAX_ROWS_PER_LAUNCH rows
            slab = max(1, min(MAX_ROWS_PER_LAUNCH, MAX_LOGIT_FLOATS // G) // c)
            slab = -(-C // -(-C // slab))  # equal slabs: 237 conditions go as 119 + 118, not 209 + 28 (a small GEMM tail)
            z_shared = None
            for s0 in range(0, C, slab):
                s1 = min(C, s0 + slab)
                n = s1 - s0
                if "z_basal" in cv:
                    z = cv["z_basal"].to(device)
                    z = (z.unsqueeze(0) if z.dim() == 2 else z).expand(c, -1, -1)
                    z = z.expand(c, C, L)[:, s0:s1] if z.shape[1] == 1 else z[:, s0:s1]
                elif "z_control" in nz:
                    zc, za = nz["z_control"], nz["z_alt"]  # (c,1,L), (n_alt,c,1,L)
                    z_all = zc.expand(c, C, L) if share_z else torch.cat([zc, za[:, :, 0, :].permute(1, 0, 2)], dim=1)
                    z = z_all[:, s0:s1]
                elif share_z:  # :165-167: z_basal of the control call is reused for every alternative
                    if z_shared is None:
                        z_shared = torch.randn((c, 1, L), device=device, generator=gen)
                    z = z_shared.expand(c, n, L)
                else:  # p_z_basal = Normal(0, 1) drawn per call (model.py:82)
                    z = torch.randn((c, n, L), device=device, generator=gen)
                off = (F_hip.dosage_matmul(plan.rows(s0, s1), EM, c) if plan is not None
                       else F_hip.dense_dosage_matmul(D_all[s0:s1], EM, c))
                z_good = torch.cat([z, off, torch.zeros((c, n, L), device=device)], dim=-1)  # model.py:121
                logits = F_hip.linear(dec.mlp(z_good.reshape(c * n, -1)), W_out, b_out)  # model.decoder (mlp.py:145-154)
                call_seed = (seed * 7919 + ci) * 4099 + s0
                if draw_stream is None:
                    ops.nb_sample_accumulate(
                        logits, lib, gene_consts, c, n, sums=sums[s0:s1], method=method, sampler=sampler, seed=call_seed)
                    continue
                # The sampler is bound by its vector ALU work, the decoder by the matrix pipes: this slab's draws run
                # on a stream of their own beside the next slab's decoder pass.  Launches on one stream keep their
                # order, so every row of `sums` is added to in the same order as without the overlap: same numbers.
                # At most three slabs of logits are alive (0.3 GB each at Norman's size): the host waits for the draws of
                # the slab three back (measured: 61.0 ms with two, 59.1 with three or five, 63.0 on one stream).
                ready = torch.cuda.Event()
                ready.record()
                with torch.cuda.stream(draw_stream):
                    draw_stream.wait_event(ready)
                    ops.nb_sample_accumulate(
                        logits, lib, gene_consts, c, n, sums=sums[s0:s1], method=method, sampler=sampler, seed=call_seed)
                    drawn = torch.cuda.Event()
                    drawn.record()
                in_flight.append((drawn, logits))
                del logits
                if len(in_flight) > 2:
                    in_flight.pop(0)[0].synchronize()
    if draw_stream is not None:
        torch.cuda.current_stream(device).wait_stream(draw_stream)
        in_flight.clear()  # (stream-ordered: whatever reuses the logits' memory is enqueued behind the wait)
    parallel.allreduce_sum_(sums_rows, world)
    if perm is not None:  # back to the data's gene order
        ordered = torch.empty_like(sums)
        ordered[:, perm] = sums
        sums = ordered
    means = sums / n_particles  # :199-200
    return (means[1:] - means[:1]).cpu().numpy()
