#!/usr/bin/env python3
"""Benchmark of the CRADLE-VAE ELBO training step on MI355X.

    python bench.py --gpus N --steps K --warmup W [--shape norman] [--step-impl auto]

A "step" = one per-minibatch ELBO forward + backward + gradient-norm clip + Adam update
(reference: PerturbationLightningModule.training_step + Lightning's backward/clip/optimizer,
cradle_vae/models/utils/perturbation_lightning_module.py:90-107,147-150, train_norman.py:96)
on one synthetic batch of the Norman shape (demo/cradle_vae_norman.yaml: N=512 cells/GPU,
P=5 particles, G=5000, T=105, L=200, H=400), inputs resident in HBM.  With N>1 (launched by
torch.distributed.run, one rank per GPU) every rank steps its own cell shard and gradients
are averaged with one RCCL all-reduce: weak scaling, value = N_gpus * 512 cells / step time.

Prints ONE JSON line (rank 0).  `roofline` describes the NB-likelihood kernel (K1), timed
with HIP events on its stream inside the timed region; `cpu_baseline` is the CPU oracle
(a port of the reference's PyTorch path, oracle/cradle_oracle.py) timed on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# N > 1: HIP's default of 4 hardware queues per process is too few once a RCCL communicator exists next to the step's
# streams: streams that share a queue run one after the other, and which ones share depends on the order they were
# created in (1.17 ms instead of 0.95 ms per step with a process group merely alive; 0.95 ... 1.3 ms at 4-7 queues
# depending on which stream the step runs on: profiles/r02_ddp_one_rank_emulation.txt).  16 gives every stream a queue
# of its own.  The runtime reads the variable when it starts, so it is set before torch is imported.
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from cradle_vae_amd.env import set_hw_queues_for_multi_gpu  # noqa: E402  (imports nothing but os)

set_hw_queues_for_multi_gpu()  # CRADLE_HW_QUEUES=<1..32>|keep chooses; an existing setting below 6 is raised, with a notice

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def build(shape, device, seed):
    from cradle_vae_amd import synthetic as syn
    from cradle_vae_amd.guide import ObservationNormalizationStatistics, cradle_vae_CorrelatedNormalGuide
    from cradle_vae_amd.loss_module import cradle_vae_ELBOLossModule
    from cradle_vae_amd.model import cradle_vae_Model

    batch = syn.make_batch(shape, seed=seed, device=device)
    torch.manual_seed(0)  # identical initial weights on every rank
    model = cradle_vae_Model(
        n_latent=shape.n_latent, n_treatments=shape.n_treatments, n_phenos=shape.n_phenos, n_qc=shape.n_qc,
        mask_prior_prob=shape.mask_prior_prob, embedding_prior_scale=1.0, likelihood_key="library_nb",
        decoder_n_layers=shape.decoder_n_layers, decoder_n_hidden=shape.decoder_n_hidden, sum_mode=False)
    stats = ObservationNormalizationStatistics(log_x_mean=batch["log_x_mean"].cpu(), log_x_std=batch["log_x_std"].cpu())
    guide = cradle_vae_CorrelatedNormalGuide(
        n_latent=shape.n_latent, n_treatments=shape.n_treatments, n_phenos=shape.n_phenos, n_qc=shape.n_qc,
        basal_encoder_n_layers=shape.basal_encoder_n_layers, basal_encoder_n_hidden=shape.basal_encoder_n_hidden,
        basal_encoder_input_normalization="log_standardize",
        embedding_encoder_n_layers=shape.embedding_encoder_n_layers,
        embedding_encoder_n_hidden=shape.embedding_encoder_n_hidden, x_normalization_stats=stats)
    lm = cradle_vae_ELBOLossModule(model=model, guide=guide, beta=shape.beta).to(device)
    # what trainer.fit takes from the data set (data.TensorPerturbationDataset.max_dosages_per_cell)
    lm.max_dosages_per_cell = max(1, int((batch["D"] != 0).sum(1).max().item()))
    # ... and what data.DeviceBatchLoader hands over with every batch: the distinct treatments in it
    batch["treatments_present"] = int((batch["D"] != 0).any(dim=0).sum().item())
    return lm, batch


MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (256 CUs x 256 flop/clk x 2.4 GHz)


def k1_pmc_traffic(G, N, P):
    """HBM bytes per K1 launch from the committed rocprofv3 --pmc passes (profiles/k1_pmc.json:
    FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate runs of tools/pmc_nb.py); None if no pass was
    made at this shape."""
    path = os.path.join(ROOT, "profiles", "k1_pmc.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except OSError:
        return None
    if d.get("shape") != {"G": G, "N": N, "P": P}:
        return None
    return d.get("traffic_bytes")


def profiled_avg_us(substr):
    """Average duration of a kernel in the committed rocprofv3 --kernel-trace --stats summary of this command
    (profiles/r03_bench_kernel_stats.csv), for the reader's comparison with the live figure: the HIP events of the live
    bracket add their own ~3 us of marker packets between dependent kernels.  None when the file is not there."""
    import csv

    path = os.path.join(ROOT, "profiles", "r03_bench_kernel_stats.csv")
    try:
        with open(path) as f:
            rows = [r for r in csv.DictReader(f) if substr in r.get("Name", "")]
        return float(rows[0]["AverageNs"]) * 1e-3 if rows else None
    except (OSError, KeyError, ValueError):
        return None


def k9_pmc_traffic(G, C, P):
    """HBM bytes per K9 launch (sampler + partial sums) from the committed rocprofv3 --pmc passes
    (profiles/k9_pmc.json, tools/pmc_round.sh); None if no pass was made at this launch shape."""
    try:
        with open(os.path.join(ROOT, "profiles", "k9_pmc.json")) as f:
            d = json.load(f)
    except OSError:
        return None
    return d.get("traffic_bytes") if d.get("shape") == {"G": G, "C": C, "P": P} else None


def measured_copy_gbs(device, mbytes=512, reps=10):
    """Achievable HBM bandwidth on this device as a plain device-to-device copy sees it (read + write
    bytes over time): context for the K1 fraction, which is quoted against the 8 TB/s peak."""
    n = mbytes * (1 << 20) // 4
    a, b = torch.empty(n, device=device), torch.empty(n, device=device)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def treatment_slots(s, lm, batch):
    """Length of the treatment axis the fused step evaluates for this batch: all T, or -- when fewer -- the bound
    N * kmax, or the distinct treatments the loader counted, rounded up to 32 (cradle_step_desc.max_present)."""
    slots = min(s.n_treatments, s.n_cells * lm.max_dosages_per_cell)
    if batch.get("treatments_present") is not None:
        slots = min(slots, max(32, (batch["treatments_present"] + 31) // 32 * 32))
    return slots


def gemm_flops_fwd(s, slots=None):
    """Forward MLP GEMM flops of one step in the minimal form of SURVEY 8(d) (2 flops per MAC); the
    embedding encoder is counted for the treatment slots the step evaluates."""
    N, P, G, T, L = s.n_cells, s.n_particles, s.n_phenos, s.n_treatments, s.n_latent
    if slots:
        T = min(T, slots)
    He, Hd, Hm = s.basal_encoder_n_hidden, s.decoder_n_hidden, s.embedding_encoder_n_hidden
    f_enc = 2 * 2 * N * G * He + 3 * 2 * P * N * (2 * L * He + 2 * He * L)
    f_dec = 2 * P * N * (3 * L * Hd + Hd * Hd * max(s.decoder_n_layers, 1) + Hd * G)
    f_emb = 2 * 2 * P * T * (L * Hm + (s.embedding_encoder_n_layers - 1) * Hm * Hm + 2 * Hm * L)
    return float(f_enc + f_dec + f_emb)


def loss_parity(lm, shape, batch, device):
    """One forward of the fused HIP step on the bench's own batch and its trained parameters against the
    fp64 CPU oracle on the same injected noise -> relative error of the loss, the relative error of every loss
    term (|got - ref| / max(1, |ref|); the parity bar is 1e-4, tests/test_fused_fullsize_gpu.py) and the number of
    straight-through mask bits that differ from the oracle's (a relaxed-Bernoulli logit within fp32 rounding of zero
    rounds to the other side than in fp64: a different -- equally valid -- discrete sample, not an arithmetic
    error; terms are only comparable when there are none)."""
    from cradle_vae_amd import synthetic as syn
    from oracle import cradle_oracle as co

    n1, n2 = syn.make_noise(shape, 100), syn.make_noise(shape, 200)
    dev = lambda d: {k: v.to(device) for k, v in d.items()}
    with torch.no_grad():
        loss, metrics = lm.loss(
            X=batch["X"], D=batch["D"], D_obs_counts=batch["D_obs_counts"], qc=batch["qc"],
            qc_obs_counts=batch["qc_obs_counts"], cfX=batch["cfX"],
            condition_values={"library_size": batch["library_size"]}, n_particles=shape.n_particles,
            noise=dev(n1), cf_noise=dev(n2))
    torch.cuda.synchronize()
    f64 = lambda d: {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in d.items()}
    params, b = f64(lm.state_dict()), f64({k: v for k, v in batch.items() if k != "idx" and torch.is_tensor(v)})
    with torch.no_grad():
        ref, ref_metrics, inter = co.elbo_loss(
            params, X=b["X"], D=b["D"], qc=b["qc"], cfX=b["cfX"], library_size=b["library_size"],
            D_obs_counts=b["D_obs_counts"], qc_obs_counts=b["qc_obs_counts"], noise=f64(n1), cf_noise=f64(n2),
            n_particles=shape.n_particles, beta=shape.beta)
    terms = {k: abs(float(metrics[k]) - float(v)) / max(1.0, abs(float(v))) for k, v in ref_metrics.items()}
    flips = None
    fs = getattr(lm, "_fused", None)
    if fs is not None and fs.desc.N * fs.desc.kmax >= fs.desc.T and not fs.desc.max_present:  # full treatment axis
        P, T, L = shape.n_particles, shape.n_treatments, shape.n_latent
        mask = fs.intermediate("mask").view(2 * P, T, L).cpu().double()
        flips = int((mask[:P] != inter["mask"]).sum()) + int((mask[P:] != inter["cf_mask"]).sum())
    terms["oracle_values"] = {k: float(v) for k, v in ref_metrics.items()}
    return abs(float(loss) - float(ref)) / abs(float(ref)), terms, flips


def cpu_baseline(shape, seconds_budget=25.0, as_written=False):
    """The reference's PyTorch path (CPU oracle restatement, fp32, autograd + Adam), timed on
    this host's cores on a bounded sample of the same workload.  ``as_written``: with the work the reference
    does and discards (the x_good/x_bad draws, distribution argument validation)."""
    from cradle_vae_amd import synthetic as syn
    from oracle import cradle_oracle as co

    torch.set_num_threads(max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "64")))))
    b = syn.make_batch(shape, seed=0)
    params = co.init_params(
        n_latent=shape.n_latent, n_treatments=shape.n_treatments, n_phenos=shape.n_phenos, n_qc=shape.n_qc,
        decoder_n_layers=shape.decoder_n_layers, decoder_n_hidden=shape.decoder_n_hidden,
        basal_encoder_n_layers=shape.basal_encoder_n_layers, basal_encoder_n_hidden=shape.basal_encoder_n_hidden,
        embedding_encoder_n_layers=shape.embedding_encoder_n_layers,
        embedding_encoder_n_hidden=shape.embedding_encoder_n_hidden, mask_prior_prob=shape.mask_prior_prob,
        log_x_mean=b["log_x_mean"], log_x_std=b["log_x_std"], randomize=False)
    keys = co.trainable_keys(params)
    leaves = [params[k].clone().requires_grad_(True) for k in keys]
    opt = torch.optim.Adam(leaves, lr=3e-4, weight_decay=1e-6)
    times = []
    t_start = time.perf_counter()
    it = 0
    while True:
        n1, n2 = syn.make_noise(shape, 100 + it), syn.make_noise(shape, 200 + it)
        p2 = dict(params)
        p2.update(dict(zip(keys, leaves)))
        t0 = time.perf_counter()
        loss, _, _ = co.elbo_loss(
            p2, X=b["X"], D=b["D"], qc=b["qc"], cfX=b["cfX"], library_size=b["library_size"],
            D_obs_counts=b["D_obs_counts"], qc_obs_counts=b["qc_obs_counts"], noise=n1, cf_noise=n2,
            n_particles=shape.n_particles, beta=shape.beta, as_written=as_written)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(leaves, 100.0)
        opt.step()
        dt = time.perf_counter() - t0
        if it > 0 or dt > seconds_budget / 2:
            times.append(dt)
        it += 1
        if time.perf_counter() - t_start > seconds_budget or it >= 12:
            break
    t = sorted(times)[len(times) // 2]
    return {
        "value": shape.n_cells / t,
        "unit": "cells/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{len(times)} timed ELBO fwd+bwd+Adam steps of the same {shape.name} batch shape "
                  f"(oracle/cradle_oracle.py, fp32, "
                  + ("as the reference writes it: with the unused x_good/x_bad draws and argument validation"
                     if as_written else "minimal form without the unused x draws") + f"), median {t:.3f} s/step",
    }


ATE_N_ALT, ATE_PARTICLES, ATE_CHUNK = 236, 2500, 128  # Norman's perturbations; eval.py:330-331 (n_particles, batch_size)


def ate_workload(shape):
    """dosages of the ATE leg: ATE_N_ALT two-guide perturbations against the empty control (eval.py:90-97)."""
    T = shape.n_treatments
    g = torch.Generator().manual_seed(0)
    D_alt = torch.zeros(ATE_N_ALT, T)
    D_alt[torch.arange(ATE_N_ALT), torch.randint(0, T, (ATE_N_ALT,), generator=g)] = 1.0
    D_alt[torch.arange(ATE_N_ALT), torch.randint(0, T, (ATE_N_ALT,), generator=g)] = 1.0
    return D_alt, torch.zeros(1, T)


def ate_cpu_baseline(lm, shape, n_alt=40):
    """The reference's ATE loop (oracle.model_average_treatment_effect: one model call per perturbation and particle
    chunk, both decoder variants, x_good and x_bad drawn; pinned draw for draw to the reference by
    tests/golden/ate_sampled_tiny.npz) on this host's cores, on a bounded sample of the same workload."""
    from oracle import cradle_oracle as co

    torch.set_num_threads(max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "64")))))
    params = {k: v.detach().cpu().float() for k, v in lm.state_dict().items()}
    D_alt, D_ctl = ate_workload(shape)
    t0 = time.perf_counter()
    _, calls = co.model_average_treatment_effect(
        params, dosages_alt=D_alt, dosages_control=D_ctl, quality=torch.ones(1, shape.n_qc),
        library_size=1e4 * torch.ones(1), n_particles=ATE_CHUNK, method="perturbseq", batch_size=ATE_CHUNK,
        max_alt=n_alt, max_chunks=1)
    dt = time.perf_counter() - t0
    return {"value": calls * ATE_CHUNK / dt, "unit": "(perturbation, particle) rows/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"one particle chunk of {ATE_CHUNK}: the guide call, the control and {n_alt} of the {ATE_N_ALT} "
                      f"perturbations ({calls} model calls of {ATE_CHUNK} rows, as the reference writes them: both decoder "
                      f"variants, x_good and x_bad drawn), {dt:.1f} s"}


def ate_bench(lm, shape, device, args):
    """The predictor half of north_star: model-side average treatment effects at Norman's evaluation size
    (eval.py:90-97,330-331 -> analysis/average_treatment_effects.py:78-203): ATE_N_ALT perturbations x ATE_PARTICLES
    particles in chunks of ATE_CHUNK, method perturbseq, counts drawn.  Whole call timed (median of 3); one more call with
    HIP events around every GEMM and sampler launch gives the kernel-level figures."""
    from cradle_vae_amd import ops
    from cradle_vae_amd.predictor import cradle_vae_Predictor

    pred = cradle_vae_Predictor(lm.model, lm.guide)
    D_alt, D_ctl = ate_workload(shape)
    kw = dict(dosages_alt=D_alt, dosages_control=D_ctl, quality=torch.ones(1, shape.n_qc), method="perturbseq",
              condition_values=dict(library_size=1e4 * torch.ones(1)), batch_size=ATE_CHUNK)
    pred.estimate_average_treatment_effects(n_particles=ATE_CHUNK, **kw)  # warm-up: one chunk
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = pred.estimate_average_treatment_effects(n_particles=ATE_PARTICLES, **kw)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    ops.KERNEL_TIMERS.clear()
    ops.KERNEL_TIMING = "all"
    pred.estimate_average_treatment_effects(n_particles=ATE_PARTICLES, **kw)
    torch.cuda.synchronize()
    ops.KERNEL_TIMING = False
    k9 = [t[1].elapsed_time(t[2]) * 1e-3 for t in ops.KERNEL_TIMERS if t[0] == "nb_sample_accumulate"]
    G, L, Hd = shape.n_phenos, shape.n_latent, shape.decoder_n_hidden
    dec = [(t[0], t[1].elapsed_time(t[2]) * 1e-3) for t in ops.KERNEL_TIMERS if isinstance(t[0], tuple) and t[0][2] >= G]
    ops.KERNEL_TIMERS.clear()
    rows = ATE_PARTICLES * (ATE_N_ALT + 1)
    res = {"workload": f"{shape.name}: {ATE_N_ALT} perturbations + control x {ATE_PARTICLES} particles (chunks of {ATE_CHUNK}), "
                       f"G={G}, method perturbseq, Gamma-Poisson counts drawn (eval.py:90-97,330-331)",
           "ms": dt * 1e3, "value": rows / dt, "unit": "(perturbation, particle) rows/s", "runs_ms": [t * 1e3 for t in times],
           "finite": bool(torch.isfinite(torch.from_numpy(out.X)).all())}
    if k9:
        # K9 per launch: reads 4*G B of logits per (particle, perturbation) row, writes 4*G B per perturbation
        conds = ATE_N_ALT + 1
        slab = max(1, min(1 << 16, (1 << 27) // G) // ATE_CHUNK)
        slab = -(-conds // -(-conds // slab))  # ate.py: equal slabs of perturbations per decoder pass
        alg = 4.0 * G * (ATE_CHUNK * slab + slab)
        full = [t for t in k9 if t >= 0.5 * max(k9)]  # (the last chunk of 2500 = 19 x 128 + 68 is shorter)
        t_k9 = sum(full) / len(full)
        res["roofline"] = {"bound": "hbm", "achieved": alg / t_k9 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": alg / t_k9 / 1e9 / HBM_PEAK_GBS, "traffic": k9_pmc_traffic(G, slab, ATE_CHUNK),
                           "kernel": "nb_sample_accumulate_kernel + partial_sums_kernel (K9: softmax, Gamma-Poisson draw, "
                                     "perturb-seq transform, particle sum)",
                           "launch_us": t_k9 * 1e6, "launches_timed": len(full), "algorithmic_bytes_per_launch": alg,
                           "rows_per_launch": ATE_CHUNK * slab, "total_ms": sum(k9) * 1e3,
                           "note": "HBM is the nominal bound; the kernel is bound by its random-number and "
                                   "transcendental ALU work (DESIGN 4, K9)"}
    if dec:
        fl = sum(2.0 * k[1] * k[2] * k[3] for k, _ in dec)
        tt = sum(t for _, t in dec)
        res["mfma"] = {"kernel": "gemm_f32_kernel NT (decoder output layer of every particle chunk x perturbation slab)",
                       "flops": fl, "total_ms": tt * 1e3, "achieved": fl / tt / 1e12, "peak": MFMA_F32_PEAK_TF,
                       "unit": "TFLOP/s", "frac": fl / tt / 1e12 / MFMA_F32_PEAK_TF, "launches_timed": len(dec),
                       "decoder_flops_per_row": 2.0 * (3 * L * Hd + Hd * Hd * max(shape.decoder_n_layers, 1) + Hd * G),
                       "whole_call_frac": 2.0 * (3 * L * Hd + Hd * Hd * max(shape.decoder_n_layers, 1) + Hd * G) * rows
                                          / dt / 1e12 / MFMA_F32_PEAK_TF}
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = ate_cpu_baseline(lm, shape)
    return res


def secondary_shape(name, device, args, precision="f32"):
    """The same step at another shape (north_star quotes the Replogle-like one too) or in the optional split-bf16
    GEMM mode, measured after the main region on the same device: side information, not the headline metric."""
    from cradle_vae_amd import synthetic as syn
    from cradle_vae_amd.trainer import TrainStep

    shape = syn.SHAPES[name]
    lm, batch = build(shape, device, seed=1000)
    lm.precision = precision
    st = TrainStep(lm, lr=3e-4, weight_decay=1e-6, gradient_clip_norm=100.0, world_size=1)
    steps = max(10, min(args.steps, 50))
    for _ in range(5):
        st(batch, n_particles=shape.n_particles)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st(batch, n_particles=shape.n_particles)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = {"value": shape.n_cells / dt, "unit": "cells/s", "ms_per_step": dt * 1e3, "steps": steps,
           "workload": f"{shape.name}: N={shape.n_cells}, P={shape.n_particles}, G={shape.n_phenos}, "
                       f"T={shape.n_treatments} ({treatment_slots(shape, lm, batch)} "
                       f"treatment slots), L={shape.n_latent}"}
    if precision != "f32":
        res["dtype"] = ("bf16x3 (bf16 hi+lo operands, 3 products, fp32 accumulate)" if precision == "bf16x3" else
                        "bf16 (operands rounded to bf16, fp32 accumulate)") + " in the GEMMs of >= 1 GFLOP; f32 elsewhere"
        if not args.no_cpu_baseline:
            rel, terms, flips = loss_parity(lm, shape, batch, device)
            terms.pop("oracle_values")
            res["loss_rel_err_vs_oracle"], res["loss_terms_max_rel_err_vs_oracle"] = rel, max(terms.values())
            res["mask_bits_differing_from_oracle"] = flips
    return res


def data_parallel_setup(lm, batch, shape, world, rank, dev_index, device, args):
    """N > 1, before the warm-up: (1) every rank reports its device through one all-gather -- the first collective of
    the run, so a communicator that does not span the ranks fails here, by name; (2) unless the environment pins it,
    the gradient-exchange form is MEASURED instead of assumed: a few steps with one bucket after the step and with
    the output layer's bucket reduced under the rest of the backward pass (CRADLE_BUCKETS = 1 / 2), each as
    reduce-scatter + sharded Adam + all-gather and as all-reduce + full Adam (CRADLE_GRAD_SYNC, trainer.TrainStep);
    the fastest of the four (max over ranks) is what the timed region runs.  One-rank emulation could not settle this
    (the overlapped form stalls there); the ranks decide from the same all-reduced numbers."""
    import torch.distributed as dist

    from cradle_vae_amd import parallel
    from cradle_vae_amd.trainer import TrainStep

    mine = torch.tensor([rank, dev_index, torch.cuda.device_count()], device=device, dtype=torch.float32)
    seen = parallel.all_gather_small(mine, world).view(world, 3).cpu().long().tolist()
    info = {"backend": dist.get_backend(), "ranks": [{"rank": r, "device": d, "visible_devices": v} for r, d, v in seen],
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")}
    if "CRADLE_BUCKETS" in os.environ or os.environ.get("CRADLE_BENCH_AUTOTUNE", "1") == "0" or lm.step == "modular":
        info["buckets"] = os.environ.get("CRADLE_BUCKETS", "1")
        info["chosen_by"] = "environment"
        return info
    # candidates: buckets x exchange form (reduce-scatter + sharded Adam + all-gather, or all-reduce + full Adam on
    # every rank); an exchange form pinned by the environment is not varied
    syncs = [os.environ["CRADLE_GRAD_SYNC"]] if "CRADLE_GRAD_SYNC" in os.environ else ["rs_ag", "allreduce"]
    timings = {}
    for sync in syncs:
        for nb in ("1", "2"):
            os.environ["CRADLE_BUCKETS"], os.environ["CRADLE_GRAD_SYNC"] = nb, sync
            st = TrainStep(lm, lr=3e-4, weight_decay=1e-6, gradient_clip_norm=100.0, world_size=world)
            if st.fused is None:
                break
            if st.grad_sync != sync:  # (a world size the arena's segments do not divide by: rs_ag falls back)
                continue
            for _ in range(3):
                st(batch, n_particles=shape.n_particles)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(10):
                st(batch, n_particles=shape.n_particles)
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
            parallel._all_reduce(t, op=dist.ReduceOp.MAX)
            timings[f"{sync}/{nb}"] = float(t) / 10 * 1e3
            trace(f"gradient exchange {sync} with CRADLE_BUCKETS={nb}: {timings[f'{sync}/{nb}']:.3f} ms/step")
            del st
    best = min(timings, key=timings.get) if timings else f"{syncs[0]}/1"
    os.environ["CRADLE_GRAD_SYNC"], os.environ["CRADLE_BUCKETS"] = best.split("/")
    info.update(buckets=best.split("/")[1], chosen_by="measured before the warm-up (10 steps each, max over ranks)",
                ms_per_step_by_sync_and_buckets=timings)
    return info


def trace(msg):
    if os.environ.get("CRADLE_BENCH_TRACE"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--shape", default="norman")
    ap.add_argument("--preheat-steps", type=int, default=100,
                    help="untimed steps before the warm-up steps: brings an idle chip up to its clocks (0 = none)")
    ap.add_argument("--step-impl", default="auto", choices=["auto", "fused", "modular"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the Replogle-shaped side measurement (N=1 only)")
    ap.add_argument("--no-ate", action="store_true", help="skip the ATE (predictor) leg (N=1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for rehearsals)")
    args = ap.parse_args()

    from cradle_vae_amd import hip, ops
    from cradle_vae_amd import synthetic as syn

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    dev_index = local_rank % max(1, torch.cuda.device_count())  # wraps only in a gloo rehearsal on fewer GPUs than ranks
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    trace("process group ready")
    hip.load()

    shape = syn.SHAPES[args.shape]
    lm, batch = build(shape, device, seed=1000 + rank)  # each rank: its own cell shard
    lm.step = args.step_impl
    from cradle_vae_amd.trainer import TrainStep

    dp_info = None
    if world > 1:
        dp_info = data_parallel_setup(lm, batch, shape, world, rank, dev_index, device, args)
    stepper = TrainStep(lm, lr=3e-4, weight_decay=1e-6, gradient_clip_norm=100.0, world_size=world)

    if stepper.fused is not None:
        # K1 and the largest GEMM are timed with HIP events recorded by the C code around their launches (rank 0)
        stepper.fused.k1_events = [] if rank == 0 else None
        stepper.fused.gemm_events = [] if rank == 0 else None
        # every 4th step of the timed region carries the four timing events (each is a barrier packet
        # of ~5 us between dependent kernels); the roofline averages over those launches
        stepper.fused.event_every = 4

    def one_step(i):
        return stepper(batch, n_particles=shape.n_particles)

    trace("modules built")
    # A chip that has idled (the seconds of set-up above) runs 10-15 % slow for its first tens of milliseconds under
    # load (tools/first_steps.py: 1.12 ms per step falling to 0.96 over ~20 steps after 50 ms of idle): without this,
    # a short timed region measures the clock ramp, not the step.  Untimed, outside --warmup and --steps, reported in
    # the line as config.preheat_steps; --preheat-steps 0 turns it off.
    for i in range(args.preheat_steps):
        one_step(i)
    for i in range(args.warmup):
        one_step(i)
        trace(f"warmup step {i} enqueued")
    torch.cuda.synchronize()
    trace("warmup done")
    if world > 1:
        dist.barrier()
    ops.KERNEL_TIMERS.clear()
    ops.KERNEL_TIMING = True
    if stepper.fused is not None and stepper.fused.k1_events is not None:
        stepper.fused.k1_events.clear()
        if stepper.fused.gemm_events is not None:
            stepper.fused.gemm_events.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = one_step(i)
    enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3  # host time to issue a step (no wait on the device)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    trace("timed region done")
    ops.KERNEL_TIMING = False
    if world > 1:
        from cradle_vae_amd import parallel

        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        parallel._all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ms_per_step = elapsed / args.steps * 1e3
    cells_per_s = shape.n_cells * world / (elapsed / args.steps)

    if rank == 0:
        # ---- roofline of K1 from the HIP events recorded around its launches in the timed region
        P, N, G = shape.n_particles, shape.n_cells, shape.n_phenos
        # (HIP events recorded by the C code immediately around nb_rows_kernel, on its launch stream)
        k1 = [t[1].elapsed_ms() * 1e-3 for t in ops.KERNEL_TIMERS if t[0] == "nb_logprob_fwd_bwd"]
        if stepper.fused is not None:
            k1 = [ev.elapsed_ms() * 1e-3 for ev in stepper.fused.k1_events]
        V = stepper.decoder_variants  # decoder variants scored per cell (2 = as the reference, 1 = live variant only)
        alg_bytes = 4.0 * G * (3 * P * N + 2 * N)  # SURVEY 8(d) per launch of one variant: fwd+bwd
        roofline = None
        if k1:
            t_k1 = sum(k1) / len(k1)
            ach = alg_bytes / t_k1 / 1e9
            roofline = {
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": k1_pmc_traffic(G, N, P),
                "kernel": "nb_rows_kernel<BWD> (K1 fused fwd+bwd, one decoder variant per launch)",
                "launch_us": t_k1 * 1e6, "launches_per_step": float(V), "launches_timed": len(k1),
                "algorithmic_bytes_per_launch": alg_bytes,
                "copy_gbs_measured": measured_copy_gbs(device),
                "launch_us_in_committed_rocprof_stats": profiled_avg_us("nb_rows_kernel<5, 256, true, false, true>")
                if (G, N, P) == (5000, 512, 5) else None,
            }
        # MFMA side: minimal-form GEMM flops of the step (fwd+bwd = 3x fwd) over the whole
        # step time, against the dense fp32 matrix-core peak -- a lower bound on MFMA utilisation
        # while the GEMMs run, since the step time also holds every non-GEMM kernel.
        gf = gemm_flops_fwd(shape, treatment_slots(shape, lm, batch)) * 3
        mfma = {"flops_per_step": gf, "achieved": gf / (elapsed / args.steps) / 1e12, "peak": MFMA_F32_PEAK_TF,
                "unit": "TFLOP/s", "frac": gf / (elapsed / args.steps) / 1e12 / MFMA_F32_PEAK_TF,
                "note": "minimal-form MLP GEMM flops (SURVEY 8d) / whole step time; fp32 v_mfma_f32_32x32x2_f32"}
        if stepper.fused is not None and stepper.fused.gemm_events:
            # the largest GEMM alone (decoder output layer forward), HIP events around its launch
            tg = [ev.elapsed_ms() * 1e-3 for ev in stepper.fused.gemm_events]
            tg = sum(tg) / len(tg)
            fl = 2.0 * P * N * G * shape.decoder_n_hidden
            mfma["largest_gemm"] = {"kernel": "gemm_f32_kernel NT (decoder output layer fwd)", "M": P * N, "N": G,
                                    "K": shape.decoder_n_hidden, "flops": fl, "launch_us": tg * 1e6,
                                    "achieved": fl / tg / 1e12, "frac": fl / tg / 1e12 / MFMA_F32_PEAK_TF}
        out = {
            "metric": f"cells/sec ELBO fwd+bwd ({shape.name.capitalize()} shape)",
            "value": cells_per_s,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "host_enqueue_ms_per_step": enqueue_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{shape.name}: ELBO fwd+bwd+clip+Adam, N={N} cells/GPU, P={P}, G={G}, "
                            f"T={shape.n_treatments}, L={shape.n_latent}, H={shape.decoder_n_hidden}",
                "step_impl": stepper.impl,
                "preheat_steps": args.preheat_steps,
                "decoder_variants_per_cell": V, "parallelism": f"dp{world}",
                **({"data_parallel": dict(dp_info, grad_sync=getattr(stepper, "grad_sync", None),
                                          overlap=getattr(stepper, "overlap", None))} if dp_info else {}),
                "treatment_slots": treatment_slots(shape, lm, batch),
                "loss": float(loss),
            },
            "roofline": roofline,
            "mfma": mfma,
        }
        if world == 1 and not args.no_ate and shape.n_treatments <= 4096:
            out["ate"] = ate_bench(lm, shape, device, args)
        if world == 1 and args.shape == "norman" and not args.no_secondary:
            out["replogle_shape"] = secondary_shape("replogle", device, args)
            # optional arithmetic mode, a side number: the headline above is fp32 (the reference's precision)
            out["bf16x3"] = secondary_shape("norman", device, args, precision="bf16x3")
            out["bf16"] = secondary_shape("norman", device, args, precision="bf16")
        if not args.no_cpu_baseline and world == 1 and shape.n_treatments > 4096:
            # the CPU restatement does what the reference does: a dense (T, T) one-hot through the embedding encoder
            # and (P, T, L) tensors over every treatment -- minutes and tens of GB per step at T = 10 000
            out["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"not run: the reference's dense form at T={shape.n_treatments} treatments"}
        elif not args.no_cpu_baseline and world == 1:
            if stepper.fused is not None:
                rel, terms, flips = loss_parity(lm, shape, batch, device)
                out["loss_rel_err_vs_oracle"] = rel
                out["loss_terms_oracle_values"] = terms.pop("oracle_values")
                out["loss_terms_max_rel_err_vs_oracle"] = max(terms.values())
                out["loss_terms_rel_err_vs_oracle"] = terms
                out["mask_bits_differing_from_oracle"] = flips
            out["cpu_baseline"] = cpu_baseline(shape, seconds_budget=15.0)
            aw = cpu_baseline(shape, seconds_budget=12.0, as_written=True)
            out["cpu_baseline"]["as_written"] = {k: aw[k] for k in ("value", "unit", "sample")}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()  # rank 0 is still reading its events / printing: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
