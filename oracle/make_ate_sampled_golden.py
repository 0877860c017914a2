"""Writes tests/golden/ate_sampled_tiny.npz: the reference's own estimate_model_average_treatment_effect
(/root/reference/cradle_vae/analysis/average_treatment_effects.py:78-203) run verbatim in the build container
WITH its count draws -- torch.manual_seed(seed) as the function itself does (:92), every draw from torch's global
CPU generator -- so that oracle.cradle_oracle.model_average_treatment_effect (the "as written" CPU port bench.py
times as the ATE baseline) is pinned to it draw for draw.

The Gamma -> Poisson sampler is pyro's (pyro-ppl 1.9.1 GammaPoisson.sample, not in the reference tree): the harness
stub restates it as Poisson(Gamma(c, rate).sample()), the published composition.  That piece stays "parity unpinned by
the reference"; everything around it (call order, RNG consumption, transforms, accumulation) is the reference's code.

    python -m oracle.make_ate_sampled_golden          (build container only: needs /root/reference)
"""
import json
import os

import numpy as np
import torch

from cradle_vae_amd import synthetic as syn
from oracle import cradle_oracle as co
from oracle import ref_harness as rh
from oracle.make_golden import GOLDEN, cfg_of, init_kwargs


def main():
    ref = rh.load_reference()
    rh._GammaPoisson.SAMPLE_MODE = "random"
    shape, seed = syn.TINY, 13
    cfg = cfg_of(shape)
    batch = syn.make_batch(shape, seed=seed, dtype=torch.float32)
    params = co.init_params(**init_kwargs(cfg), log_x_mean=batch["log_x_mean"], log_x_std=batch["log_x_std"],
                            seed=seed + 100, dtype=torch.float32)
    lm = rh.build_reference_loss_module(ref, params, cfg, torch.float32)
    T, Q = cfg["n_treatments"], cfg["n_qc"]
    n_alt, particles, chunk = 4, 20, 8
    D_alt = torch.zeros(n_alt, T)
    for i in range(n_alt):
        D_alt[i, i % T] = 1.0
    D_alt[n_alt - 1, n_alt % T] = 1.0  # a 2-guide combination
    D_ctl, quality, lib = torch.zeros(1, T), torch.ones(1, Q), 1e4 * torch.ones((1,))
    out = {"cfg": np.array(json.dumps(cfg)), "D_alt": D_alt.numpy(), "D_control": D_ctl.numpy(),
           "n_particles": np.array(particles), "chunk": np.array(chunk)}
    for k, v in params.items():
        out[f"params/{k}"] = v.numpy()
    for method in ("mean", "perturbseq"):
        for s in (0, 5):
            ate = ref.estimate_model_average_treatment_effect(
                model=lm.model, guide=lm.guide, dosages_alt=D_alt, dosages_control=D_ctl, quality=quality,
                n_particles=particles, method=method, condition_values=dict(library_size=lib.clone()),
                batch_size=chunk, seed=s)
            out[f"ate/{method}/seed{s}"] = np.asarray(ate)
            mine, calls = co.model_average_treatment_effect(
                params, dosages_alt=D_alt, dosages_control=D_ctl, quality=quality, library_size=lib,
                n_particles=particles, method=method, batch_size=chunk, seed=s)
            err = float((mine - torch.from_numpy(np.asarray(ate))).abs().max())
            print(f"{method} seed {s}: |oracle - reference|_max = {err:.3e} over {calls} model calls, "
                  f"max |ATE| {float(np.abs(ate).max()):.3f}")
            assert err <= 1e-5 * max(1.0, float(np.abs(ate).max())), "the port does not follow the reference's draws"
    path = os.path.join(GOLDEN, "ate_sampled_tiny.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: ({os.path.getsize(path)} B)")


if __name__ == "__main__":
    main()
