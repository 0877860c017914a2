"""Builds the reference's own ``TrainConfigPerturbationLightningModule`` (perturbation_lightning_module.py:178-247)
from the reference's demo yamls (through its own preprocess_config / add_data_info_to_config) and records what a
Lightning checkpoint of it would hold: the ordered (key, shape, dtype) list of ``state_dict()``, the optimizer it
configures, and a state dict of small random values that the reference module accepts with ``strict=True`` together
with the ELBO it gives on a fixed batch.  Written to tests/golden/lightning_layout.json / lightning_layout.npz.
TEST INFRASTRUCTURE, build container only (reads /root/reference)."""
import json
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_harness as rh  # noqa: E402
from oracle.make_config_golden import REF, reference_functions  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
SIZES = dict(T=11, G=36, Q=1)


class SmallDataModule:
    """What add_data_info_to_config asks a data module for, at a size a CPU test builds in milliseconds."""

    class _Frame:
        def __init__(self, n):
            self.shape = (n, 0)

    def __init__(self, stats):
        self.stats = stats

    def get_d_var_info(self):
        return self._Frame(SIZES["T"])

    def get_x_var_info(self):
        return self._Frame(SIZES["G"])

    def get_qc_var_info(self):
        return self._Frame(SIZES["Q"])

    def get_x_train_statistics(self):
        return self.stats

    def get_qc_n(self):
        return {"n_qc_pass": 900, "n_qc_fail": 100}


def main():
    ref = rh.load_reference()
    from cradle_vae.models.utils.perturbation_lightning_module import TrainConfigPerturbationLightningModule

    pre, add = reference_functions()
    g = torch.Generator().manual_seed(5)
    log_mean = torch.randn(SIZES["G"], generator=g)
    log_std = torch.rand(SIZES["G"], generator=g) + 0.5
    layouts, arrays = {}, {}
    names = sorted(n for n in os.listdir(os.path.join(REF, "demo")) if n.endswith(".yaml"))
    for name in names + ["small_values"]:
        raw = yaml.safe_load(open(os.path.join(REF, "demo", "cradle_vae_norman.yaml" if name == "small_values" else name)))
        if name == "small_values":
            # the norman yaml at widths whose state dict is a few kB: the case that also stores values
            raw.update({"model_kwargs.n_latent": 6, "guide_kwargs.n_latent": 6, "model_kwargs.decoder_n_hidden": 12,
                        "guide_kwargs.basal_encoder_n_hidden": 12, "guide_kwargs.embedding_encoder_n_hidden": 10})
        stats = ref.ObservationNormalizationStatistics(log_x_mean=log_mean.clone(), log_x_std=log_std.clone())
        config = add(pre(dict(raw)), SmallDataModule(stats))
        torch.manual_seed(0)
        T, Q = SIZES["T"], SIZES["Q"]
        counts = dict(D_obs_counts_train=torch.arange(1.0, T + 1), D_obs_counts_val=torch.arange(2.0, T + 2),
                      D_obs_counts_test=torch.arange(3.0, T + 3), qc_obs_counts_train=torch.full((Q,), 9.0),
                      qc_obs_counts_val=torch.full((Q,), 8.0), qc_obs_counts_test=torch.full((Q,), 7.0))
        mod = TrainConfigPerturbationLightningModule(config=config, **counts)
        sd = mod.state_dict()
        opt = mod.configure_optimizers()
        layouts[name] = dict(
            state_dict=[[k, list(v.shape), str(v.dtype)] for k, v in sd.items()],
            n_parameters=sum(p.numel() for p in mod.parameters()),
            optimizer=dict(cls=type(opt).__name__, lr=opt.defaults["lr"], weight_decay=opt.defaults["weight_decay"],
                           betas=list(opt.defaults["betas"]), eps=opt.defaults["eps"],
                           n_param_tensors=sum(len(gr["params"]) for gr in opt.param_groups)),
            pos_weight=mod.loss_module.pos_weight, n_particles=mod.n_particles,
            predictor_shares_modules=(mod.predictor.model is mod.loss_module.model
                                      and mod.predictor.guide is mod.loss_module.guide),
        )
        if name == "small_values":
            # values the reference accepts (strict) + what it computes from them: a checkpoint written by the
            # reference is exactly this dict under "state_dict"
            gen = torch.Generator().manual_seed(11)
            new_sd = {}
            trainable = {k for k, _ in mod.named_parameters(remove_duplicate=False)}
            for k, v in sd.items():
                # parameters get new values; buffers (priors, statistics, obs counts) stay as constructed
                new_sd[k] = 0.1 * torch.randn(v.shape, generator=gen) if k in trainable else v.clone()
            # tied tensors (predictor.* are the loss module's own) must hold the same values
            for k in list(new_sd):
                if k.startswith("predictor."):
                    new_sd[k] = new_sd["loss_module." + k[len("predictor."):]]
            missing, unexpected = mod.load_state_dict(new_sd, strict=True)
            assert not missing and not unexpected
            N, P = 16, 3
            X = torch.poisson(torch.rand(N, SIZES["G"], generator=gen) * 4)
            cfX = torch.poisson(torch.rand(N, SIZES["G"], generator=gen) * 4)
            D = torch.zeros(N, T)
            D[torch.arange(N), torch.randint(0, T, (N,), generator=gen)] = 1.0
            D[:3] = 0
            qc = (torch.rand(N, Q, generator=gen) < 0.3).float()
            L = config["model_kwargs"]["n_latent"]
            noise = dict(u_mask=torch.rand(P, T, L, generator=gen), eps_E=torch.randn(P, T, L, generator=gen),
                         eps_qc_E=torch.randn(P, Q, L, generator=gen), eps_z=torch.randn(P, N, L, generator=gen))
            cf_noise = dict(u_mask=torch.rand(P, T, L, generator=gen), eps_E=torch.randn(P, T, L, generator=gen),
                            eps_qc_E=torch.randn(P, Q, L, generator=gen), eps_z=torch.randn(P, N, L, generator=gen))
            batch = dict(X=X, cfX=cfX, D=D, qc=qc, library_size=X.sum(1), idx=torch.arange(N),
                         D_obs_counts=mod.D_obs_counts_train, qc_obs_counts=mod.qc_obs_counts_train)
            loss, metrics, _ = rh.reference_loss_and_grads(mod.loss_module, batch, noise, cf_noise, P)
            arrays.update({f"sd/{k}": v.numpy() for k, v in new_sd.items() if not k.startswith("predictor.")})
            arrays.update({f"batch/{k}": v.numpy() for k, v in batch.items()})
            arrays.update({f"noise/{k}": v.numpy() for k, v in noise.items()})
            arrays.update({f"cf_noise/{k}": v.numpy() for k, v in cf_noise.items()})
            arrays["loss"] = np.asarray(float(loss))
            arrays["n_particles"] = np.asarray(P)
            for k, v in metrics.items():
                arrays[f"metrics/{k}"] = np.asarray(float(v))
            # (the reference's _init_loss_module writes the model and guide objects into loss_module_kwargs)
            plain = {k: v for k, v in config.items() if k not in ("guide_kwargs", "loss_module_kwargs")}
            plain["loss_module_kwargs"] = {k: v for k, v in config["loss_module_kwargs"].items()
                                           if k not in ("model", "guide")}
            layouts[name]["config"] = json.loads(json.dumps(plain))
            gk = {k: v for k, v in config["guide_kwargs"].items() if k != "x_normalization_stats"}
            layouts[name]["config"]["guide_kwargs"] = gk
            arrays["log_x_mean"], arrays["log_x_std"] = log_mean.numpy(), log_std.numpy()
    with open(os.path.join(GOLDEN, "lightning_layout.json"), "w") as f:
        json.dump(layouts, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(GOLDEN, "lightning_layout.npz"), **arrays)
    print({k: (len(v["state_dict"]), v["n_parameters"]) for k, v in layouts.items()}, float(arrays["loss"]))


if __name__ == "__main__":
    sys.exit(main())
