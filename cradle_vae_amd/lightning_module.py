"""Training/evaluation module -- mirror of the reference's ``PerturbationLightningModule`` and
``TrainConfigPerturbationLightningModule`` (cradle_vae/models/utils/perturbation_lightning_module.py:
13-247).

With pytorch_lightning importable the classes derive from ``pl.LightningModule`` and are driven
by ``pl.Trainer`` exactly like the reference's.  Without it (this image) they derive from a small
stand-in with the same ``log`` / ``save_hyperparameters`` surface and are driven by
``cradle_vae_amd.trainer.fit`` (one process per GPU, RCCL gradient all-reduce).

``training_step`` returns a scalar whose ``.backward()`` delivers the gradients the fused HIP
step already computed (``cradle_vae_amd.step._FusedLoss``), so Lightning's automatic
optimisation works unchanged.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

try:  # pragma: no cover - pytorch_lightning is absent from the build image
    import pytorch_lightning as pl

    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class _Base(nn.Module):
        """The part of ``pl.LightningModule`` the reference's module uses."""

        def __init__(self):
            super().__init__()
            self.logged: Dict[str, torch.Tensor] = {}
            if not hasattr(self, "hparams"):
                self.hparams = {}

        def log(self, name, value, **kwargs):
            self.logged[name] = value.detach() if torch.is_tensor(value) else value

        def save_hyperparameters(self, *args, ignore=(), **kwargs):
            # called before nn.Module.__init__ in the reference (:197-200): plain attribute only
            object.__setattr__(self, "hparams", {"__ignore__": tuple(ignore)})


def _counts_or_placeholder(value: Optional[torch.Tensor], given: bool, n: int) -> torch.Tensor:
    # the reference keys all three D_* buffers on D_obs_counts_train and all three qc_* buffers
    # on qc_obs_counts_train being given (:47-75)
    if given:
        return value
    return -1 * torch.ones((n,))


class PerturbationLightningModule(_Base):
    def __init__(
        self,
        loss_module: nn.Module,
        lr: float,
        n_treatments: int,
        n_qc: int,
        D_obs_counts_train: Optional[torch.Tensor] = None,
        D_obs_counts_val: Optional[torch.Tensor] = None,
        D_obs_counts_test: Optional[torch.Tensor] = None,
        qc_obs_counts_train: Optional[torch.Tensor] = None,
        qc_obs_counts_val: Optional[torch.Tensor] = None,
        qc_obs_counts_test: Optional[torch.Tensor] = None,
        weight_decay: float = 1e-6,
        n_particles: int = 5,
        predictor: Optional[nn.Module] = None,
        x_var_info=None,
        use_scheduler=False,
    ):
        super().__init__()
        self.loss_module = loss_module
        self.lr = lr
        self.weight_decay = weight_decay
        self.n_treatments = n_treatments
        self.n_qc = n_qc
        self.predictor = predictor
        self.x_var_info = x_var_info
        self.use_scheduler = use_scheduler
        d_given, q_given = D_obs_counts_train is not None, qc_obs_counts_train is not None
        for split, v in (("train", D_obs_counts_train), ("val", D_obs_counts_val), ("test", D_obs_counts_test)):
            self.register_buffer(f"D_obs_counts_{split}", _counts_or_placeholder(v, d_given, n_treatments))
        for split, v in (("train", qc_obs_counts_train), ("val", qc_obs_counts_val), ("test", qc_obs_counts_test)):
            self.register_buffer(f"qc_obs_counts_{split}", _counts_or_placeholder(v, q_given, n_qc))
        self.n_particles = n_particles

    def _get_extra_condition_values(self, batch):
        """:81-88 -- everything but X and D rides along as a condition value."""
        return {k: v for k, v in batch.items() if k not in ("X", "D", "treatments_present")}  # (the last: loader metadata)

    def _step(self, batch, split: str, test: bool):
        loss, metrics = self.loss_module.loss(
            X=batch["X"], D=batch["D"], qc=batch["qc"], cfX=batch["cfX"],
            condition_values=self._get_extra_condition_values(batch),
            D_obs_counts=getattr(self, f"D_obs_counts_{split}"), qc_obs_counts=getattr(self, f"qc_obs_counts_{split}"),
            n_particles=self.n_particles, test=test, x_var_info=self.x_var_info)
        return loss, metrics

    def training_step(self, batch, batch_idx: int = 0):
        loss, metrics = self._step(batch, "train", test=False)  # :90-103
        self.log("train/loss", loss, on_step=True, on_epoch=True, prog_bar=True)
        for k, v in metrics.items():
            self.log(f"train/{k}", v, on_step=True, on_epoch=True)
        return loss

    def validation_step(self, batch, batch_idx: int = 0):
        with torch.no_grad():
            loss, metrics = self._step(batch, "val", test=True)  # :109-122
        self.log("val/loss", loss, on_step=False, on_epoch=True, prog_bar=True)
        for k, v in metrics.items():
            self.log(f"val/{k}", v, on_step=False, on_epoch=True, prog_bar=True)
        return loss

    def test_step(self, batch, batch_idx: int = 0):
        with torch.no_grad():
            loss, metrics = self._step(batch, "test", test=True)  # :128-141
        self.log("test/loss", loss, on_step=False, on_epoch=True, prog_bar=True)
        for k, v in metrics.items():
            self.log(f"test/{k}", v, on_step=False, on_epoch=True, prog_bar=True)
        return loss

    def configure_optimizers(self):
        """:147-164."""
        optimizer = torch.optim.Adam(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        if self.use_scheduler:
            optimizer = torch.optim.SGD(self.parameters(), lr=self.lr, weight_decay=self.weight_decay, momentum=0.9)
            scheduler = torch.optim.lr_scheduler.CyclicLR(
                optimizer, base_lr=0.0005, max_lr=0.005, step_size_up=447, step_size_down=None, mode="triangular")
            return {"optimizer": optimizer,
                    "lr_scheduler": {"scheduler": scheduler, "name": "cyclicLR_lr", "interval": "step"}}
        return optimizer

    def init_weights(self, m):
        """:166-174 (not applied by the reference's constructor either)."""
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)


class TrainConfigPerturbationLightningModule(PerturbationLightningModule):
    """Built from a config dict (:178-247): class names are resolved in ``cradle_vae_amd.models``
    (and, for the data module, in ``cradle_vae_amd.data`` / a ``data_namespace`` the caller
    registers), so the reference's yaml configs work unchanged."""

    data_namespace = None  # object with the data-module classes (the reference's `cradle_vae.data`)

    def __init__(
        self,
        config: Dict,
        D_obs_counts_train: Optional[torch.Tensor] = None,
        D_obs_counts_val: Optional[torch.Tensor] = None,
        D_obs_counts_test: Optional[torch.Tensor] = None,
        qc_obs_counts_train: Optional[torch.Tensor] = None,
        qc_obs_counts_val: Optional[torch.Tensor] = None,
        qc_obs_counts_test: Optional[torch.Tensor] = None,
        x_var_info=None,
        use_scheduler=False,
    ):
        self.save_hyperparameters(
            ignore=["D_obs_counts_train", "D_obs_counts_val", "D_obs_counts_test",
                    "qc_obs_counts_train", "qc_obs_counts_val", "qc_obs_counts_test"])
        object.__setattr__(self, "config", config)
        model = self._init_model()
        guide = self._init_guide()
        loss_module = self._init_loss_module(model, guide)
        predictor = self._init_predictor(model, guide)
        super().__init__(
            loss_module,
            n_treatments=config["model_kwargs"]["n_treatments"],
            n_qc=config["model_kwargs"]["n_qc"],
            D_obs_counts_train=D_obs_counts_train, D_obs_counts_val=D_obs_counts_val,
            D_obs_counts_test=D_obs_counts_test, qc_obs_counts_train=qc_obs_counts_train,
            qc_obs_counts_val=qc_obs_counts_val, qc_obs_counts_test=qc_obs_counts_test,
            predictor=predictor, x_var_info=x_var_info,
            **config.get("lightning_module_kwargs", dict()),
            use_scheduler=use_scheduler,
        )
        if not HAVE_LIGHTNING:
            self.hparams = {"config": config, "x_var_info": x_var_info, "use_scheduler": use_scheduler}

    @staticmethod
    def _models():
        from . import models

        return models

    def _init_model(self):
        return getattr(self._models(), self.config["model"])(**self.config.get("model_kwargs", dict()))

    def _init_guide(self):
        return getattr(self._models(), self.config["guide"])(**self.config.get("guide_kwargs", dict()))

    def _init_loss_module(self, model, guide):
        kwargs = dict(self.config.get("loss_module_kwargs", dict()))
        kwargs["model"] = model
        kwargs["guide"] = guide
        dm = self.config["data_module_kwargs"]  # :236
        kwargs["pos_weight"] = round(dm["n_qc_pass"] / (dm["n_qc_pass"] + dm["n_qc_fail"]), 3)
        return getattr(self._models(), self.config["loss_module"])(**kwargs)

    def _init_predictor(self, model, guide):
        return getattr(self._models(), self.config["predictor"])(model, guide)

    def get_data_module(self):
        """:244-247.  Dataset-specific data modules (scanpy / rapids pipelines) are outside this
        build; register a namespace that provides them."""
        ns = type(self).data_namespace
        if ns is None:
            from . import data as own

            if hasattr(own, self.config["data_module"]):  # in-memory / synthetic data modules of this build
                ns = own
        if ns is None:
            raise NotImplementedError(
                "no data-module namespace registered: set TrainConfigPerturbationLightningModule.data_namespace to an "
                "object exposing the data-module classes named in config['data_module'] (e.g. the reference's "
                "cradle_vae.data)")
        return getattr(ns, self.config["data_module"])(**self.config.get("data_module_kwargs", dict()))


# ---------------------------------------------------------------------------------------------
# Checkpoints in Lightning's layout ({"state_dict", "hyper_parameters": {"config", ...}}; keys
# "loss_module.model.*", "loss_module.guide.*", "predictor.model.*", "predictor.guide.*", obs-count
# buffers), so that files interchange with the reference's (train_norman.py:106-109, eval.py:313-322).
# ---------------------------------------------------------------------------------------------
def _plain_config(config: Dict) -> Dict:
    """Copy of ``config`` with the normalisation-statistics dataclass replaced by a dict of
    tensors, so that the file loads with ``torch.load(weights_only=True)``."""
    import copy
    import dataclasses

    # plain dicts only (preprocess_config hands out a defaultdict, which a weights_only load refuses)
    cfg = {k: (dict(v) if isinstance(v, dict) else copy.copy(v)) for k, v in config.items()}
    gk = dict(cfg.get("guide_kwargs", {}))
    stats = gk.get("x_normalization_stats")
    if dataclasses.is_dataclass(stats):
        gk["x_normalization_stats"] = {"__stats__": True, **{
            f.name: getattr(stats, f.name) for f in dataclasses.fields(stats) if getattr(stats, f.name) is not None}}
    cfg["guide_kwargs"] = gk
    return cfg


def _restore_config(config: Dict) -> Dict:
    from .guide import ObservationNormalizationStatistics

    cfg = dict(config)
    gk = dict(cfg.get("guide_kwargs", {}))
    stats = gk.get("x_normalization_stats")
    if isinstance(stats, dict):
        gk["x_normalization_stats"] = ObservationNormalizationStatistics(
            **{k: v for k, v in stats.items() if k != "__stats__"})
    cfg["guide_kwargs"] = gk
    return cfg


def save_checkpoint(module: "TrainConfigPerturbationLightningModule", path: str, **extra) -> None:
    ckpt = {
        "state_dict": {k: v.detach().cpu().clone() for k, v in module.state_dict().items()},
        "hyper_parameters": {"config": _plain_config(module.config), "use_scheduler": bool(module.use_scheduler)},
    }
    ckpt.update(extra)
    torch.save(ckpt, path)


def load_checkpoint(path: str, map_location=None) -> "TrainConfigPerturbationLightningModule":
    """Rebuilds the module from a checkpoint written by ``save_checkpoint``.  Always loaded with
    ``weights_only=True``: nothing in the file is executed.  A checkpoint Lightning wrote for the reference
    pickles its config objects (a dataclass of tensors, a pandas Index) and is refused by that loader; convert
    it once, offline and explicitly, in an environment that trusts the file -- read it there, keep
    ``state_dict`` and the plain config, and write them with ``save_checkpoint``'s layout."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    hp = ckpt["hyper_parameters"]
    sd = ckpt["state_dict"]
    config = _restore_config(hp["config"])

    def buf(name):
        return sd.get(name)

    module = TrainConfigPerturbationLightningModule(
        config=config,
        D_obs_counts_train=buf("D_obs_counts_train"), D_obs_counts_val=buf("D_obs_counts_val"),
        D_obs_counts_test=buf("D_obs_counts_test"), qc_obs_counts_train=buf("qc_obs_counts_train"),
        qc_obs_counts_val=buf("qc_obs_counts_val"), qc_obs_counts_test=buf("qc_obs_counts_test"),
        x_var_info=hp.get("x_var_info"), use_scheduler=hp.get("use_scheduler", False))
    module.load_state_dict(sd)
    if map_location is not None:
        module = module.to(map_location)
    return module


if not HAVE_LIGHTNING:
    TrainConfigPerturbationLightningModule.load_from_checkpoint = classmethod(
        lambda cls, checkpoint_path, map_location=None, **kw: load_checkpoint(checkpoint_path, map_location))
