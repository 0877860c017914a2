#!/usr/bin/env python3
"""Training entry point with the reference's CLI and yaml semantics (train_norman.py:48-129,270-292; the four
train_*.py of the reference differ only in their default yaml and Replogle's early-stopping patience):

    python train.py --config demo/cradle_vae_synthetic.yaml
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py --config ...   # N GPUs

config -> preprocess_config -> data module -> add_data_info_to_config -> TrainConfigPerturbationLightningModule ->
fit (best-by-val/loss checkpoint, a checkpoint every 2000 steps, early stopping after `patience` epochs without
improvement) -> reload the best checkpoint -> validation IWELBO (100 particles) -> results/<name>/summary.csv.

Differences from running the reference's script: no wandb (``use_wandb: True`` logs locally and says so), no cupy
device selection (``devices[0]`` picks the HIP device, LOCAL_RANK under torch.distributed.run), and the data module
named by ``data_module`` is looked up in ``cradle_vae_amd.data`` or in the module given by ``--data-namespace``
(e.g. the reference's ``cradle_vae.data`` where its scanpy / h5ad dependencies are installed).
"""
import argparse
import csv
import importlib
import json
import os
import sys
from os.path import exists, join

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# N > 1: sixteen hardware queues per process instead of HIP's four, set before the runtime starts (see bench.py)
from cradle_vae_amd.env import set_hw_queues_for_multi_gpu  # noqa: E402  (imports nothing but os)

set_hw_queues_for_multi_gpu()

RESULTS_BASE_DIR = "results/"


def init_experiment_local(config, base=None):
    """train_norman.py:132-143: results/<name>, <name>-2, <name>-3, ... for repeated runs."""
    base = RESULTS_BASE_DIR if base is None else base
    results_dir = join(base, config["name"])
    i = 2
    while exists(results_dir):
        results_dir = join(base, f"{config['name']}-{i}")
        i += 1
    os.makedirs(results_dir)
    return results_dir


def get_data_module(config, namespace=None, **extra):
    """train_norman.py:157-160."""
    from cradle_vae_amd import data as own

    kwargs = dict(config.get("data_module_kwargs", dict()))
    for ns in (namespace, own):
        if ns is not None and hasattr(ns, config["data_module"]):
            cls = getattr(ns, config["data_module"])
            if ns is own:
                kwargs.update(extra)  # device / rank / world_size: this build's data modules live on the GPU
            kwargs.pop("n_qc_pass", None), kwargs.pop("n_qc_fail", None)  # written back by add_data_info_to_config
            return cls(**kwargs)
    raise SystemExit(f"data module {config['data_module']!r} not found: pass --data-namespace <python module that "
                     "defines it> (the reference's data modules need its h5ad files and scanpy)")


class Checkpointing:
    """The two ModelCheckpoint callbacks and the EarlyStopping of train_norman.py:230-267 for trainer.fit."""

    def __init__(self, results_dir, patience=50, every_n_train_steps=2000, rank=0):
        self.dir = join(results_dir, "checkpoints")
        self.patience, self.every, self.rank = patience, every_n_train_steps, rank
        self.best, self.best_model_path, self.bad_epochs = None, None, 0
        self.rows = []
        if rank == 0:
            os.makedirs(self.dir, exist_ok=True)

    def __call__(self, module, epoch, record):
        from cradle_vae_amd.lightning_module import save_checkpoint

        self.rows.append(dict(record))
        val, steps = record.get("val/loss"), record["steps"]
        key = val if val is not None else record["train/loss"]
        if self.best is None or key < self.best:
            self.best, self.bad_epochs = key, 0
            if self.rank == 0:
                if self.best_model_path and exists(self.best_model_path):
                    os.remove(self.best_model_path)  # save_top_k=1
                self.best_model_path = join(self.dir, f"best-epoch={epoch}-step={steps}-val_loss={key:.2f}.ckpt")
                save_checkpoint(module, self.best_model_path)
        else:
            self.bad_epochs += 1
        if self.bad_epochs >= self.patience:
            raise StopIteration  # EarlyStopping(monitor="val/loss", patience=...): the key is the same on every rank

    def on_train_step(self, module, epoch, steps):
        """ModelCheckpoint(every_n_train_steps=2000) of train_norman.py:240-247: fires inside an epoch too."""
        from cradle_vae_amd.lightning_module import save_checkpoint

        if self.rank == 0 and self.every and steps % self.every == 0:
            key = float("nan") if self.best is None else self.best
            save_checkpoint(module, join(self.dir, f"epoch={epoch}-step={steps}-val_loss={key:.2f}.ckpt"))


def train(config, data_namespace=None, results_base=None, patience=50, quiet=False):
    import torch

    from cradle_vae_amd import parallel
    from cradle_vae_amd.config import add_data_info_to_config, preprocess_config
    from cradle_vae_amd.data import DeviceBatchLoader, TensorPerturbationDataset
    from cradle_vae_amd.lightning_module import TrainConfigPerturbationLightningModule, load_checkpoint
    from cradle_vae_amd.trainer import fit

    config = preprocess_config(config)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
    else:
        devs = config.get("devices") or [0]
        local = int(devs[0]) if isinstance(devs, (list, tuple)) else 0
        if local >= torch.cuda.device_count():
            local = 0  # the reference's yamls name the authors' GPU ids
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
    if config.get("use_wandb") and rank == 0 and not quiet:
        print("use_wandb: True -- wandb is not part of this build; logging to the local results directory", flush=True)
    results_dir = init_experiment_local(config, results_base) if rank == 0 else None

    data_module = get_data_module(config, data_namespace, device=device, rank=rank, world_size=world)
    config = add_data_info_to_config(config, data_module)
    if data_namespace is not None:
        TrainConfigPerturbationLightningModule.data_namespace = data_namespace
    torch.manual_seed(config.get("seed", 0))  # pl.seed_everything(config["seed"])
    as_float = lambda t: t.float() if torch.is_tensor(t) else t
    module = TrainConfigPerturbationLightningModule(
        config=config,
        D_obs_counts_train=as_float(data_module.get_train_perturbation_obs_counts()),
        D_obs_counts_val=as_float(data_module.get_val_perturbation_obs_counts()),
        D_obs_counts_test=as_float(data_module.get_test_perturbation_obs_counts()),
        qc_obs_counts_train=as_float(data_module.get_train_qc_obs_counts()),
        qc_obs_counts_val=as_float(data_module.get_val_qc_obs_counts()),
        qc_obs_counts_test=as_float(data_module.get_test_qc_obs_counts()),
        x_var_info=None, use_scheduler=False).to(device)
    ckpt = Checkpointing(results_dir or "/tmp", patience=patience, rank=rank)
    max_steps = config.get("max_steps", -1)
    hist = fit(module, data_module.train_dataloader(), data_module.val_dataloader(),
               max_epochs=config.get("max_epochs") or 1000, max_steps=None if max_steps in (None, -1) else max_steps,
               gradient_clip_val=config.get("gradient_clip_norm"), seed=config.get("seed", 0), callbacks=(ckpt,),
               step_callbacks=(ckpt.on_train_step,))
    val_iwelbo, best = None, ckpt.best_model_path
    if world > 1:
        # the evaluation of the best checkpoint below is rank 0's alone (100-particle IWELBO over the validation set:
        # minutes on a large one).  The other ranks have nothing left to exchange: the group is dissolved here, after
        # one last barrier, rather than left waiting in a collective whose watchdog would fire
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        with open(join(results_dir, "metrics.csv"), "w", newline="") as f:  # CSVLogger's file
            keys = sorted({k for r in hist for k in r})
            w = csv.DictWriter(f, fieldnames=keys)
            w.writeheader()
            w.writerows(hist)
        module = load_checkpoint(best, map_location=device)  # train_norman.py:106-109
        ds = data_module.val_dataloader().dataset
        bs = max(1, int(getattr(data_module, "batch_size", 512) / 4))  # :112
        loader = DeviceBatchLoader(ds, bs, device=device) if isinstance(ds, TensorPerturbationDataset) else \
            torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=False)
        if module.predictor is not None:
            val_iwelbo = float(module.predictor.compute_predictive_iwelbo(loader, n_particles=100)["IWELBO"].mean())
        with open(join(results_dir, "summary.csv"), "w", newline="") as f:  # :124-127
            w = csv.writer(f)
            w.writerow(["val/IWELBO", "best_checkpoint_path"])
            w.writerow([val_iwelbo, best])
        if not quiet:
            print(json.dumps({"results_dir": results_dir, "epochs": len(hist), "best_checkpoint_path": best,
                              "val/loss": ckpt.best, "val/IWELBO": val_iwelbo}), flush=True)
    return val_iwelbo, results_dir


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, default="./demo/cradle_vae_synthetic.yaml",
                        help="experiment configuration yaml path (the reference's format)")
    parser.add_argument("--data-namespace", default=None, help="python module that defines the data module class")
    parser.add_argument("--results-dir", default=None, help="base directory for results (default results/)")
    parser.add_argument("--patience", type=int, default=50, help="early-stopping patience (100 in train_replogle.py)")
    parser.add_argument("--max_epochs", type=int, default=None, help="override the yaml's max_epochs")
    args, _unknown = parser.parse_known_args(argv)  # the reference ignores unknown arguments too (:281)
    from cradle_vae_amd.config import load_config

    config = load_config(args.config)
    if args.max_epochs is not None:
        config["max_epochs"] = args.max_epochs
    ns = importlib.import_module(args.data_namespace) if args.data_namespace else None
    train(config, ns, args.results_dir, args.patience)


if __name__ == "__main__":
    main()
