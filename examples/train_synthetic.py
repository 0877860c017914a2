#!/usr/bin/env python3
"""End-to-end run on synthetic Perturb-seq-like data: the flow of the reference's train_norman.py
(config -> data module -> TrainConfigPerturbationLightningModule -> fit -> checkpoint) followed by
eval.py's evaluation, on the MI355X HIP path.

    python examples/train_synthetic.py [--epochs 20] [--cells 8192] [--genes 400] [--guides 12]
    python -m torch.distributed.run --nproc-per-node N examples/train_synthetic.py   # N GPUs
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cradle_vae_amd.env import set_hw_queues_for_multi_gpu  # noqa: E402  (sixteen hardware queues for N > 1, see bench.py)

set_hw_queues_for_multi_gpu()

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--cells", type=int, default=8192)
    ap.add_argument("--genes", type=int, default=400)
    ap.add_argument("--guides", type=int, default=12)
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--ate-particles", type=int, default=512)
    ap.add_argument("--out", default="/tmp/cradle_synthetic.ckpt")
    args = ap.parse_args()

    from cradle_vae_amd import parallel
    from cradle_vae_amd.data import SyntheticPerturbationDataModule
    from cradle_vae_amd.evaluate import evaluate_module
    from cradle_vae_amd.lightning_module import TrainConfigPerturbationLightningModule, save_checkpoint
    from cradle_vae_amd.trainer import fit

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
    dm_kwargs = dict(n_cells=args.cells, n_genes=args.genes, n_guides=args.guides, batch_size=args.batch_size, seed=0)
    dm = SyntheticPerturbationDataModule(device=device, rank=rank, world_size=world, **dm_kwargs)
    config = dict(
        model="cradle_vae_Model", guide="cradle_vae_CorrelatedNormalGuide", loss_module="cradle_vae_ELBOLossModule",
        predictor="cradle_vae_Predictor", data_module="SyntheticPerturbationDataModule",
        data_module_kwargs=dict(dm_kwargs, **dm.get_qc_n()),
        lightning_module_kwargs=dict(lr=1e-3, n_particles=5), loss_module_kwargs=dict(beta=0.5),
        model_kwargs=dict(n_latent=32, n_treatments=args.guides, n_phenos=args.genes, n_qc=1, mask_prior_prob=0.05,
                          embedding_prior_scale=1.0, likelihood_key="library_nb", decoder_n_layers=1,
                          decoder_n_hidden=128, sum_mode=False),
        guide_kwargs=dict(n_latent=32, n_treatments=args.guides, n_phenos=args.genes, n_qc=1, basal_encoder_n_layers=1,
                          basal_encoder_n_hidden=128, basal_encoder_input_normalization="log_standardize",
                          embedding_encoder_n_layers=4, embedding_encoder_n_hidden=128,
                          x_normalization_stats=dm.get_x_train_statistics()))
    torch.manual_seed(0)
    module = TrainConfigPerturbationLightningModule(
        config=config,
        D_obs_counts_train=dm.get_train_perturbation_obs_counts().float(), D_obs_counts_val=dm.get_val_perturbation_obs_counts().float(),
        D_obs_counts_test=dm.get_test_perturbation_obs_counts().float(), qc_obs_counts_train=dm.get_train_qc_obs_counts().float(),
        qc_obs_counts_val=dm.get_val_qc_obs_counts().float(), qc_obs_counts_test=dm.get_test_qc_obs_counts().float(),
        x_var_info=None).to(device)
    hist = fit(module, dm.train_dataloader(), dm.val_dataloader(), max_epochs=args.epochs, gradient_clip_val=100.0,
               log_every=0)
    if rank == 0:
        for h in hist[:: max(1, len(hist) // 5)]:
            print(json.dumps(h))
        save_checkpoint(module, args.out)
        metrics = evaluate_module(module, dm, "perturbseq", batch_size=args.batch_size, ate_n_particles=args.ate_particles)
        print(json.dumps({k: (float(v) if hasattr(v, "__float__") else v) for k, v in metrics.items()}, indent=1))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
