#!/usr/bin/env python3
"""Evaluation entry point with the reference's flags (eval.py:325-369): test-set IWELBO (100 particles), model
average treatment effects against the data's, per split, for a local experiment directory (results/<name>:
its best checkpoint, metrics written to test_metrics.csv beside it, eval.py:206-243) or a single checkpoint.

    python eval.py --experiment_path results/cradle_vae_synthetic [--batch_size 128] [--ate_n_particles 2500]

Not here: evaluating a wandb run (``--wandb``: no wandb in this build) and the qc_ratio stage (rapids_singlecell /
scrublet on sampled cells, eval.py:119-127).  ``--perturbseq/--qc_pass`` take true/false (the reference declares them
``type=bool``, which reads every non-empty string as True); ``--devices`` takes integers."""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _flag(v):
    return str(v).lower() in ("1", "true", "yes", "y")


def best_checkpoint(experiment_path):
    if os.path.isfile(experiment_path):
        return experiment_path
    summary = os.path.join(experiment_path, "summary.csv")
    if os.path.exists(summary):
        with open(summary) as f:
            row = next(csv.DictReader(f))
        if row.get("best_checkpoint_path") and os.path.exists(row["best_checkpoint_path"]):
            return row["best_checkpoint_path"]
    found = sorted(glob.glob(os.path.join(experiment_path, "checkpoints", "best-*.ckpt")))
    if not found:
        raise SystemExit(f"no best-*.ckpt under {experiment_path}/checkpoints")
    return found[-1]


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--experiment_path", required=True)
    parser.add_argument("--wandb", type=_flag, default=False)
    parser.add_argument("--perturbseq", type=_flag, default=True)
    parser.add_argument("--batch_size", type=int, default=128)
    parser.add_argument("--ate_n_particles", type=int, default=2500)
    parser.add_argument("--qc_pass", type=_flag, default=True)
    parser.add_argument("--devices", type=int, nargs="*", default=[0])
    parser.add_argument("--thr", type=int, default=3)
    args = parser.parse_args(argv)
    if args.wandb:
        raise SystemExit("--wandb: evaluating a wandb run needs wandb, which is not part of this build; "
                         "pass a local experiment directory or checkpoint")
    from cradle_vae_amd.evaluate import evaluate_checkpoint

    method = "perturbseq" if args.perturbseq else "mean"
    ckpt = best_checkpoint(args.experiment_path)
    metrics = evaluate_checkpoint(ckpt, method, batch_size=args.batch_size, ate_n_particles=args.ate_n_particles,
                                  qc_pass=args.qc_pass, thr=args.thr, devices=args.devices)
    metrics = {k: (float(v) if hasattr(v, "__float__") else v) for k, v in metrics.items()}
    out_dir = args.experiment_path if os.path.isdir(args.experiment_path) else os.path.dirname(ckpt)
    with open(os.path.join(out_dir, "test_metrics.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(list(metrics))
        w.writerow(list(metrics.values()))
    print(json.dumps(metrics, indent=1))
    return metrics


if __name__ == "__main__":
    main()
