"""Loader for tests/golden/*.npz (fixtures written by oracle/make_golden.py)."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def elbo_cases():
    return sorted(os.path.basename(p)[len("elbo_"):-4] for p in glob.glob(os.path.join(GOLDEN, "elbo_*.npz")))


def _group(z, prefix, dtype=None, device="cpu"):
    out = {}
    for k in z.files:
        if k.startswith(prefix + "/"):
            t = torch.from_numpy(z[k])
            if dtype is not None and t.is_floating_point():
                t = t.to(dtype)
            out[k[len(prefix) + 1:]] = t.to(device)
    return out


def load_elbo(name, dtype=torch.float32, device="cpu"):
    z = np.load(os.path.join(GOLDEN, f"elbo_{name}.npz"), allow_pickle=False)
    case = dict(cfg=json.loads(str(z["cfg"])))
    for grp in ("batch", "params", "noise", "cf_noise"):
        case[grp] = _group(z, grp, dtype, device)
    for tag in ("f32", "f64"):
        case[tag] = dict(
            loss=float(z[f"{tag}/loss"]),
            metrics={k: float(v) for k, v in _group(z, f"{tag}/metrics").items()},
            grads=_group(z, f"{tag}/grads"),
            samples=_group(z, f"{tag}/samples"),  # the reference guide's own draws: mask, E, qc_E, z_basal
        )
    return case


def load_predictor(name, dtype=torch.float32, device="cpu"):
    z = np.load(os.path.join(GOLDEN, f"predictor_{name}.npz"), allow_pickle=False)
    case = dict(cfg=json.loads(str(z["cfg"])))
    for grp in ("batch", "params"):
        case[grp] = _group(z, grp, dtype, device)
    case["ate"] = _group(z, "ate", dtype, device)
    case["iwelbo"] = _group(z, "iwelbo", dtype, device)
    case["samples"] = _group(z, "samples", dtype, device)
    return case
