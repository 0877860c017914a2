"""Runs the reference's own ``_get_cfX`` (cradle_vae/data/norman/data_module.py:175-191) on a small synthetic
AnnData stand-in and writes inputs, the peer draws it made and its outputs to tests/golden/cfx_tiny.npz.
TEST INFRASTRUCTURE, build container only.  The data module cannot be imported here (scanpy, anndata, rapids), so
the method's definition is taken out of the file's syntax tree and executed as it stands; ``X.cuda()`` (no GPU in
this container) is made the identity and ``np.random.choice`` is wrapped to record what it returned."""
import ast
import os
import sys

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh  # noqa: E402

REF = rh.REFERENCE_ROOT
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cfx_tiny.npz")


def reference_get_cfx():
    tree = ast.parse(open(os.path.join(REF, "cradle_vae", "data", "norman", "data_module.py")).read())
    fn = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "_get_cfX":
            fn = node
    ns = dict(np=np, torch=torch, pd=pd, print=lambda *a, **k: None)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "data_module.py", "exec"), ns)
    return ns["_get_cfX"]


def main():
    rh._install_stubs()
    import anndata  # the harness's stand-in

    rng = np.random.default_rng(3)
    n, G, n_t = 260, 9, 5
    X = torch.from_numpy(rng.poisson(4.0, size=(n, G)).astype(np.float32))
    treatment = rng.integers(0, n_t, size=n)
    treatment[:70] = 0  # one treatment with more than 50 QC-failing cells in the train split
    split = np.where(rng.random(n) < 0.7, "train", np.where(rng.random(n) < 0.5, "val", "test"))
    total_qc = (rng.random(n) < 0.35).astype(int)
    total_qc[treatment == 4] = 0  # a treatment without QC-failing cells -> zero rows
    total_qc[:70] = 1
    split[:70] = "train"
    obs = pd.DataFrame({"treatment": [f"t{t}" for t in treatment], "split": split, "total_qc": total_qc},
                       index=[f"c{i}" for i in range(n)])
    adata = anndata.AnnData(X=X.numpy(), obs=obs)
    get_cfx = reference_get_cfx()
    draws = []
    real_choice = np.random.choice

    def recording_choice(a, size=None, *args, **kw):
        out = real_choice(a, size, *args, **kw)
        draws.append(np.asarray(out, dtype=np.int64))
        return out

    out = dict(X=X.numpy(), treatment=treatment.astype(np.int64), split=split.astype("U5"), total_qc=total_qc.astype(np.int64))
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    np.random.choice = recording_choice
    try:
        np.random.seed(11)
        for sp in ("train", "val", "test"):
            draws.clear()
            cfx = get_cfx(None, adata, X, sp)
            out[f"cfX/{sp}"] = cfx.numpy()
            k = max(len(d) for d in draws)
            padded = np.full((len(draws), k), -1, dtype=np.int64)
            for i, d in enumerate(draws):
                padded[i, :len(d)] = d
            out[f"draws/{sp}"] = padded  # per cell of the split, in order: the values np.random.choice returned
    finally:
        np.random.choice = real_choice
        torch.Tensor.cuda = real_cuda
    np.savez_compressed(OUT, **out)
    print(OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
