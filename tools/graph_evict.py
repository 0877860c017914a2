"""Forces graph-cache evictions (more than 8 distinct descriptors).  CRADLE_GRAPH_DEBUG=1 python tools/graph_evict.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import golden_io as gio
from tests.test_modular_gpu import build_loss_module
case = gio.load_elbo("mid", torch.float32, "cuda:0")
lm = build_loss_module(case)
fs = lm._fused_step()
b, cfg = case["batch"], case["cfg"]
for i in range(14):
    X = b["X"].clone() if i % 2 else torch.cat([b["X"], b["X"]])[: b["X"].shape[0]].contiguous()
    keep = [torch.empty(1000 * (i + 1), device="cuda:0")]
    X = torch.empty_like(b["X"]).copy_(b["X"])
    out = fs.run(X=X, D=b["D"], D_obs_counts=b["D_obs_counts"], qc=b["qc"], qc_obs_counts=b["qc_obs_counts"], cfX=b["cfX"],
                 library_size=b["library_size"], n_particles=cfg["n_particles"], noise=case["noise"], cf_noise=case["cf_noise"])
    torch.cuda.synchronize()
    print(i, float(out[0]), fs._recaptures, X.data_ptr(), flush=True)
    fs.stage_inputs = False
    hold = getattr(sys.modules[__name__], "hold", [])
    hold.append(X); sys.modules[__name__].hold = hold
print("done")
