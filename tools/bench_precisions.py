"""The fused step in both GEMM precisions at three shapes, each with the loss of its batch against the fp64 CPU oracle
(GPU only): python tools/bench_precisions.py"""
import sys, json, types
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
args = types.SimpleNamespace(steps=50, no_cpu_baseline=False)
for name in ("norman", "replogle", "adamson"):
    f = bench.secondary_shape(name, "cuda:0", args, precision="f32")
    b = bench.secondary_shape(name, "cuda:0", args, precision="bf16x3")
    print(json.dumps({"shape": name, "f32_ms": f["ms_per_step"], "f32_cells_s": f["value"], "bf16x3_ms": b["ms_per_step"], "bf16x3_cells_s": b["value"],
                      "bf16x3_loss_rel_err_vs_oracle": b["loss_rel_err_vs_oracle"], "bf16x3_terms": b["loss_terms_max_rel_err_vs_oracle"]}), flush=True)
