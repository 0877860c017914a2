"""The fused step in its three GEMM precisions (f32; the optional bf16x3 and bf16 modes) at three shapes, each with the loss of its batch against the fp64 CPU oracle
(GPU only): python tools/bench_precisions.py"""
import sys, json, types
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
args = types.SimpleNamespace(steps=50, no_cpu_baseline=False)
for name in ("norman", "replogle", "adamson"):
    row = {"shape": name}
    for precision in ("f32", "bf16x3", "bf16"):
        r = bench.secondary_shape(name, "cuda:0", args, precision=precision)
        row[precision + "_ms"], row[precision + "_cells_s"] = round(r["ms_per_step"], 4), round(r["value"])
        if precision != "f32":
            row[precision + "_loss_rel_err_vs_oracle"] = r["loss_rel_err_vs_oracle"]
            row[precision + "_terms_max_rel_err_vs_oracle"] = r["loss_terms_max_rel_err_vs_oracle"]
    print(json.dumps(row), flush=True)
