"""Which batch seeds of a full-size parity case are free of LeakyReLU inputs within fp32 rounding of zero (see the
docstring of tests/test_fused_fullsize_gpu.py): python tools/seed_scan.py <shape> <kmax> <seed> [<seed> ...]
CRADLE_SCAN_PARAM_SEEDS=a,b,c scans the parameter seed at the first batch seed instead (with a full treatment axis the
embedding encoder's inputs do not depend on the batch: a kink there follows the parameters, not the batch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cradle_vae_amd import synthetic as syn  # noqa: E402
from tests.test_fused_fullsize_gpu import compare_with_oracle  # noqa: E402

shape, kmax = syn.SHAPES[sys.argv[1]], int(sys.argv[2])
pseeds = [int(v) for v in os.environ.get("CRADLE_SCAN_PARAM_SEEDS", "23").split(",")]
for seed in map(int, sys.argv[3:]):
    for ps in pseeds:
        try:
            worst = compare_with_oracle(shape, kmax, seed, compact=shape.n_cells * kmax < shape.n_treatments,
                                        precisions=("f32",), param_seed=ps)
            print(f"{shape.name} batch seed {seed} parameter seed {ps}: ok {worst}", flush=True)
        except AssertionError as e:
            print(f"{shape.name} batch seed {seed} parameter seed {ps}: FAILED {e}", flush=True)
