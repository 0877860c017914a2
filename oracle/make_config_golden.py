"""Runs the reference's own ``preprocess_config`` / ``add_data_info_to_config`` (train_norman.py:163-227) on its
four demo yamls and writes inputs + outputs to tests/golden/config_demo.json.  TEST INFRASTRUCTURE, build
container only.  train_norman.py cannot be imported as a module here (unconditional ``import cupy`` / ``import
wandb`` / pytorch_lightning), so the two function definitions are taken out of its syntax tree and executed as they
stand, with the names they use bound to the standard library."""
import ast
import json
import os
import sys
from collections import defaultdict
from typing import Any, DefaultDict, Dict

import yaml

REF = os.environ.get("CRADLE_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config_demo.json")


def reference_functions(script="train_norman.py"):
    tree = ast.parse(open(os.path.join(REF, script)).read())
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("preprocess_config", "add_data_info_to_config")]
    ns = dict(defaultdict=defaultdict, Dict=Dict, Any=Any, DefaultDict=DefaultDict, PerturbationDataModule=object,
              wandb=type("wandb", (), {"config": None}), print=lambda *a, **k: None)
    exec(compile(ast.Module(body=wanted, type_ignores=[]), script, "exec"), ns)
    return ns["preprocess_config"], ns["add_data_info_to_config"]


class FakeDataModule:
    """What add_data_info_to_config asks a data module for."""

    class _Frame:
        def __init__(self, n):
            self.shape = (n, 0)

    def get_d_var_info(self):
        return self._Frame(105)

    def get_x_var_info(self):
        return self._Frame(5000)

    def get_qc_var_info(self):
        return self._Frame(1)

    def get_x_train_statistics(self):
        return "STATS"

    def get_qc_n(self):
        return {"n_qc_pass": 900, "n_qc_fail": 100}


def main():
    pre, add = reference_functions()
    cases = {}
    for name in sorted(os.listdir(os.path.join(REF, "demo"))):
        if not name.endswith(".yaml"):
            continue
        raw = yaml.safe_load(open(os.path.join(REF, "demo", name)))
        processed = json.loads(json.dumps(pre(dict(raw))))
        with_data = json.loads(json.dumps(add(pre(dict(raw)), FakeDataModule())))
        cases[name] = dict(input=raw, processed=processed, with_data_info=with_data)
    # the rules the demo files do not exercise: shared keys, shared n_latent, clip -1, deeper dots
    extra = {"a--model_kwargs.x--guide_kwargs.x": 3, "n_latent": 7, "gradient_clip_norm": -1, "g.h.i": 1, "name": "t"}
    cases["synthetic_rules"] = dict(input=extra, processed=json.loads(json.dumps(pre(dict(extra)))))
    with open(OUT, "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)
    print(OUT, {k: len(v["processed"]) for k, v in cases.items()})


if __name__ == "__main__":
    sys.exit(main())
