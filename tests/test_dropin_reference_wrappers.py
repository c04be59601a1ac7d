"""CPU, build container only: the reference's UNMODIFIED wrappers (`models/vivit.py`, `utils/misc.py`) import and
construct on top of THIS package (drop-in contract): same constructor kwargs, sub-module names
and state_dict keys/shapes as when they sit on the reference's own `eventful_transformer`.  Runs in
subprocesses (two different packages share one import name).  Skipped where /root/reference is absent (GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present on this machine")

PROBE = r'''
import json, sys, yaml
sys.dont_write_bytecode = True
sys.path.insert(0, "%(oracle)s")
from _refimport import _install_stubs
_install_stubs()
for p in reversed(%(paths)r):
    sys.path.insert(0, p)
import eventful_transformer.blocks as blocks_mod
from models.vivit import FactorizedViViT                      # reference file, unmodified
from utils.misc import set_policies                           # reference file, unmodified
from eventful_transformer.policies import TokenNormTopK
cfg = yaml.safe_load(open("%(ref)s/configs/models/vivit_b_kinetics400.yml"))["model"]
cfg["spatial_config"]["block_class"] = "EventfulBlock"
cfg["spatial_config"]["block_config"]["matmul_2_cast"] = "bfloat16"
model = FactorizedViViT(**cfg)
set_policies(model, TokenNormTopK, k=128)
gates = [n for n, m in model.named_modules() if type(m).__name__ in ("TokenGate", "TokenDeltaGate")]
model.reset(); model.counting(); model.clear_counts(); model.no_counting()
print(json.dumps({"impl": blocks_mod.__file__,
                  "keys": {k: list(v.shape) for k, v in model.state_dict().items()},
                  "gates": gates,
                  "policies": sum(1 for m in model.modules() if type(m).__name__ == "TokenNormTopK"),
                  "blocks": [type(b).__name__ for b in model.spatial_model.backbone.blocks]}))
'''


def _run(paths):
    code = PROBE % dict(oracle=os.path.join(ROOT, "oracle"), paths=paths, ref=REF)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp", timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_reference_vivit_wrapper_builds_on_this_package():
    ours = _run([os.path.join(ROOT, "eventful-transformer_amd"), REF])
    theirs = _run([REF])
    assert ours["impl"].startswith(ROOT) and theirs["impl"].startswith(REF)
    assert ours["keys"] == theirs["keys"]            # identical state_dict keys and shapes -> checkpoints load
    assert ours["gates"] == theirs["gates"]          # set_policies finds the same gates under the same names
    assert ours["policies"] == theirs["policies"] == len(ours["gates"])
    assert ours["blocks"] == theirs["blocks"] == ["EventfulBlock"] * 12
