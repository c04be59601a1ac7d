"""The 256-row persistent gated-linear kernel (csrc/evt_linear_big.hip) on shapes its automatic choice would not pick:
every tile configuration is forced through EVT_GEMM_BIG in a fresh process (tests/big_tile_check.py)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tile", [(2, "256x256"), (3, "256x128"), (4, "256x192")])
def test_forced_big_tiles(mode, tile):
    # (EVT_GEMM_SMALL=0: the bit-for-bit comparison launch must run on the 128x128 kernel, not the small-row-count kernel)
    env = dict(os.environ, EVT_GEMM_BIG=str(mode), EVT_GEMM="split", EVT_GEMM_SMALL="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "big_tile_check.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "BIG_TILES_OK" in r.stdout, f"{tile}: {r.stdout[-2000:]}\n{r.stderr[-4000:]}"
