"""LABBOOK R6.8: waves that feed matrix-core instructions straight from LDS reads can make a packed-fp32 FMA of ANOTHER kernel on the same SIMD lose a term (found through the
opt-in matrix-core shadow MLP, csrc/mlp_mc.hip; reproducer scripts/ubench/pkfma_beside_mfma.hip).  The product's own matrix-core kernels -- the LPIPS trunk -- were measured not to
do this; this test keeps that measured: one `Model` training iteration (every loss term) repeated while the trunk runs back to back on a side stream of the same process must
stay BITWISE the iteration made alone.  (With `mlp_mc.hip` as the neighbour two runs in three differ: profiles/r06_coresidency/model_iteration.txt.)"""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_model_iteration_is_bitwise_beside_a_running_lpips_trunk(capsys):
    env = {k: v for k, v in os.environ.items() if k != "GOM_MLP_MATRIX_CORES"}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "coresidency_product.py"), "lpips", "120"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"x 120: (\d+) runs differ", out.stdout)
    assert m, out.stdout[-1000:]
    with capsys.disabled():
        print("\n[co-residency] " + out.stdout.strip().splitlines()[-1])
    assert int(m.group(1)) == 0, out.stdout[-3000:]
