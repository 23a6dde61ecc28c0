"""The experiment switches that stay in the library must keep working:
each setting runs the numeric path in a process of its own (the library reads its environment once) on a single matrix
with a blocked root (1 and 300 right-hand sides, fused step) and on an interleaved Cholesky batch."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETTINGS = [
    {},
    {"CS3_SWEEP_FIRST": "1", "CS3_ABSORB": "1"},
    {"CS3_NO_OVERLAP": "1"},
    {"CS3_NO_ROOT_PIPE": "1", "CS3_NBK": "32"},
    {"CS3_NO_GEMM_SWEEPS": "1"},
    {"CS3_NO_FUSED_PERMUTE": "1", "CS3_SPLIT16": "1", "CS3_SOLVE_FORK": "1"},
    {"CS3_RHS_LANES_RMAX": "64"},
    {"CS3_IL_SWEEP_RMAX": "32", "CS3_WG_NB": "32"},
    {"CS3_IL_RMAX": "24", "CS3_BLOCK_NW": "8", "CS3_BATCH_ECONOMY_MIN": "64"},
    {"CS3_WG_MIN_BATCH": "100000", "CS3_IL_MIN_BATCH": "100000"},
    {"CS3_NO_GRAPH": "1"},
    {"CS3_PROMOTE_MAX": "0", "CS3_RIDE_MAX": "0"},
    {"CS3_PROMOTE_MAX": "100000", "CS3_RIDE_MAX": "100000"},      # every small front beside a workgroup group joins it
    {"CS3_LDS_GRID": "0", "CS3_WG_MIN_BATCH": "100000"},            # fronts of order 33..64 through the four-wave shared elimination
    {"CS3_FLAG_SYNC": "1"},                                         # fused step: memory-word hand-over to the side queue instead of graph dependencies
]


@pytest.mark.gpu
@pytest.mark.parametrize("setting", SETTINGS, ids=lambda s: ",".join("%s=%s" % kv for kv in s.items()) or "defaults")
def test_switch_setting_keeps_the_path_correct(gpu, setting):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **setting)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_worker.py")], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (setting, p.stdout[-500:], p.stderr[-1500:])
