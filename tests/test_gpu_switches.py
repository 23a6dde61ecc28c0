"""The switches that stay in the library (ten; round 3 removed the rejected experiments) must keep working:
each setting runs the numeric path in a process of its own (the library reads its environment once) on a single matrix
with a blocked root (1 and 300 right-hand sides, fused step) and on an interleaved Cholesky batch."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETTINGS = [
    {},
    {"CS3_NO_OVERLAP": "1"},                                        # fused step without the overlapped forward sweep
    {"CS3_NO_GEMM_SWEEPS": "1"},                                    # many right-hand sides by substitution only
    {"CS3_NO_GRAPH": "1"},                                          # eager launches instead of the captured graphs
    {"CS3_SUBTREE": "0"},                                           # level schedule for the whole tree
    {"CS3_SUB_HEIGHT": "64"},                                       # whole subtrees in the forest's tasks
    {"CS3_RELAX_Z": "0.2", "CS3_SUB_HEIGHT": "2"},                  # less amalgamation, shallow tasks
    {"CS3_WG_MIN_BATCH": "100000", "CS3_IL_MIN_BATCH": "100000"},   # batches without the lane = matrix / one-workgroup kernels
]


@pytest.mark.gpu
@pytest.mark.parametrize("setting", SETTINGS, ids=lambda s: ",".join("%s=%s" % kv for kv in s.items()) or "defaults")
def test_switch_setting_keeps_the_path_correct(gpu, setting):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **setting)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_worker.py")], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (setting, p.stdout[-500:], p.stderr[-1500:])
