"""The inequality the exact score mode rests on, checked for every evaluation with the diagnostic build
-DNDTPSO_VERIFY_MARGIN: |fp32 score - fp64 score| <= derived bound B < half the arbitration margin.  Runs in a
subprocess (the diagnostic library is selected per process with NDTPSO_LIB)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "verify_margin.py"), *args], capture_output=True, text=True,
                       timeout=1500, env={k: v for k, v in os.environ.items() if k != "NDTPSO_LIB"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("args", [("--workload", "config3", "--pairs", "512"), ("--workload", "config5", "--pairs", "130"),
                                  ("--workload", "random"), ("--workload", "converged")])
def test_fp32_score_error_stays_below_half_the_arbitration_margin(args):
    d = _run(*args)
    assert d["evaluations_checked"] > 1e4
    # the derived bound really bounds the measured error, in every evaluation
    assert d["max_err_over_bound"] <= 1.0, d
    # ... and stays below half the margin: decisions outside the margin are the fp64 mode's by proof, not by luck
    assert d["max_bound_over_half_tau"] < 1.0, d
    assert d["max_err_over_half_tau"] < 0.25, d
    # the folded binning of the fp32 loop agrees with the reference's on every point of every evaluation
    assert d["points_binned_differently"] == 0, d
