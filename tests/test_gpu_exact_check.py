"""The start-up known-answer check of NDTPSO_SCORE_EXACT (ndtpso_selftest.inc).

The exact mode's promise -- the fp64 mode's pose and cost bit for bit -- once hung on a compiler flag (a miscompiled
arbitration returned wrong poses silently, ndtpso_slam_amd/build.py).  Now the library checks itself the first time the mode
is asked for: the shipped build must pass, and a build whose arbitration is deliberately off by 2^-44
(-DNDTPSO_BREAK_ARBITRATION) must be refused the mode and still return the fp64 mode's results.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
from ndtpso_slam_amd import capi, synth
ctx = capi.Context(0)
chk = ctx.exact_check()
chk["report"] = ctx.exact_check_report()
p = synth.make_pairs(520, seed=77)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
args = (p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, 0.5), (0, 0, 0), (0.1, 0.1, 3.1415e-3), capi.PSOConfig.make(30, 24))
px, cx, sx = ctx.align_pairs(*args, seeds=p.seeds, mode=capi.SCORE_EXACT)
err = ctx._lib.ndtpso_last_error(ctx._h).decode()
p64, c64, s64 = ctx.align_pairs(*args, seeds=p.seeds, mode=capi.SCORE_F64)
print(json.dumps(dict(check=chk, equal=bool(np.array_equal(px, p64) and np.array_equal(cx, c64)),
                      arbitrated=int(sx["arbitrated"].sum()), err=err)))
"""


def _child(lib=None):
    env = {k: v for k, v in os.environ.items() if k not in ("NDTPSO_LIB", "NDTPSO_EXACT_CHECK")}
    if lib:
        env["NDTPSO_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r.stderr


def test_shipped_library_passes_its_start_up_check():
    d, _ = _child()
    assert d["check"]["state"] == 1, d
    # the check is only worth something if its problem really is arbitrated, in both kernel families
    assert d["check"]["arbitrated_batch"] > 0 and d["check"]["arbitrated_single"] > 0, d
    assert d["check"]["ms"] < 80., d      # once per process and device
    assert d["equal"] and d["arbitrated"] > 0, d
    # every arbitrating kernel instantiation the dispatchers can reach went through the check, each confirmed by the launch it
    # recorded, each with comparisons arbitrated, none differing from the fp64 mode
    fam = d["check"]["report"]["families"]
    assert len(fam) == 16 and len({f["instantiation"] for f in fam}) == 15, fam    # (swarm in HBM under the clipping kernel: the same
    for f in fam:                                                                  #  instantiation, another branch of it)
        assert f["state"] == "passed" and f["launched"] == f["instantiation"] and f["arbitrated"] > 0 and f["mismatched"] == 0, f


def test_a_library_with_a_broken_arbitration_is_refused_the_exact_mode():
    from ndtpso_slam_amd import build
    lib = build.variant_path("broken")
    if not os.path.exists(lib):
        pytest.skip("the -DNDTPSO_BREAK_ARBITRATION variant was not built (python -m ndtpso_slam_amd.build broken)")
    d, stderr = _child(lib)
    assert d["check"]["state"] == 2, d
    assert "refused" in d["err"] and "refused" in stderr, (d, stderr[-500:])
    # ... by EVERY family of the check: each was reached, and each saw costs that differ from the fp64 mode's
    fam = d["check"]["report"]["families"]
    assert len(fam) == 16
    for f in fam:
        assert f["state"] == "refused" and f["launched"] == f["instantiation"] and f["mismatched"] > 0, f
    # ... and what it returns for an exact-mode request is the fp64 mode's result, from the fp64 kernel: nothing arbitrated
    assert d["equal"] and d["arbitrated"] == 0, d
