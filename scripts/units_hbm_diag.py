"""Diagnostic for the arbitration's unit form on swarms kept in HBM (tests/test_gpu_fullsize.py::test_unit_form_on_swarms_kept_in_hbm):
runs the test's two batches with NDTPSO_UNITS_HBM / NDTPSO_WAVES from the environment and prints which pairs differ from
the fp64 mode and how.   NDTPSO_UNITS_HBM=16 python scripts/units_hbm_diag.py [case]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndtpso_slam_amd import capi, synth
cases = [(130, 1024, 12, 1081, 0.5), (140, 2048, 5, 2048, 0.25)]
if len(sys.argv) > 1: cases = [cases[int(sys.argv[1])]]
ctx = capi.Context(0)
for (B, Pn, In, beams, cs) in cases:
    p = synth.make_pairs(B, n_beams=beams, seed=300 + In)
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    args = (p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, cs), (0, 0, 0), (0.1, 0.1, 3.1415e-3), capi.PSOConfig.make(In, Pn))
    p64, c64, s64 = ctx.align_pairs(*args, seeds=p.seeds, mode=capi.SCORE_F64)
    for rep in range(2):
        px, cx, sx = ctx.align_pairs(*args, seeds=p.seeds, mode=capi.SCORE_EXACT)
        bad = np.nonzero(~(px == p64).all(axis=1))[0]
        print(json.dumps(dict(case=[B, Pn, In, beams, cs], rep=rep, equal=int(B - len(bad)), pairs=B, costs_equal=bool(np.array_equal(cx, c64)),
                              arbitrated=float(sx["arbitrated"].mean()), flagged=int((sx["status"] != 0).sum()), bad=bad.tolist()[:40],
                              bad_arb=sx["arbitrated"][bad].tolist()[:40], dpose=np.abs(px[bad] - p64[bad]).max(axis=1).tolist()[:10] if len(bad) else [],
                              evals_x=sx["cost_evals"][bad].tolist()[:10], evals_64=s64["cost_evals"][bad].tolist()[:10])))
