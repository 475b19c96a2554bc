"""BASELINE config 5 under the profiler: 256 pairs of 2048 particles x 200 iterations, 2048 beams, 0.25 m cells in one launch
(one workgroup per CU, swarm in an HBM workspace), and one clustered pair (32 workgroups) -- a command for rocprofv3.
usage: python scripts/config5_profile.py [pairs] [mode exact|f32|f64] [launches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ndtpso_slam_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = {"exact": capi.SCORE_EXACT, "f32": capi.SCORE_F32, "f64": capi.SCORE_F64}[sys.argv[2] if len(sys.argv) > 2 else "exact"]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
p = synth.make_pairs(B, n_beams=2048, seed=21)
ctx = capi.Context(0)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
for k in range(n):
    t = time.perf_counter()
    pose, cost, st = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, 0.25), (0, 0, 0), (.1, .1, 3.1415e-3),
                                     capi.PSOConfig.make(200, 2048), seeds=p.seeds, mode=mode)
    dt = time.perf_counter() - t
    print("config 5: %d pairs in %.1f ms -> %.0f align/s (host buffers), evals/alignment %.0f, arbitrated %.1f, status %s"
          % (B, 1e3 * dt, B / dt, st["cost_evals"].mean(), st["arbitrated"].mean(), np.unique(st["status"])))
