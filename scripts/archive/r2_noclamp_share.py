import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from ndtpso_slam_amd import capi, synth
p = synth.make_pairs(512, seed=2024)
ctx = capi.Context(0)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
pose, cost, st = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, 0.5), (0, 0, 0), (.1, .1, 3.1415e-3), capi.PSOConfig.make(70, 70), seeds=p.seeds, mode=capi.SCORE_F32)
print("evals mean %.0f; no-clamp evals mean %.0f min %d; share %.4f" % (st["cost_evals"].mean(), st["gbest_updates"].mean(), st["gbest_updates"].min(), st["gbest_updates"].sum() / st["cost_evals"].sum()))
