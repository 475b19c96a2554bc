import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/scripts") else ".")
import numpy as np
from ndtpso_slam_amd import capi, synth
p = synth.make_pairs(512, seed=2024)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
ctx = capi.Context(0)
for _ in range(3):
    pose, cost, st = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, 0.5), (0, 0, 0), (0.1, 0.1, 3.1415e-3), capi.PSOConfig.make(70, 70), seeds=p.seeds, mode=capi.SCORE_EXACT)
a = st["arbitrated"].astype(float); t = st["rounds"] / 100.0
sel = a > 0
print("arbitration time per alignment (us): total/ events: %.1f us per event; per-alignment examples:" % (t[sel].sum() / a[sel].sum()))
for k in np.argsort(-a)[:8]: print("   events %d  time %.0f us  -> %.1f us/event" % (a[k], t[k], t[k] / a[k]))
