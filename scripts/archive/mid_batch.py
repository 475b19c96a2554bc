"""Fused pairs path, batch sizes around the number of compute units (70 x 70)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from ndtpso_slam_amd import capi, synth
p = synth.make_pairs(64, seed=0)
ctx = capi.Context(0)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
grid = capi.Grid(60, 60, 0.5)
cfg = capi.PSOConfig.make(70, 70)
for B in (64, 128, 192, 256, 257, 320, 384, 512, 1024):
    sel = np.arange(B) % 64
    run = lambda: ctx.align_pairs(p.ref_ranges[sel], p.new_ranges[sel], geom, grid, (0, 0, 0), (.1, .1, 3.1415e-3), cfg, seeds=p.seeds[sel])
    run()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); out = run(); ts.append(time.perf_counter() - t)
    print(f"batch {B:4d}: {1e3 * np.median(ts):.3f} ms  {B / np.median(ts):.0f} align/s")
