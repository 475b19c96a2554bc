"""ndtpso_align on a staged table: LDS paths vs the table read from HBM (paths 4 / 5), wall time per call."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from ndtpso_slam_amd import capi, synth
p = synth.make_pairs(2, seed=0)
ctx = capi.Context(0)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
grid = capi.Grid(60, 60, 0.5)
xy = ctx.scan_to_points(p.new_ranges[0], geom)
ctx.ref_from_scan(grid, p.ref_ranges[0], geom)
for P, I in ((30, 50), (70, 70)):
    cfg = capi.PSOConfig.make(I, P)
    for path in (None, "1", "5"):
        if path is None: os.environ.pop("NDTPSO_PATH", None)
        else: os.environ["NDTPSO_PATH"] = path
        for mode in (capi.SCORE_F32, capi.SCORE_F64):
            ctx.align(xy, (0, 0, 0), (.1, .1, .003), cfg, seed=3, mode=mode)
            t = time.perf_counter()
            for _ in range(20):
                pose, cost, st = ctx.align(xy, (0, 0, 0), (.1, .1, .003), cfg, seed=3, mode=mode)
            dt = (time.perf_counter() - t) / 20
            print(f"{P}x{I} path {path or 'auto'} mode {'f32' if mode == 0 else 'f64'}: {dt*1e3:.3f} ms  pose {pose}")
