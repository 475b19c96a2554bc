"""Cluster shape sweep for one alignment: waves per workgroup x workgroups, rand() table from the host (the node's mode)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from ndtpso_slam_amd import capi, synth
from oracle import pyoracle
p = synth.make_pairs(1, seed=0)
ctx = capi.Context(0)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
grid = capi.Grid(60, 60, 0.5)
xy = ctx.scan_to_points(p.new_ranges[0], geom)
ctx.ref_from_scan(grid, p.ref_ranges[0], geom)
for P, I in ((30, 50), (70, 70)):
    cfg = capi.PSOConfig.make(I, P)
    table = pyoracle.glibc_rand(5, 3 + 3 * P + 6 * P * I)
    for cw in (1, 2, 4, 8, 16):
        for K in (0, 2, 4, 8, 16, 32):
            os.environ["NDTPSO_CLUSTER_WAVES"] = str(cw)
            os.environ["NDTPSO_CLUSTER"] = str(K)
            ctx.align(xy, (0, 0, 0), (.1, .1, .003), cfg, rand_table=table)
            ts = []
            for _ in range(7):
                t = time.perf_counter()
                pose, cost, st = ctx.align(xy, (0, 0, 0), (.1, .1, .003), cfg, rand_table=table)
                ts.append(time.perf_counter() - t)
            print(f"{P}x{I} waves/WG {cw:2d} K {K:2d}: {1e3 * np.median(ts):.3f} ms  rounds {st['rounds']} evals {st['cost_evals']}")
