"""BASELINE config 5 (2048 particles x 200 iterations, 2048 beams, 0.25 m cells): one pair / small batches, one
workgroup per pair vs clusters (the swarm lives in an HBM workspace per workgroup)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from ndtpso_slam_amd import capi, synth
p = synth.make_pairs(8, n_beams=2048, seed=21)
ctx = capi.Context(0)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
grid = capi.Grid(60, 60, 0.25)
cfg = capi.PSOConfig.make(200, 2048)
for B in (1, 4, 8):
    sel = np.arange(B)
    res = {}
    for K in ("0", None) + tuple(os.environ.get("C5_KS", "").split()):
        if K is None: os.environ.pop("NDTPSO_CLUSTER", None)
        else: os.environ["NDTPSO_CLUSTER"] = K
        t = time.perf_counter()
        out = ctx.align_pairs(p.ref_ranges[sel], p.new_ranges[sel], geom, grid, (0, 0, 0), (.1, .1, 3.1415e-3), cfg, seeds=p.seeds[sel])
        res[K] = (out, time.perf_counter() - t)
    a, b = res["0"][0], res[None][0]
    for K in res:
        if K not in ("0", None):
            print(f"   K={K}: {1e3 * res[K][1]:.1f} ms rounds {res[K][0][2]['rounds'][0]} identical {np.array_equal(res[K][0][0], a[0])}")
    print(f"config 5, {B} pair(s): identical {np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])}  one WG each {1e3 * res['0'][1]:.1f} ms  clustered {1e3 * res[None][1]:.1f} ms  rounds {a[2]['rounds'][0]} -> {b[2]['rounds'][0]} status {b[2]['status']}")
