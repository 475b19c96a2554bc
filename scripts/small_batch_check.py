"""Small batches of the fused pairs path: clustered (idle CUs join in) vs one workgroup per alignment."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from ndtpso_slam_amd import capi, synth
p = synth.make_pairs(64, seed=0)
ctx = capi.Context(0)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
grid = capi.Grid(60, 60, 0.5)
for P, I in ((70, 70), (30, 50)):
    cfg = capi.PSOConfig.make(I, P)
    for mode in (capi.SCORE_F32, capi.SCORE_F64):
        for B in (1, 2, 8, 28, 64, 128):
            sel = np.arange(B) % 64
            res = {}
            for K in ("0", None):
                if K is None: os.environ.pop("NDTPSO_CLUSTER", None)
                else: os.environ["NDTPSO_CLUSTER"] = K
                run = lambda: ctx.align_pairs(p.ref_ranges[sel], p.new_ranges[sel], geom, grid, (0, 0, 0), (.1, .1, 3.1415e-3), cfg, seeds=p.seeds[sel], mode=mode)
                run()
                ts = []
                for _ in range(5):
                    t = time.perf_counter(); out = run(); ts.append(time.perf_counter() - t)
                res[K] = (out, np.median(ts))
            a, b = res["0"][0], res[None][0]
            same = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and (b[2]["status"] == 0).all()
            print(f"{P}x{I} {'f32' if mode == 0 else 'f64'} batch {B:3d}: identical {same}  one WG each {1e3 * res['0'][1]:.3f} ms  clustered {1e3 * res[None][1]:.3f} ms")
            assert same
print("small batch check ok")
