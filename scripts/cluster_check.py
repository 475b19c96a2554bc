"""One alignment on several compute units (cluster mode of ndtpso_align) vs one workgroup: identical results, time."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from ndtpso_slam_amd import capi, synth
p = synth.make_pairs(3, seed=0)
ctx = capi.Context(0)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
for cs in (0.5, 0.3):
    grid = capi.Grid(60, 60, cs)
    for b in range(2):
        xy = ctx.scan_to_points(p.new_ranges[b], geom)
        ctx.ref_from_scan(grid, p.ref_ranges[b], geom)
        for P, I in ((30, 50), (70, 70), (5, 8), (200, 20)):
            cfg = capi.PSOConfig.make(I, P)
            for mode in (capi.SCORE_F32, capi.SCORE_F64):
                res = {}
                for K in ("0", "auto", "3", "32"):
                    if K == "auto": os.environ.pop("NDTPSO_CLUSTER", None)
                    else: os.environ["NDTPSO_CLUSTER"] = K
                    ctx.align(xy, (0, 0, 0), (.1, .1, .003), cfg, seed=int(p.seeds[b]), mode=mode)
                    t = time.perf_counter()
                    for _ in range(5):
                        pose, cost, st = ctx.align(xy, (0, 0, 0), (.1, .1, .003), cfg, seed=int(p.seeds[b]), mode=mode)
                    res[K] = (pose, cost, st, (time.perf_counter() - t) / 5)
                base = res["0"]
                ok = all(np.array_equal(r[0], base[0]) and r[1] == base[1] and r[2]["cost_evals"] >= 1 and r[2]["status"] == base[2]["status"] for r in res.values())
                print(f"cs {cs} pair {b} {P}x{I} {'f32' if mode == 0 else 'f64'}: identical {ok}  ms " +
                      " ".join(f"{k}:{1e3 * r[3]:.3f}" for k, r in res.items()) +
                      f"  rounds {base[2]['rounds']}->{res['auto'][2]['rounds']} evals {base[2]['cost_evals']}->{res['auto'][2]['cost_evals']}")
                assert ok
print("cluster check ok")
