"""Debug helper: replays one sequence of tests/test_gpu_resident.py::test_resident_map_random_operation_sequences up to a
given operation and compares scores / alignments of the resident map with the oracle's in detail.
usage: python tests/campaigns/debug_fuzz_case.py seed frame_w frame_h cs stop_it"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from ndtpso_slam_amd import capi
from oracle import pyoracle as oracle
import np_ref
seed, frame_w, frame_h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cs, stop = float(sys.argv[4]), int(sys.argv[5])
ctx = capi.Context(0)
rng = np.random.default_rng(seed)
grid = capi.Grid(frame_w, frame_h, cs)
rmap = capi.ResidentMap(ctx, grid, og_cell_size=0.0, pool_bytes=32 << 20)
ref = oracle.Frame((0, 0, 0), frame_w, frame_h, cs)
scan = capi.ResidentScan(ctx, 4096)
hw, hh = frame_w / 2, frame_h / 2
cfg, ocfg = capi.PSOConfig.make(6, 9), oracle.PSOConfig.make(6, 9)


def cloud(n):
    xy = np.stack([rng.uniform(-hw * 1.1, hw * 1.1, n), rng.uniform(-hh * 1.1, hh * 1.1, n)], axis=1)
    if n:
        k = rng.integers(0, n, size=max(1, n // 9))
        xy[k] = np.round(xy[k] / cs) * cs
        xy[rng.integers(0, n, size=max(1, n // 5))] *= 0.2
    return xy


for it in range(stop + 1):
    op = rng.choice(["add", "add", "update", "build", "align", "reset"], p=[.3, .2, .2, .15, .1, .05])
    if op in ("add", "update"):
        n = int(rng.choice([0, 1, 2, 7, 64, 300, 1024, 1025, 2500]))
        xy = cloud(n)
        pose = None if op == "add" else rng.uniform(-1, 1, 3) * (0.3, 0.3, 0.2)
        rmap.insert_host(xy, pose)
        if pose is not None and n:
            c, s = np_ref.cos_sin(pose[2])   # one sincos(), like the reference built by GCC
            xy = np.stack([xy[:, 0] * c - xy[:, 1] * s + pose[0], xy[:, 0] * s + xy[:, 1] * c + pose[1]], axis=1)
        for q in xy:
            ref.add_point(q[0], q[1])
    elif op == "build":
        rmap.build(); ref.build()
    elif op == "align":
        new_xy = cloud(200) * 0.5
        new = oracle.Frame((0, 0, 0), frame_w, frame_h, float(max(frame_w, frame_h)))
        for q in new_xy:
            new.add_point(q[0], q[1])
        new_xy = new.points()
        if len(new_xy) == 0:
            continue
        scan.set(new_xy)
        table = oracle.glibc_rand(int(rng.integers(1, 1 << 30)), 3 + 3 * 9 + 6 * 9 * 6)
        got, cost, st = rmap.align(scan, (0, 0, 0), (.2, .2, .05), cfg, rand_table=table, mode=capi.SCORE_F64)
        want, want_cost, tr = ref.pso((0, 0, 0), new, (.2, .2, .05), ocfg, table=table)
        probes = rng.uniform(-1, 1, (5, 3)) * (.3, .3, .1)
        print(it, "align", got, cost, st, "| oracle", want, want_cost, tr)
        if it == stop:
            q = np.random.default_rng(1).uniform(-1, 1, (2000, 3)) * (.3, .3, .08)
            gc = rmap.cost(scan, q, mode=capi.SCORE_F64)
            wc = np.array([ref.cost(p, new) for p in q])
            print("nan pattern equal:", np.array_equal(np.isnan(gc), np.isnan(wc)), "nan counts", np.isnan(gc).sum(), np.isnan(wc).sum(),
                  "inf", np.isinf(gc).sum(), np.isinf(wc).sum())
            fin = np.isfinite(wc) & np.isfinite(gc)
            print("max |dcost| finite:", np.abs(gc[fin] - wc[fin]).max(), "max rel", (np.abs(gc[fin] - wc[fin]) / np.abs(wc[fin])).max())
            bad = np.where(np.isnan(gc) != np.isnan(wc))[0][:5]
            for b in bad:
                print("  pose", q[b], "device", gc[b], "oracle", wc[b])
            cells = [c for c in rmap.cells() if c["built"] and not np.isfinite(c["icov"]).all()]
            print("device non-finite cells:", [(c["index"], c["count"], c["mean"], c["icov"]) for c in cells])
            ocells = [c for c in ref.cells() if c["built"] and not np.isfinite(c["icov"]).all()]
            print("oracle non-finite cells:", [(c["index"], c["count"], c["mean"], c["icov"]) for c in ocells])
            for sd in range(5):
                t2 = oracle.glibc_rand(1000 + sd, 3 + 3 * 9 + 6 * 9 * 6)
                g2, c2, _ = rmap.align(scan, (0, 0, 0), (.2, .2, .05), cfg, rand_table=t2, mode=capi.SCORE_F64)
                w2, wc2, _ = ref.pso((0, 0, 0), new, (.2, .2, .05), ocfg, table=t2)
                print("  table", sd, "equal" if np.array_equal(g2, w2) else ("DIFF", g2, w2), c2, wc2)
    else:
        rmap.reset(); ref.reset_cells()
