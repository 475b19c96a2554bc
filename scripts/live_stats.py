"""Live-sequence statistics: per-scan time, comparisons arbitrated and hand-overs to the fp64 kernel on the node's
per-scan sequence (resident map, 30 x 50 PSO).  usage: python scripts/live_stats.py [n_scans] [score]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndtpso_slam_amd import capi, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_host_library import _trajectory
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mode = {"exact": capi.SCORE_EXACT, "f32": capi.SCORE_F32, "f64": capi.SCORE_F64}[sys.argv[2] if len(sys.argv) > 2 else "exact"]
ranges, _ = _trajectory(n)
geom = capi.ScanGeom(synth.N_BEAMS, float(synth.ANGLE_MIN), float(synth.ANGLE_INC), float(synth.RANGE_MAX), 0.1)
grid = capi.Grid(60, 60, 0.5)
cfg = capi.PSOConfig.make(50, 30)
n_draw = 3 + 3 * 30 + 6 * 30 * 50
tables = np.random.default_rng(5).integers(0, 2**31 - 1, size=(n, n_draw), dtype=np.int64).astype(np.int32)
ctx = capi.Context(0)
rmap = capi.ResidentMap(ctx, grid, og_cell_size=0.1, pool_bytes=256 << 20)
scan = capi.ResidentScan(ctx, 4096)
prev = np.zeros(3); hist = [np.zeros(3), np.zeros(3)]
arb, flags, evals, ts, rounds = [], [], [], [], []
for k in range(n):
    t0 = time.perf_counter()
    scan.load_scan(ranges[k], geom, clip=grid)
    if k > 0:
        dev = np.array((0.1, 0.1, 3.1415e-3)) if k <= 2 else np.abs(2.0 * (hist[-1] - hist[-2]))
        prev, _, st = rmap.align(scan, prev, dev, cfg, rand_table=tables[k], mode=mode)
        hist.append(prev.copy())
        arb.append(int(st["arbitrated"])); flags.append(int(st["status"]) & 0xffff); evals.append(int(st["cost_evals"])); rounds.append(int(st["rounds"]))
    rmap.insert(scan, prev)
    ts.append(time.perf_counter() - t0)
ctx.synchronize()
arb = np.array(arb); ts = np.array(ts[5:]); print("rounds field mean (arbitration ticks / 100 = us in -DNDTPSO_PROFILE_ARB builds): %.1f" % (np.mean(rounds) / 100.0))
print("scans %d  ms/scan median %.3f mean %.3f  -> %.0f scans/s" % (n, 1e3 * np.median(ts), 1e3 * ts.mean(), 1 / ts.mean()))
print("arbitrated per alignment: mean %.2f median %d p90 %d max %d; alignments with status flags %d; evals mean %.0f"
      % (arb.mean(), np.median(arb), np.percentile(arb, 90), arb.max(), int((np.array(flags) != 0).sum()), np.mean(evals)))
