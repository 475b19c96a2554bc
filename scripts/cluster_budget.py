"""Where a clustered k_align of the live sequence spends its time: -DNDTPSO_PHASE_BUDGET build (python -m ndtpso_slam_amd.build
budget), the first workgroup's account (thread 0 = the wave that polls the exchange; thread 64 = a wave that does not), read
back after every alignment and averaged.   usage (GPU box): python scripts/cluster_budget.py [n_scans]   (NDTPSO_CLUSTER_SPEC=0 ...)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NDTPSO_LIB", os.path.join(ROOT, "ndtpso_slam_amd", "lib", "libndtpso_hip_budget.so"))
from ndtpso_slam_amd import capi, synth
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_host_library import _trajectory
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
L = capi.load(build_if_missing=False)
L.ndtpso_profile_phase_budget.argtypes = [C.c_void_p, C.c_uint32]
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
acc = []
buf = np.zeros(16, dtype=np.uint32)
for k in range(n):
    scan.load_scan(ranges[k], geom, clip=grid)
    if k > 0:
        dev = np.array((0.1, 0.1, 3.1415e-3)) if k <= 2 else np.abs(2.0 * (hist[-1] - hist[-2]))
        prev, _, st = rmap.align(scan, prev, dev, cfg, rand_table=tables[k], mode=capi.SCORE_EXACT)
        hist.append(prev.copy())
        ctx.synchronize()
        L.ndtpso_profile_phase_budget(buf.ctypes.data, 1)
        if k > 5: acc.append(np.concatenate([buf.astype(np.float64) * 0.01, [float(st["rounds"]), float(st["gbest_updates"])]]))
    rmap.insert(scan, prev)
a = np.mean(acc, axis=0)
names = {1: "swarm initialisation", 2: "top of iteration", 3: "proposal steps + barrier", 4: "thread 0: evaluation + exchange (poll)",
         5: "between the last round and the end of the iteration", 6: "thread 0: wait at the round's barrier", 7: "arbitration", 8: "commit steps + gbest barriers", 9: "end of iteration",
         10: "final cost", 11: "commit-and-pick steps (SpecP)", 12: "thread 64: evaluation (+ the next proposals)", 14: "thread 64: wait at the round's barrier"}
print("mean over %d alignments, us (rounds %.1f, gbest moves %.1f):" % (len(acc), a[16], a[17]))
for k in sorted(names):
    if names[k] != "-": print("  %-50s %7.1f" % (names[k], a[k]))
print("  sum of thread 0's phases %.1f" % a[1:12].sum())
print("  sweeps of the exchange's slots by the polling wave: %.1f per alignment, %.2f per round" % (a[15] * 100, a[15] * 100 / a[16]))
