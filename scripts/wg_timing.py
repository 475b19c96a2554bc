"""Per-workgroup start/end times of the fused kernel (ndtpso_align_stats.t_start/t_end): load balance between
the two workgroups sharing a CU and across the chip.  Run on the GPU box: python scripts/wg_timing.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from ndtpso_slam_amd import capi, synth
B = 512
p = synth.make_pairs(B, seed=2024)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
ctx = capi.Context(0)
for rep in range(2):
    got, cost, st = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, 0.5), (0, 0, 0), (0.1, 0.1, 3.1415e-3),
                                    capi.PSOConfig.make(70, 70), seeds=p.seeds, mode=capi.SCORE_F32)
t0 = st["t_start"].astype(np.int64); t1 = st["t_end"].astype(np.int64)
base = t0.min(); s = (t0 - base) / 100.0; e = (t1 - base) / 100.0   # microseconds at 100 MHz
print("kernel span us", e.max(), "start spread us", s.max(), "end min/mean/max", e.min(), e.mean(), e.max())
print("duration min/mean/max", (e - s).min(), (e - s).mean(), (e - s).max())
print("residency (mean duration / span)", (e - s).mean() / e.max())
half = np.arange(B) >= B // 2   # later-dispatched partner of each CU (dispatch-order assumption)
print("mean duration first half / second half of the grid:", (e - s)[~half].mean(), (e - s)[half].mean())
ev = st["cost_evals"].astype(float)
print("corr(duration, evals)", np.corrcoef(e - s, ev)[0, 1])
order = np.argsort(e)
print("latest 5 ends", e[order[-5:]], "their evals", ev[order[-5:]], "starts", s[order[-5:]])
print("late starters (start > 100us):", (s > 100).sum())
d = e - s
print("corr(duration, n_built)", np.corrcoef(d, st["n_built"].astype(float))[0, 1])
# blockIdx vs duration trend
print("corr(duration, blockIdx)", np.corrcoef(d, np.arange(B))[0, 1])
nv = st["n_points"].astype(float)
print("n_points min/max", nv.min(), nv.max())
