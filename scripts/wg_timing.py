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
t0 = st["reserved"][:, 0].astype(np.int64); t1 = st["reserved"][:, 1].astype(np.int64)
base = t0.min(); s = (t0 - base) / 100.0; e = (t1 - base) / 100.0   # microseconds at 100 MHz
print("kernel span us", e.max(), "start spread us", s.max(), "end min/mean/max", e.min(), e.mean(), e.max())
print("duration min/mean/max", (e - s).min(), (e - s).mean(), (e - s).max())
print("residency (mean duration / span)", (e - s).mean() / e.max())
hw = st["gbest_updates"]
cu = ((hw >> 8) & 0xf); se = ((hw >> 13) & 0x7); xcc = (hw >> 16) & 0xf
key = xcc * 1000 + se * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc,se,cu):", len(u), "WGs per CU histogram:", np.bincount(cnt))
ev = st["cost_evals"].astype(float)
print("corr(duration, evals)", np.corrcoef(e - s, ev)[0, 1])
order = np.argsort(e)
print("latest 5 ends", e[order[-5:]], "their evals", ev[order[-5:]], "starts", s[order[-5:]])
print("late starters (start > 100us):", (s > 100).sum())
d = e - s
print("corr(duration, n_built)", np.corrcoef(d, st["n_built"].astype(float))[0, 1])
for x in range(8):
    m = xcc == x
    print("xcc", x, "n", m.sum(), "mean dur %.0f  min %.0f max %.0f" % (d[m].mean(), d[m].min(), d[m].max()))
# partners on the same CU
pairs_d = []
for k in u:
    idx = np.nonzero(key == k)[0]
    if len(idx) == 2: pairs_d.append((d[idx[0]], d[idx[1]]))
pairs_d = np.array(pairs_d)
print("corr between CU partners' durations", np.corrcoef(pairs_d[:, 0], pairs_d[:, 1])[0, 1])
print("CU finish time (max of partners): min/mean/max", pairs_d.max(1).min(), pairs_d.max(1).mean(), pairs_d.max(1).max())
print("sum of partner evals vs CU finish corr", np.corrcoef([ev[np.nonzero(key == k)[0]].sum() for k in u], [d[np.nonzero(key == k)[0]].max() for k in u])[0, 1])
# blockIdx vs duration trend
print("corr(duration, blockIdx)", np.corrcoef(d, np.arange(B))[0, 1])
nv = st["n_points"].astype(float)
print("n_points min/max", nv.min(), nv.max())
