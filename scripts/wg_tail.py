"""Where the tail of a launch comes from: per-workgroup durations of the fused pairs kernel (ndtpso_align_stats.t_start /
t_end, 100 MHz) over several launches of BASELINE config 3 -- are the slow workgroups the same PAIRS every launch (work
inherent to the pair: phases, gbest updates, arbitrated comparisons) or the same PLACES (compute unit, partner)?
    python scripts/wg_tail.py [--score exact|f32] [--out gpurun_out/wg_tail.json]     (GPU box)"""
import argparse, json, sys
import numpy as np, torch
sys.path.insert(0, '.')
from ndtpso_slam_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--score", default="exact")
ap.add_argument("--out", default="")
ap.add_argument("--launches", type=int, default=8)
args = ap.parse_args()
B = 512
p = synth.make_pairs(B, seed=2024)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
mode = {"exact": capi.SCORE_EXACT, "f32": capi.SCORE_F32, "f64": capi.SCORE_F64}[args.score]
ctx = capi.Context(0)
D, S0 = [], []
perm = None
for rep in range(args.launches + 2):
    # odd launches run the pairs in reversed order: a pair then sits in another workgroup (place), the work stays its own
    rev = rep % 2 == 1
    idx = np.arange(B)[::-1] if rev else np.arange(B)
    got, cost, st = ctx.align_pairs(p.ref_ranges[idx], p.new_ranges[idx], geom, capi.Grid(60, 60, 0.5), (0, 0, 0), (0.1, 0.1, 3.1415e-3),
                                    capi.PSOConfig.make(70, 70), seeds=p.seeds[idx], mode=mode)
    if rep < 2:
        continue
    t0 = st["t_start"].astype(np.int64); t1 = st["t_end"].astype(np.int64)
    d = (t1 - t0) * 0.01
    D.append((rev, d, st.copy(), (t0 - t0.min()) * 0.01, (t1 - t0.min()) * 0.01))
out = {}
by_place = np.stack([d for rev, d, *_ in D])                       # [launch, workgroup]
by_pair = np.stack([d[::-1] if rev else d for rev, d, *_ in D])   # [launch, pair]
def mean_corr(a):
    c = np.corrcoef(a)
    return float(c[np.triu_indices(len(a), 1)].mean())
fw = [i for i, (rev, *_) in enumerate(D) if not rev]
bw = [i for i, (rev, *_) in enumerate(D) if rev]
out["duration_us"] = {"mean": float(by_place.mean()), "p50": float(np.median(by_place)), "p95": float(np.percentile(by_place, 95)),
                      "max_mean_over_launches": float(by_place.max(axis=1).mean())}
out["span_us_mean"] = float(np.mean([e.max() for *_, e in D]))
out["corr_same_pair_same_place"] = mean_corr(by_place[fw])
out["corr_same_pair_other_place"] = float(np.mean([np.corrcoef(by_pair[i], by_pair[j])[0, 1] for i in fw for j in bw]))
out["corr_same_place_other_pair"] = float(np.mean([np.corrcoef(by_place[i], by_place[j])[0, 1] for i in fw for j in bw]))
rev, d, st, s0, e0 = D[0]
feat = {k: st[k].astype(float) for k in ("cost_evals", "rounds", "gbest_updates", "arbitrated", "n_points", "n_built")}
out["corr_duration_feature"] = {k: float(np.corrcoef(d, v)[0, 1]) if v.std() > 0 else None for k, v in feat.items()}
X = np.stack([v for v in feat.values()] + [np.ones(B)], axis=1)
coef, res, *_ = np.linalg.lstsq(X, d, rcond=None)
pred = X @ coef
out["linear_fit"] = {"r2": float(1 - ((d - pred) ** 2).sum() / ((d - d.mean()) ** 2).sum()), "coef": dict(zip(list(feat) + ["const"], map(float, coef)))}
# the partner: blocks b and b + B/2 share a compute unit (dispatch-order assumption)
h = B // 2
out["corr_partner_durations"] = float(np.corrcoef(d[:h], d[h:])[0, 1])
out["cu_pair_sum_us"] = {"mean": float((d[:h] + d[h:]).mean() / 2), "max": float((d[:h] + d[h:]).max() / 2)}
out["by_block_mod_8_us"] = [float(d[np.arange(B) % 8 == k].mean()) for k in range(8)]
out["first_half_second_half_us"] = [float(d[:h].mean()), float(d[h:].mean())]
out["start_spread_us"] = float(s0.max())
print(json.dumps(out, indent=1))
if args.out:
    json.dump(out, open(args.out, "w"), indent=1)
