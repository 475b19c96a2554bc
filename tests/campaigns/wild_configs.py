"""One-off: the fused pairs path on configurations far outside the suite's and the benchmark's -- cells of 0.1 ... 2 m (most not
powers of two), frames of 10 ... 300 m (also not square), 5 ... 3000 beams, swarms of 1 ... 300 particles, 0 ... 30 iterations,
batches of 1 ... 700 pairs (clusters of workgroups, one workgroup per pair at 16 waves, two per compute unit at 8), ranges cut
short or mostly dropped, guesses metres off, deviations from 1e-9 to 1 -- against the oracle.  For every configuration: the fp64
and the exact mode through ndtpso_align_pairs agree bit for bit on every pair, nothing stays flagged, and up to six pairs equal the
oracle's (pose 1e-9, cost 1e-8 relative); a configuration the library refuses must be refused LOUDLY (an NdtpsoError), never
answered wrongly.   usage: python tests/campaigns/wild_configs.py [n] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from ndtpso_slam_amd import capi, synth  # noqa: E402
from oracle import pyoracle  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606060)
ctx = capi.Context(0)
refused, flagged_cases, checked, t0 = [], [], 0, time.time()
for case in range(n_cases):
    n_beams = int(rng.choice([5, 17, 64, 65, 181, 361, 541, 721, 1081, 1441, 2048, 3000]))
    fw, fh = int(rng.choice([10, 20, 40, 60, 100, 150, 300])), int(rng.choice([10, 20, 40, 60, 100, 150, 300]))
    cs = float(rng.choice([0.1, 0.125, 0.15, 0.2, 0.25, 0.3, 0.4, 0.5, 0.7, 1.0, 1.3, 2.0]))
    if fw / cs > 4000 or fh / cs > 4000:
        cs = 0.5
    P, I = int(rng.choice([1, 2, 3, 8, 16, 17, 30, 64, 65, 70, 128, 300])), int(rng.integers(0, 31))
    B = int(rng.choice([1, 2, 3, 9, 40, 130, 300, 530, 700]))
    if P * I * B * n_beams > 6e9:      # (keep a case to seconds)
        B = max(1, int(6e9 / (P * max(I, 1) * n_beams)))
    p = synth.make_pairs(B, n_beams=n_beams, seed=int(rng.integers(1, 10**6)))
    ref, new = p.ref_ranges.copy(), p.new_ranges.copy()
    kind = int(rng.integers(0, 5))
    if kind == 1:                      # most beams dropped
        ref[rng.random(ref.shape) < 0.8] = 0.0
    elif kind == 2:                    # a short-range sensor
        ref, new = np.minimum(ref, 4.0).astype(np.float32), np.minimum(new, 4.0).astype(np.float32)
    elif kind == 3:                    # scan B nearly empty
        new[rng.random(new.shape) < 0.97] = 0.0
    guess = rng.uniform(-1, 1, (B, 3)) * np.array([1.0, 1.0, 0.3]) * float(rng.choice([0.0, 0.05, 1.0, 3.0]))
    dev = np.abs(rng.normal(0, 1, (B, 3))) * float(rng.choice([1e-9, 1e-4, 0.1, 1.0])) + 1e-12
    geom = capi.ScanGeom(n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    grid, cfg = capi.Grid(fw, fh, cs), capi.PSOConfig.make(I, P)
    tag = dict(case=case, beams=n_beams, frame=(fw, fh), cs=cs, P=P, I=I, B=B, kind=kind)
    try:
        got64, cost64, st64 = ctx.align_pairs(ref, new, geom, grid, guess, dev, cfg, seeds=p.seeds, mode=capi.SCORE_F64)
        gotx, costx, stx = ctx.align_pairs(ref, new, geom, grid, guess, dev, cfg, seeds=p.seeds, mode=capi.SCORE_EXACT)
    except capi.NdtpsoError as e:
        refused.append((tag, str(e)))
        print("case %d refused loudly: %s  %s" % (case, e, tag), flush=True)
        continue
    left = np.nonzero(((st64["status"] | stx["status"]) & 0xffff) != 0)[0]
    if left.size:                      # pairs no path could hold: flagged, never answered wrongly -- reported, and left out below
        flagged_cases.append((tag, int(left.size)))
        print("case %d: %d of %d pairs left flagged (status %s)  %s" % (case, left.size, B, np.unique(st64["status"][left] & 0xffff), tag), flush=True)
    keep = np.setdiff1d(np.arange(B), left)
    if keep.size == 0:
        continue
    got64, cost64, gotx, costx = got64[keep], cost64[keep], gotx[keep], costx[keep]
    ref, new, guess, dev, seeds_k = ref[keep], new[keep], guess[keep], dev[keep], p.seeds[keep]
    B = int(keep.size)
    same = np.array_equal(got64, gotx, equal_nan=True) and np.array_equal(cost64, costx, equal_nan=True)
    assert same, (tag, np.nonzero((got64 != gotx).any(axis=1))[0][:8])
    k = np.unique(rng.integers(0, B, size=min(B, 6)))
    want, wcost, _ = pyoracle.align_pairs(ref[k], new[k], p.angle_min, p.angle_inc, p.range_max, 0.1, fw, fh, cs, guess[k], dev[k],
                                          pyoracle.PSOConfig.make(I, P), seeds_k[k])
    dp = np.abs(got64[k] - want)
    ok = np.all((dp < 1e-9) | (np.isnan(want) & np.isnan(got64[k])))
    okc = np.all((np.abs(cost64[k] - wcost) <= 1e-8 * np.maximum(1.0, np.abs(wcost))) | (np.isnan(wcost) & np.isnan(cost64[k])))
    assert ok and okc, (tag, got64[k], want, cost64[k], wcost)
    checked += 1
    if case % 20 == 0:
        print("case %d ok %s (%.0f s)" % (case, tag, time.time() - t0), flush=True)
print("%d / %d configurations identical to the oracle (fp64 == exact on every pair); %d refused loudly: %s; %d with pairs left flagged: %s"
      % (checked, n_cases, len(refused), sorted({r[1] for r in refused}), len(flagged_cases), flagged_cases[:6]))
