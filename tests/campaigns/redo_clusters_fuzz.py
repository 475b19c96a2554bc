"""One-off: a few flagged pairs behind a large batch, redone on clusters of workgroups (round 6, align_pairs_dev_on_stream) --
random shapes around the cell sides where a handful of rooms outgrow the two-per-CU cell table: batches of 130 ... 700 pairs,
361 ... 1441 beams, cells of 0.2 ... 0.3 m (any value, not only the suite's), swarms of 17 ... 70 particles.  For every case and both
modes (fp64 score, exact): how many pairs the main launch leaves flagged (NDTPSO_NO_REDO), two calls with NDTPSO_REDO_CLUSTERS=0
and three without it -- poses and costs bit for bit the same, nothing left flagged -- and whether the calls went through clusters
(process counter).   usage: python tests/campaigns/redo_clusters_fuzz.py [n] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: E402,F401  (its HIP runtime first)
from ndtpso_slam_amd import capi, synth  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260930)
ctx = capi.Context(0)
used, t0, bad = 0, time.time(), []
for case in range(n_cases):
    n_beams = int(rng.choice([361, 541, 721, 1081, 1441]))
    fr = int(rng.choice([60, 100]))
    cs = float(np.round(rng.uniform(0.2, 0.3), 4))
    P, I = int(rng.choice([17, 24, 30, 70])), int(rng.integers(4, 16))
    B = int(rng.choice([130, 300, 512, 700]))
    p = synth.make_pairs(B, n_beams=n_beams, seed=int(rng.integers(1, 10**6)))
    geom = capi.ScanGeom(n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    grid, cfg = capi.Grid(fr, fr, cs), capi.PSOConfig.make(I, P)
    guess = rng.uniform(-1, 1, (B, 3)) * np.array([0.3, 0.3, 0.05]) * float(rng.choice([0.0, 1.0]))
    dev = (0.1, 0.1, 3.1415e-3)
    args = (p.ref_ranges, p.new_ranges, geom, grid, guess, dev, cfg)
    row = dict(case=case, beams=n_beams, frame=fr, cs=cs, P=P, I=I, B=B)
    for mname, mode in (("f64", capi.SCORE_F64), ("exact", capi.SCORE_EXACT)):
        os.environ["NDTPSO_NO_REDO"] = "1"
        _, _, st0 = ctx.align_pairs(*args, seeds=p.seeds, mode=mode)
        del os.environ["NDTPSO_NO_REDO"]
        row[mname + "_flagged"] = int(((st0["status"] & 0xffff) != 0).sum())
        os.environ["NDTPSO_REDO_CLUSTERS"] = "0"
        for _ in range(2):
            want = ctx.align_pairs(*args, seeds=p.seeds, mode=mode)
        del os.environ["NDTPSO_REDO_CLUSTERS"]
        before = capi.process_counters()["batches_redone_on_clusters"]
        for rep in range(3):
            got = ctx.align_pairs(*args, seeds=p.seeds, mode=mode)
            ok = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and \
                np.array_equal(got[2]["status"] & 0xffff, want[2]["status"] & 0xffff)
            if not ok:
                bad.append((row, mname, rep))
                print("DIFFERENT:", row, mname, rep, flush=True)
        row[mname + "_cluster_calls"] = capi.process_counters()["batches_redone_on_clusters"] - before
        used += row[mname + "_cluster_calls"] > 0
    print(row, flush=True)
print("%d cases, %d (case, mode) runs went through clusters, %d differences, %.0f s" % (n_cases, used, len(bad), time.time() - t0))
sys.exit(1 if bad else 0)
