"""Config-5 soak on the GPU box: large swarms whose state lives in the HBM workspace (staging of the next round's constants
by the light wave), several batch sizes and iteration counts, exact mode against the fp64 mode bit for bit.
usage: python scripts/config5_soak.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndtpso_slam_amd import capi, synth
ctx = capi.Context(0)
bad = 0
for (B, P, I, beams, cs) in [(256, 2048, 3, 2048, 0.25), (140, 2048, 9, 2048, 0.25), (130, 1024, 12, 1081, 0.5), (200, 700, 20, 1441, 0.3),
                             (256, 2048, 40, 2048, 0.25)]:
    p = synth.make_pairs(B, n_beams=beams, seed=21 + I)
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    args = (p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, cs), (0, 0, 0), (0.1, 0.1, 3.1415e-3), capi.PSOConfig.make(I, P))
    pe, ce, se = ctx.align_pairs(*args, seeds=p.seeds, mode=capi.SCORE_EXACT)
    pf, cf, sf = ctx.align_pairs(*args, seeds=p.seeds, mode=capi.SCORE_F32)
    n64 = min(B, 132)
    p6, c6, s6 = ctx.align_pairs(p.ref_ranges[:n64], p.new_ranges[:n64], *args[2:], seeds=p.seeds[:n64], mode=capi.SCORE_F64)
    same = np.array_equal(pe[:n64], p6) and np.array_equal(ce[:n64], c6)
    plan = capi.align_pairs_describe(geom, capi.Grid(60, 60, cs), capi.PSOConfig.make(I, P), capi.SCORE_EXACT, B)[1]
    print("B %d P %d I %d beams %d cs %.2f: swarm_in_hbm %d, exact == f64 on %d pairs: %s, flags %d, f32 within 1e-6 of exact on %.1f %%"
          % (B, P, I, beams, cs, plan["swarm_in_hbm"], n64, same, int(((se["status"] & 0xffff) != 0).sum()),
             100 * (np.abs(pf - pe).max(axis=1) < 1e-6).mean()))
    bad += 0 if same else 1
sys.exit(1 if bad else 0)
