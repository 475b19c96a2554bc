"""How often does an alignment contain a cost comparison closer than a relative epsilon?  (Diagnostic builds:
hipcc ... -DNDTPSO_COUNT_AMBIG=<eps>, loaded through NDTPSO_LIB; the count comes back in stats.gbest_updates.)
usage: NDTPSO_LIB=ab/lib_amb_2e-7.so python scripts/ambiguous_comparisons.py [pairs]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from ndtpso_slam_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p = synth.make_pairs(B, seed=2024)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
ctx = capi.Context(0)
for P, I in ((70, 70), (30, 50)):
    _, _, st = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, 0.5), (0, 0, 0), (0.1, 0.1, 3.1415e-3),
                               capi.PSOConfig.make(I, P), seeds=p.seeds, mode=capi.SCORE_F32)
    n = st["gbest_updates"].astype(int)
    print(f"{P} x {I}: alignments with at least one close comparison: {(n > 0).sum()} / {B}; mean count {n.mean():.2f}, max {n.max()}")
