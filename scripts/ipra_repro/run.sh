#!/bin/bash
# On a gfx950 box, after scripts/ipra_repro/make.sh: the failing configuration (130 pairs, 1024 particles x 12 iterations,
# swarm kept in HBM, the arbitration's unit form forced onto it) through the three builds.  Prints one JSON line per run:
# equal = pairs whose exact-mode pose equals the fp64 mode's.
#   * ipra1 twice with the whole batch, then the pairs it gets wrong ALONE, one workgroup each on an otherwise idle device
#     (NDTPSO_CLUSTER=0): a race would care who else runs; a wrong instruction does not.
cd "$(dirname "$0")/_build/tree"
export NDTPSO_UNITS_HBM=16 PYTHONPATH=.
for lib in ipra1 ipra0 ipra1_fenced; do
  echo "== $lib"
  NDTPSO_LIB=../$lib.so python scripts/units_hbm_diag.py 0 2>&1 | tail -2
done
echo "== ipra1, workgroup sizes"
for w in 4 16; do NDTPSO_WAVES=$w NDTPSO_LIB=../ipra1.so python scripts/units_hbm_diag.py 0 2>&1 | tail -1; done
echo "== ipra1, the failing pairs alone"
NDTPSO_LIB=../ipra1.so NDTPSO_CLUSTER=0 python - <<'PY'
import json, os, sys
import numpy as np
sys.path.insert(0, ".")
from ndtpso_slam_amd import capi, synth
B, Pn, In, beams, cs = 130, 1024, 12, 1081, 0.5
p = synth.make_pairs(B, n_beams=beams, seed=300 + In)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
ctx = capi.Context(0)
def run(idx, mode):
    return ctx.align_pairs(p.ref_ranges[idx], p.new_ranges[idx], geom, capi.Grid(60, 60, cs), (0, 0, 0), (0.1, 0.1, 3.1415e-3),
                           capi.PSOConfig.make(In, Pn), seeds=p.seeds[idx], mode=mode)
all_idx = np.arange(B)
p64, c64, _ = run(all_idx, capi.SCORE_F64)
px, cx, sx = run(all_idx, capi.SCORE_EXACT)
bad = np.nonzero(~(px == p64).all(axis=1))[0]
print(json.dumps(dict(batch_of_130_wrong=bad.tolist())))
alone = []
for b in bad.tolist():
    q, _, _ = run(np.array([b]), capi.SCORE_EXACT)
    alone.append(dict(pair=b, alone_equals_fp64=bool((q[0] == p64[b]).all()), alone_equals_batch_result=bool((q[0] == px[b]).all())))
print(json.dumps(dict(alone=alone)))
PY
