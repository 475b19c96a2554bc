"""Arbitration trace (-DNDTPSO_TRACE_ARB builds) of the unit form on swarms kept in HBM: runs case 0 of scripts/units_hbm_diag.py in
the exact mode and dumps, per pair, the sequence of gbest moves and arbitrations (task, score) into a JSON file.
   NDTPSO_LIB=... NDTPSO_UNITS_HBM=16 python scripts/units_hbm_trace.py out.json"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndtpso_slam_amd import capi, synth
B, Pn, In, beams, cs = 130, 1024, 12, 1081, 0.5
L = capi.load(build_if_missing=False)
ctx = capi.Context(0)
p = synth.make_pairs(B, n_beams=beams, seed=300 + In)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
args = (p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, cs), (0, 0, 0), (0.1, 0.1, 3.1415e-3), capi.PSOConfig.make(In, Pn))
p64, c64, s64 = ctx.align_pairs(*args, seeds=p.seeds, mode=capi.SCORE_F64)
KD = 2048
buf = np.zeros(B * KD, dtype=np.float64)
L.ndtpso_profile_arb_trace.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
L.ndtpso_profile_arb_trace(None, 0, 1)
px, cx, sx = ctx.align_pairs(*args, seeds=p.seeds, mode=capi.SCORE_EXACT)
rc = L.ndtpso_profile_arb_trace(buf.ctypes.data, B, 0)
bad = np.nonzero(~(px == p64).all(axis=1))[0].tolist()
out = {"rc": rc, "bad": bad, "arbitrated": sx["arbitrated"].tolist(), "trace": {}}
buf = buf.reshape(B, KD)
for b in range(B):
    n = int(buf[b, 0])
    out["trace"][b] = [float.hex(float(v)) if abs(v) < 1e3 and v != int(v) else v for v in buf[b, 1:n + 1].tolist()]
json.dump(out, open(sys.argv[1], "w"))
print("bad", bad, "rc", rc)
