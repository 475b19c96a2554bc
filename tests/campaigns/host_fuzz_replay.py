"""Debug helper: re-runs an op script of host/replay/frame_fuzz (as written by tests/test_host_library.py) against the
oracle with a `build` inserted after every op, and reports the first build whose cells differ.
usage: python tests/campaigns/host_fuzz_replay.py ops.txt [resident=1] [score=f64]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import pyoracle as oracle
import test_host_library as T
T._build()
src = sys.argv[1]
resident = sys.argv[2] if len(sys.argv) > 2 else "1"
score = sys.argv[3] if len(sys.argv) > 3 else "f64"
fh = float.fromhex
ops = [l.split() for l in open(src)]
W, H, cs = int(ops[0][1]), int(ops[0][2]), fh(ops[0][3])
ref = oracle.Frame((0, 0, 0), W, H, cs)


def new_frame(tok):
    n = int(tok[0])
    nf = oracle.Frame((0, 0, 0), W, H, float(max(W, H)))
    for i in range(n):
        nf.add_point(fh(tok[1 + 2 * i]), fh(tok[2 + 2 * i]))
    return nf


out_lines, expect, expect_pts = [" ".join(ops[0])], [], []
for k, t in enumerate(ops[1:], 1):
    if t[0] == "end":
        break
    if t[0] == "points":
        continue
    out_lines.append(" ".join(t))
    if t[0] == "add":
        for i in range(int(t[1])):
            ref.add_point(fh(t[2 + 2 * i]), fh(t[3 + 2 * i]))
    elif t[0] == "update":
        ref.update([fh(v) for v in t[1:4]], new_frame(t[4:]))
    elif t[0] == "build":
        ref.build()
    elif t[0] == "reset":
        ref.reset_cells()
    elif t[0] == "cost":
        ref.cost([fh(v) for v in t[1:4]], new_frame(t[4:]))
    elif t[0] == "pso":
        ref.pso([fh(v) for v in t[2:5]], new_frame(t[10:]), [fh(v) for v in t[5:8]],
                oracle.PSOConfig.make(int(t[8]), int(t[9])), seed=int(t[1]))
    elif t[0] == "align":
        ref.align([fh(v) for v in t[2:5]], new_frame(t[5:]), None, seed=int(t[1]))
    if t[0] != "build":
        out_lines.append("build")
        ref.build()
    expect.append((k, t[0], [(c["index"], c["built"], c["count"], c["mean"]) for c in ref.cells()]))
    out_lines.append("points")
    pxy = ref.points()
    sx = sy = 0.0
    for q in pxy:
        sx += q[0]
        sy += q[1]
    expect_pts.append((len(pxy), sx, sy))
out_lines.append("end")
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "ops.txt")
    open(path, "w").write("\n".join(out_lines) + "\n")
    out = subprocess.check_output([os.path.join(T.HOST, "replay", "frame_fuzz"), path], text=True,
                                  env=dict(os.environ, NDTPSO_RESIDENT=resident, NDTPSO_SCORE=score))
builds = [l.split() for l in out.splitlines() if l.startswith("build")]
assert len(builds) == len(expect), (len(builds), len(expect))
pts_lines = [l.split() for l in out.splitlines() if l.startswith("points")]
for (k, op, _), g, w in zip(expect, pts_lines, expect_pts):
    same = (int(g[1]), fh(g[2]), fh(g[3])) == w
    if not same:
        print(f"slot-0 points after op {k} ({op}): device {g[1:]} oracle {(w[0], w[1].hex(), w[2].hex())}")
        break
for got, (k, op, cells) in zip(builds, expect):
    bad = []
    if int(got[1]) != len(cells):
        bad.append(("created", int(got[1]), len(cells)))
    else:
        for i, (index, built, count, mean) in enumerate(cells):
            gi, gb, gx, gy = got[2 + 4 * i: 6 + 4 * i]
            if (int(gi), bool(int(gb))) != (index, built) or (built and not np.array_equal([fh(gx), fh(gy)], mean, equal_nan=True)):
                bad.append((index, int(gb), built, count, gx, gy, float(mean[0]).hex(), float(mean[1]).hex()))
    print(f"after op {k} ({op}): {'ok' if not bad else bad[:4]}")
    if bad:
        break
