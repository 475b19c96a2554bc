"""Device timeline of one scan of the live sequence (host/replay/node_replay on the C++ drop-in): kernel and copy spans
from a rocprofv3 --kernel-trace --memory-copy-trace run, folded over the scans (one period = one k_align start to the
next), mean start offset / duration / gap before each operation.
  on the GPU box:  python scripts/live_timeline.py run <out_dir> [n_scans]   (writes scans, runs rocprofv3, folds)
  anywhere:        python scripts/live_timeline.py fold <out_dir>"""
import csv, glob, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_scans(path, n):
    sys.path.insert(0, ROOT)
    from ndtpso_slam_amd import synth
    rng = np.random.default_rng(4)
    s = np.linspace(0.0, 0.6 * n / 200, n)
    poses = np.stack([2.0 + 1.2 * s, -1.0 + 0.8 * np.sin(1.5 * s), 0.3 + 0.25 * s], axis=1)
    clean = synth.raycast(poses)
    ranges = np.where(clean > 0, clean + rng.normal(0, 0.01, clean.shape), 0.0).astype(np.float32)
    with open(path, "wb") as f:
        np.array([n, synth.N_BEAMS], dtype=np.int32).tofile(f)
        np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
        ranges.tofile(f)


def fold(out):
    ops = []
    for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
    for f in glob.glob(os.path.join(out, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r["Direction"]))
    ops.sort()
    starts = [i for i, o in enumerate(ops) if o[2].startswith("k_align") or "k_align<" in o[2]]
    periods = [ops[a:b + 1] for a, b in zip(starts[20:-2], starts[21:-1])]  # steady state; includes the next k_align
    shape = [tuple(o[2] for o in p) for p in periods]
    common = max(set(shape), key=shape.count)
    sel = [p for p, s in zip(periods, shape) if s == common]
    rows = []
    for j, name in enumerate(common):
        st = np.array([p[j][0] - p[0][0] for p in sel]) / 1e3
        du = np.array([p[j][1] - p[j][0] for p in sel]) / 1e3
        gap = np.array([p[j][0] - p[j - 1][1] for p in sel]) / 1e3 if j else np.zeros(len(sel))
        rows.append({"op": name, "start_us": round(float(st.mean()), 1), "dur_us": round(float(du.mean()), 1), "gap_before_us": round(float(gap.mean()), 1)})
    res = {"periods_folded": len(sel), "of": len(periods), "period_us": rows[-1]["start_us"],
           "busy_us": round(sum(r["dur_us"] for r in rows[:-1]), 1), "gaps_us": round(sum(r["gap_before_us"] for r in rows), 1), "ops": rows[:-1]}
    json.dump(res, open(os.path.join(out, "live_timeline.json"), "w"), indent=1)
    print("period %.1f us = busy %.1f + gaps %.1f  (%d of %d periods share this shape)" % (res["period_us"], res["busy_us"], res["gaps_us"], len(sel), len(periods)))
    for r in rows:
        print("  +%7.1f  gap %5.1f  dur %6.1f  %s" % (r["start_us"], r["gap_before_us"], r["dur_us"], r["op"]))


if __name__ == "__main__":
    out = os.path.abspath(sys.argv[2])
    os.makedirs(out, exist_ok=True)
    if sys.argv[1] == "run":
        n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
        scans = os.path.join(out, "scans.bin")
        write_scans(scans, n)
        env = dict(os.environ, NDTPSO_RESIDENT="1", NDTPSO_SCORE=os.environ.get("NDTPSO_SCORE", "exact"), TMPDIR="/tmp")
        exe = os.path.join(ROOT, "host", "replay", "node_replay")
        p = subprocess.run(["rocprofv3", "--kernel-trace", "--memory-copy-trace", "--output-format", "csv", "-d", os.path.join(out, "trace"), "-o", "t", "--",
                            exe, scans, "60", "0.5", "50", "30", "7"], cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
        print(p.stderr[-300:])
        os.remove(scans)
    fold(out)
