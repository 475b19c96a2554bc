"""Long node replay: resident frames (clustered alignment) vs host-kept frames vs the oracle, pose by pose.
usage: python tests/campaigns/soak_replay.py [n_scans] [--oracle]      (SOAK_FRAME=60 SOAK_CELL=0.5 SOAK_SCORE=f64|exact in the environment)"""
import os, subprocess, sys, time
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
from ndtpso_slam_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1500
FRAME, CELL = os.environ.get("SOAK_FRAME", "60"), os.environ.get("SOAK_CELL", "0.5")
rng = np.random.default_rng(9)
s = np.linspace(0.0, 1.0, n)
# two laps around the room centre, heading turning with the path: the map is revisited, window slots rotate
poses = np.stack([6.0 * np.cos(4 * np.pi * s), 4.0 * np.sin(4 * np.pi * s), 4 * np.pi * s + np.pi / 2], axis=1)
clean = synth.raycast(poses)
ranges = np.where(clean > 0, clean + rng.normal(0, 0.01, clean.shape), 0.0).astype(np.float32)
with open('/tmp/soak.bin', 'wb') as f:
    np.array([n, synth.N_BEAMS], dtype=np.int32).tofile(f)
    np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
    ranges.tofile(f)
outs = {}
for tag, env in (("resident+cluster", {}), ("resident, one WG", {"NDTPSO_CLUSTER": "0"}), ("host frames", {"NDTPSO_RESIDENT": "0"})):
    t = time.time()
    r = subprocess.run(['host/replay/node_replay', '/tmp/soak.bin', FRAME, CELL, '50', '30', '7', '0.1', '/tmp/soak_' + tag.split()[0].strip(','), '5'],
                       capture_output=True, text=True, env=dict(os.environ, NDTPSO_SCORE=os.environ.get("SOAK_SCORE", "f64"), **env))
    outs[tag] = np.array([[float(v) for v in l.split()[1:]] for l in r.stdout.strip().splitlines()])
    print(tag, r.stderr.strip().splitlines()[-1], "wall %.1f s" % (time.time() - t), "final pose", outs[tag][-1])
base = outs["host frames"]
for tag, o in outs.items():
    d = np.abs(o - base)
    first = np.nonzero(d.max(axis=1) > 0)[0]
    print(f"{tag:18s} vs host frames: max |dpose| {d.max():.3e}  first differing scan {first[0] if len(first) else None}")
truth = poses.copy(); truth[:, :2] -= poses[0, :2]
if '--oracle' in sys.argv:
    from oracle import pyoracle as O
    cfg = O.PSOConfig.make(50, 30)
    n_draw = 3 + 3 * 30 + 6 * 30 * 50
    stream = O.glibc_rand(7, n_draw * n)
    ref = O.Frame((0, 0, 0), int(FRAME), int(FRAME), float(CELL))
    prev = np.zeros(3); want = []
    t = time.time()
    for k in range(n):
        cur = O.Frame((0, 0, 0), int(FRAME), int(FRAME), float(FRAME))
        cur.load_laser(ranges[k], synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX)
        pose = prev.copy() if k == 0 else ref.align(prev, cur, cfg, table=stream[(k - 1) * n_draw:k * n_draw])
        prev = pose; ref.update(pose, cur); want.append(pose)
    want = np.array(want)
    d = np.abs(outs["resident+cluster"] - want)
    first = np.nonzero(d.max(axis=1) > 0)[0]
    print("oracle: %.1f s; resident+cluster vs oracle max |dpose| %.3e, first differing scan %s" % (time.time() - t, d.max(), first[0] if len(first) else None))
    if len(first):
        k = first[0]
        print("  at that scan |dpose| =", d[k], "; ten scans later", d[min(k + 10, n - 1)], "; final poses", outs["resident+cluster"][-1], want[-1])
