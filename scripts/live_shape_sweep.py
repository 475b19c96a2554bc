"""Cluster shapes (workgroups x waves) on the live sequence through the C++ drop-in (host/replay/node_replay), 400 scans, two passes;
the pose logs must not depend on the shape.   usage (GPU box): python scripts/live_shape_sweep.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, ROOT)
from live_timeline import write_scans
exe = os.path.join(ROOT, "host", "replay", "node_replay")
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "scans.bin"); write_scans(path, 400)
    ref = None
    for rep in range(2):
        for K, W in ((None, None), (8, 4), (16, 2), (10, 3), (6, 5), (4, 8), (11, 3), (8, 5), (16, 4), (32, 1)):
            env = dict(os.environ, NDTPSO_RESIDENT="1", NDTPSO_SCORE="exact")
            if K: env["NDTPSO_CLUSTER"] = str(K); env["NDTPSO_CLUSTER_WAVES"] = str(W)
            p = subprocess.run([exe, path, "60", "0.5", "50", "30", "7"], capture_output=True, text=True, timeout=300, env=env)
            m = re.search(r"matching rate: ([0-9.]+) Hz", p.stderr)
            if ref is None: ref = p.stdout
            print(K, W, m.group(1) if m else p.stderr[-200:], "same" if p.stdout == ref else "DIFFERENT", flush=True)
