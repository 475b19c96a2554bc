"""R replicas of the live sequence per process x N processes at once on one GPU (host/replay/node_replicas): is the ceiling of
one process -- ~20 k scans/s at 16 replicas -- the device's or the process's (the HIP runtime's launch path is shared by a
process's threads)?    python scripts/replicas_multiprocess.py "1x16" "2x16" "4x8" "2x32" ...   (processes x replicas)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ndtpso_slam_amd import synth  # noqa: E402

exe = os.path.join(ROOT, "host", "replay", "node_replicas")
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "scans.bin")
    bench._write_live_scans(synth, path, 300)
    for spec in sys.argv[1:] or ["1x16", "2x16"]:
        n_proc, R = (int(x) for x in spec.split("x"))
        env = dict(os.environ, NDTPSO_RESIDENT="1", NDTPSO_SCORE="exact")
        procs = [subprocess.Popen([exe, path, "60", "0.5", "50", "30", str(7 + 100 * k), str(R)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
                 for k in range(n_proc)]
        outs = [p.communicate(timeout=600) for p in procs]
        rows = [json.loads(o[0].strip().splitlines()[-1]) for o in outs if o[0].strip()]
        print(json.dumps({"processes": n_proc, "replicas_per_process": R, "aggregate_scans_per_s": sum(r["aggregate_scans_per_s"] for r in rows),
                          "per_process": [r["aggregate_scans_per_s"] for r in rows], "ms_per_scan_mean": [r["ms_per_scan_mean"] for r in rows],
                          "ms_per_scan_p95": [r["ms_per_scan_p95"] for r in rows], "ms_per_scan_max": [r["ms_per_scan_max"] for r in rows],
                          "cluster_timeouts": sum(r["cluster_timeouts"] for r in rows), "failed": sum(r["failed_alignments"] for r in rows)}), flush=True)
