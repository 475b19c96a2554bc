"""R replicas of the live sequence in one process (host/replay/node_replicas through bench._live_replicas): aggregate rate and
latencies over R, for settings of the environment given as KEY=VALUE[,KEY=VALUE...] groups separated by spaces.
    python scripts/replicas_wait_ab.py "NDTPSO_WAIT=spin" "NDTPSO_CLUSTER_MAX_INFLIGHT=8" ...      (GPU box; "" = defaults)"""
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ndtpso_slam_amd import synth  # noqa: E402

groups = sys.argv[1:] or [""]
rs = [int(x) for x in os.environ.get("REPLICAS", "16,32,64").split(",")]
keep = ("replicas", "aggregate_scans_per_s", "ms_per_scan_mean", "ms_per_scan_p95", "ms_per_scan_max", "cluster_timeouts",
        "host_cpus_busy_user", "host_cpus_busy_sys", "ms_between_scans_mean", "failed_alignments", "error")
for g in groups:
    env = dict(kv.split("=", 1) for kv in g.split(",") if kv)
    for R in rs:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        d = bench._live_replicas(synth, "exact", R, n_scans=200)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        print(g or "defaults", json.dumps({k: d.get(k) for k in keep}), flush=True)
