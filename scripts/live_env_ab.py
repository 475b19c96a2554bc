"""Same-box A/B of one environment switch of the library on the live sequence through the C++ drop-in (host/replay/node_replay):
runs with the switch set ("other") and without ("default") interleaved; the pose logs must be byte-identical.
   usage: python scripts/live_env_ab.py [n_scans] [repeats] [VAR=VALUE]      default switch: NDTPSO_CLUSTER_SPREAD=1 (a cluster's
   workgroups spread over the XCDs instead of on one); NDTPSO_CLUSTER_SPEC=0: no ready-made next proposals (SpecP)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from live_timeline import write_scans
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
var, val = (sys.argv[3] if len(sys.argv) > 3 else "NDTPSO_CLUSTER_SPREAD=1").split("=")
exe = os.path.join(ROOT, "host", "replay", "node_replay")
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "scans.bin")
    write_scans(path, n)
    logs, rates = {}, {"other": [], "default": []}
    for r in range(rep):
        for which in ("other", "default"):
            env = dict(os.environ, NDTPSO_RESIDENT="1", NDTPSO_SCORE=os.environ.get("NDTPSO_SCORE", "exact"))
            if which == "other":
                env[var] = val
            p = subprocess.run([exe, path, "60", "0.5", "50", "30", "7"], capture_output=True, text=True, timeout=300, env=env)
            m = re.search(r"matching rate: ([0-9.]+) Hz \(([0-9.]+) ms per scan\)", p.stderr)
            if p.returncode != 0 or not m:
                sys.exit("node_replay failed (%s): %s" % (which, p.stderr[-400:]))
            rates[which].append(float(m.group(1)))
            print("%-8s %s scans/s  %s ms/scan" % (which, m.group(1), m.group(2)), flush=True)
            logs.setdefault(which, p.stdout)
            if logs[which] != p.stdout:
                sys.exit("run-to-run difference in the pose log (%s)" % which)
    print("pose logs identical: %s (%d scans); best with %s=%s %.1f, best default %.1f scans/s"
          % (logs["other"] == logs["default"], n, var, val, max(rates["other"]), max(rates["default"])))
