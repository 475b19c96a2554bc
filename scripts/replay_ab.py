"""Same-box A/B of two builds of libndtpso_hip.so on the live sequence through the C++ drop-in (host/replay/node_replay):
runs interleaved, the pose logs must be byte-identical.
  usage: python scripts/replay_ab.py <dir holding the other libndtpso_hip.so> [n_scans] [repeats]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from live_timeline import write_scans
other = os.path.abspath(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 3
exe = os.path.join(ROOT, "host", "replay", "node_replay")
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "scans.bin")
    write_scans(path, n)
    logs, rates = {}, {"other": [], "this": []}
    for r in range(rep):
        for which in ("other", "this"):
            env = dict(os.environ, NDTPSO_RESIDENT="1", NDTPSO_SCORE=os.environ.get("NDTPSO_SCORE", "exact"))
            if which == "other":
                env["LD_LIBRARY_PATH"] = other + ":" + env.get("LD_LIBRARY_PATH", "")
            p = subprocess.run([exe, path, "60", "0.5", "50", "30", "7"], capture_output=True, text=True, timeout=300, env=env)
            m = re.search(r"matching rate: ([0-9.]+) Hz \(([0-9.]+) ms per scan\)", p.stderr)
            if p.returncode != 0 or not m:
                sys.exit("node_replay failed (%s): %s" % (which, p.stderr[-400:]))
            rates[which].append(float(m.group(1)))
            print("%-5s %s scans/s  %s ms/scan" % (which, m.group(1), m.group(2)), flush=True)
            logs.setdefault(which, p.stdout)
            if logs[which] != p.stdout:
                sys.exit("run-to-run difference in the pose log (%s)" % which)
    same = logs["other"] == logs["this"]
    print("pose logs identical: %s (%d scans); best other %.1f, best this %.1f scans/s" % (same, len(logs["this"].splitlines()), max(rates["other"]), max(rates["this"])))
    sys.exit(0 if same else 1)
