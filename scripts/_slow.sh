cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, tempfile, subprocess
sys.path.insert(0, os.getcwd())
import bench
from ndtpso_slam_amd import synth
d = tempfile.mkdtemp()
path = os.path.join(d, "scans.bin")
bench._write_live_scans(synth, path, 300)
for R in (16, 32):
    r = subprocess.run(["host/replay/node_replicas", path, "60", "0.5", "50", "30", "7", str(R)], capture_output=True, text=True,
                       env=dict(os.environ, NDTPSO_RESIDENT="1", NDTPSO_SCORE="exact", NODE_REPLICAS_SLOWEST="1"))
    print(r.stdout.strip()[-400:])
    print("\n".join(r.stderr.strip().splitlines()[-26:]))
PY
