"""Same-box A/B of the cluster placement on bench.py's own live sequence (C++ drop-in): NDTPSO_CLUSTER_SPREAD=1 against the default."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ndtpso_slam_amd import synth
for rep in range(3):
    for spread in ("1", "0"):
        os.environ["NDTPSO_CLUSTER_SPREAD"] = spread
        r = bench._live_sequence_cpp(synth, "exact")
        print("spread" if spread == "1" else "one_xcd", r.get("scans_per_s"), r.get("ms_per_scan"), flush=True)
