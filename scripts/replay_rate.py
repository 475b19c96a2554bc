import os, subprocess, sys, numpy as np
sys.path.insert(0, '.')
from ndtpso_slam_amd import synth
sys.path.insert(0, 'tests')
from test_host_library import _trajectory
n = 60
ranges, _ = _trajectory(n)
with open('/tmp/scans.bin', 'wb') as f:
    np.array([n, synth.N_BEAMS], dtype=np.int32).tofile(f)
    np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
    ranges.tofile(f)
for (I, P) in ((50, 30), (70, 70)):
    r = subprocess.run(['host/replay/node_replay', '/tmp/scans.bin', '60', '0.5', str(I), str(P), '7'], capture_output=True, text=True)
    print(P, 'x', I, r.stderr.strip())
