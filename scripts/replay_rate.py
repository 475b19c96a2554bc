"""Node replay (host/replay/node_replay) matching rate, resident vs host-bookkeeping frames, and optionally a
rocprofv3 kernel trace of the resident run.  usage: python scripts/replay_rate.py [n_scans] [--prof]"""
import os, subprocess, sys, numpy as np
sys.path.insert(0, '.')
from ndtpso_slam_amd import synth
sys.path.insert(0, 'tests')
from test_host_library import _trajectory
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
ranges, _ = _trajectory(n)
with open('/tmp/scans.bin', 'wb') as f:
    np.array([n, synth.N_BEAMS], dtype=np.int32).tofile(f)
    np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
    ranges.tofile(f)
cmd = ['host/replay/node_replay', '/tmp/scans.bin', '60', '0.5', '50', '30', '7']
for resident in ('1', '0'):
    for score in ('exact', 'f32', 'f64'):
        for og in ([], ['0.1', '/tmp/replay_dump', '10']):
            env = dict(os.environ, NDTPSO_RESIDENT=resident, NDTPSO_SCORE=score)
            r = subprocess.run(cmd + og, capture_output=True, text=True, env=env)
            print('resident', resident, score, 'og+dump' if og else 'plain', r.stderr.strip().splitlines()[-1])
if '--prof' in sys.argv:
    os.makedirs('gpurun_out/replay_prof', exist_ok=True)
    env = dict(os.environ, NDTPSO_RESIDENT='1', NDTPSO_SCORE=os.environ.get('REPLAY_PROF_SCORE', 'exact'), TMPDIR='/tmp')
    subprocess.run(['rocprofv3', '--kernel-trace', '--stats', '-d', 'gpurun_out/replay_prof', '-o', 'replay', '--output-format', 'csv', '--'] + cmd + ['0.1', '/tmp/replay_dump2', '10'],
                   env=env, capture_output=True, text=True)
    for root, _, files in os.walk('gpurun_out/replay_prof'):
        for fn in files:
            if fn.endswith('kernel_stats.csv'):
                print(open(os.path.join(root, fn)).read())
