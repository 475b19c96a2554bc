"""PCIe-inclusive rate of the host-buffer entry point (ndtpso_align_pairs) on the bench workload,
next to the device-resident one (ndtpso_align_pairs_dev) that bench.py reports.

    timeout 300 python scripts/host_buffer_rate.py [pairs] [repeats]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ndtpso_slam_amd import capi, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10

pairs = synth.make_pairs(B, seed=2024)
geom = capi.ScanGeom(pairs.n_beams, float(pairs.angle_min), float(pairs.angle_inc), float(pairs.range_max), 0.1)
grid = capi.Grid(60, 60, 0.5)
cfg = capi.PSOConfig.make(70, 70)
ctx = capi.Context(0)
guess = np.zeros((B, 3))
dev = np.tile(np.array([0.1, 0.1, 3.1415e-3]), (B, 1))
seeds = pairs.seeds.astype(np.uint32)


def host_call():
    return ctx.align_pairs(pairs.ref_ranges, pairs.new_ranges, geom, grid, guess, dev, cfg, seeds=seeds)


for _ in range(2):
    pose_h, _, _ = host_call()
t0 = time.perf_counter()
for _ in range(R):
    host_call()
t_host = (time.perf_counter() - t0) / R

d = torch.device("cuda", 0)
stream = torch.cuda.current_stream(d)
ctx.set_stream(stream.cuda_stream)
d_ref = torch.from_numpy(pairs.ref_ranges).to(d)
d_new = torch.from_numpy(pairs.new_ranges).to(d)
d_guess = torch.zeros(B, 3, dtype=torch.float64, device=d)
d_dev = torch.from_numpy(dev).to(d)
d_seeds = torch.from_numpy(seeds.astype(np.int64)).to(d).to(torch.int32)
d_pose = torch.zeros(B, 3, dtype=torch.float64, device=d)
d_cost = torch.zeros(B, dtype=torch.float64, device=d)
d_stats = torch.zeros(B, 8, dtype=torch.int32, device=d)


def dev_call():
    ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(), cfg,
                        d_seeds.data_ptr(), 0, capi.SCORE_F32, d_pose.data_ptr(), d_cost.data_ptr(),
                        d_stats.data_ptr())


for _ in range(2):
    dev_call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    dev_call()
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / R

same = np.array_equal(pose_h, d_pose.cpu().numpy())
h2d = pairs.ref_ranges.nbytes + pairs.new_ranges.nbytes + guess.nbytes + dev.nbytes + seeds.nbytes
print(f"pairs {B}: host buffers {t_host * 1e3:.3f} ms ({B / t_host:.0f} align/s), "
      f"device resident {t_dev * 1e3:.3f} ms ({B / t_dev:.0f} align/s), "
      f"H2D {h2d / 1e6:.2f} MB, poses identical {same}")
