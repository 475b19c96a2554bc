"""One-off: the benchmark's launches (512 pairs of config 3, two batches in flight, exact mode) for N seconds, every launch's poses
and costs compared on the device with the first launch's -- the same inputs must give the same bits whatever else is in flight
(races between lanes, workspace reuse, the striding redo kernels).   usage: python tests/campaigns/determinism_soak.py [seconds]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from ndtpso_slam_amd import capi, synth  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device("cuda", 0)
B = 512
p = synth.make_pairs(B, seed=2024)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
grid, cfg = capi.Grid(60, 60, 0.5), capi.PSOConfig.make(70, 70)
ctx = capi.Context(0)
stream = torch.cuda.current_stream(dev)
ctx.set_stream(stream.cuda_stream)
ctx.set_pipeline_depth(2)
d_ref, d_new = torch.from_numpy(p.ref_ranges).to(dev), torch.from_numpy(p.new_ranges).to(dev)
d_guess = torch.zeros(B, 3, dtype=torch.float64, device=dev)
d_dev = torch.tensor((0.1, 0.1, 3.1415e-3), dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
d_seeds = torch.from_numpy(p.seeds.astype(np.int64)).to(dev).to(torch.int32)
outs = [(torch.zeros(B, 3, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.float64, device=dev), torch.zeros(B, 8, dtype=torch.int32, device=dev)) for _ in range(4)]


def launch(k, mode):
    po, co, st = outs[k % 4]
    ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(), cfg, d_seeds.data_ptr(), 0, mode,
                        po.data_ptr(), co.data_ptr(), st.data_ptr())


launch(0, capi.SCORE_EXACT)
ctx.pipeline_flush(0)
torch.cuda.synchronize()
ref_pose, ref_cost = outs[0][0].clone(), outs[0][1].clone()
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < secs:
    for k in range(4):
        launch(k, capi.SCORE_EXACT)
    ctx.pipeline_flush(0)
    torch.cuda.synchronize()
    for k in range(4):
        if not (torch.equal(outs[k][0], ref_pose) and torch.equal(outs[k][1], ref_cost)):
            bad += 1
    n += 4
print("%d launches of %d alignments in %.0f s (%.0f alignments/s incl. the comparisons): %d launches differ from the first" % (n, B, time.time() - t0, n * B / (time.time() - t0), bad))
assert bad == 0
