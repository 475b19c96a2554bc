"""Timing of library variants per cost evaluation (diagnostic builds change the PSO's trajectory, hence its replays).
usage: NDTPSO_LIB=... python scripts/r2_ab_evals.py [pairs] [mode]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ndtpso_slam_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mode = {"f32": capi.SCORE_F32, "exact": capi.SCORE_EXACT, "f64": capi.SCORE_F64}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
p = synth.make_pairs(B, seed=2024)
dev = torch.device("cuda", 0)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
grid = capi.Grid(60, 60, 0.5); cfg = capi.PSOConfig.make(70, 70)
ctx = capi.Context(0); stream = torch.cuda.current_stream(dev); ctx.set_stream(stream.cuda_stream)
d_ref = torch.from_numpy(p.ref_ranges).to(dev); d_new = torch.from_numpy(p.new_ranges).to(dev)
d_guess = torch.zeros(B, 3, dtype=torch.float64, device=dev)
d_dev = torch.tensor((0.1, 0.1, 3.1415e-3), dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
d_seeds = torch.from_numpy(p.seeds.astype(np.int64)).to(dev).to(torch.int32)
d_pose = torch.zeros(B, 3, dtype=torch.float64, device=dev); d_cost = torch.zeros(B, dtype=torch.float64, device=dev)
d_stats = torch.zeros(B, 8, dtype=torch.int32, device=dev)
def run(steps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(steps):
        ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(), cfg,
                            d_seeds.data_ptr(), 0, mode, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
    b.record(stream); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
run(10)
ms = np.median([run(30) for _ in range(3)])
st = d_stats.cpu().numpy().view(capi.STATS_DTYPE).reshape(B)
ev = st["cost_evals"].astype(np.float64)
print("%s: %.4f ms/launch, evals/alignment mean %.0f max %.0f, status %s -> %.3f us per 1000 evals of the launch's longest alignment"
      % (os.environ.get("NDTPSO_LIB", "cur"), ms, ev.mean(), ev.max(), np.unique(st["status"] & 0xffff), 1e3 * ms / ev.max() * 1e3))
