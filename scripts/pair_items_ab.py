"""Short scans: the kernels that score two particles per wave (k_align_pairs<..., PAIR>) against the one-item kernels, exact mode,
512 pairs of 181 / 361 / 541 beams x swarms of 8 ... 128 particles -- ms per launch both ways (NDTPSO_PAIR_MAX_BEAMS=4096 in the
environment takes the launcher's rule out of it) and whether the poses agree.  The rule in launch_pairs is fitted to this table.

    NDTPSO_PAIR_MAX_BEAMS=4096 python scripts/pair_items_ab.py        # GPU box
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
from ndtpso_slam_amd import capi, synth
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
stream = torch.cuda.current_stream(dev)
ctx.set_stream(stream.cuda_stream)
DEV = (0.1, 0.1, 3.1415e-3)
B = 512
def run(nb, P, I, launches=30):
    p = synth.make_pairs(B, n_beams=nb, seed=2024)
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    grid = capi.Grid(60, 60, 0.5)
    cfg = capi.PSOConfig.make(I, P)
    d_ref, d_new = torch.from_numpy(p.ref_ranges).to(dev), torch.from_numpy(p.new_ranges).to(dev)
    d_guess = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_dev = torch.tensor(DEV, dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    d_seeds = torch.from_numpy(p.seeds.astype(np.int64)).to(dev).to(torch.int32)
    d_pose = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_cost = torch.zeros(B, dtype=torch.float64, device=dev)
    d_stats = torch.zeros(B, 8, dtype=torch.int32, device=dev)
    def launch():
        ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(), cfg,
                            d_seeds.data_ptr(), 0, capi.SCORE_EXACT, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
    out = {}
    for tag, env in (("one", "0"), ("two", "1")):
        os.environ["NDTPSO_PAIR_ITEMS"] = env
        for _ in range(10):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            launch()
        e1.record()
        torch.cuda.synchronize()
        out[tag] = (e0.elapsed_time(e1) / launches, d_pose.cpu().numpy().copy())
    return out
for nb in (181, 361, 541):
    for P, I in ((8, 50), (16, 50), (17, 50), (24, 50), (30, 50), (48, 50), (70, 70), (128, 30)):
        o = run(nb, P, I)
        print("%d beams, %3d x %d: one item per wave %.3f ms, two %.3f ms (%+.1f %%), poses equal %s" % (
            nb, P, I, o["one"][0], o["two"][0], 100 * (o["one"][0] / o["two"][0] - 1), np.array_equal(o["one"][1], o["two"][1])), flush=True)
