"""Single-pair latency (clustered) and small-batch timing, device seeds vs host rand() tables, exact mode."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ndtpso_slam_amd import capi, synth
import ctypes
_libc = ctypes.CDLL(None)


def libc_rand(seed, n):
    """srand(seed); n x rand() -- the table the drop-in library hands over"""
    _libc.srand(ctypes.c_uint(int(seed)))
    return np.array([_libc.rand() for _ in range(n)], dtype=np.int32)


p = synth.make_pairs(8, seed=2024)
dev = torch.device("cuda", 0)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
grid = capi.Grid(60, 60, 0.5)
ctx = capi.Context(0); stream = torch.cuda.current_stream(dev); ctx.set_stream(stream.cuda_stream)
for (P, I) in ((70, 70), (30, 50)):
    cfg = capi.PSOConfig.make(I, P)
    n_draw = 3 + 3 * P + 6 * P * I
    for B in (1, 4):
        d_ref = torch.from_numpy(p.ref_ranges[:B]).to(dev); d_new = torch.from_numpy(p.new_ranges[:B]).to(dev)
        d_guess = torch.zeros(B, 3, dtype=torch.float64, device=dev)
        d_dev = torch.tensor((0.1, 0.1, 3.1415e-3), dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
        d_seeds = torch.from_numpy(p.seeds[:B].astype(np.int64)).to(dev).to(torch.int32)
        tabs = np.stack([libc_rand(int(sd), n_draw) for sd in p.seeds[:B]])
        d_tabs = torch.from_numpy(tabs).to(dev)
        d_pose = torch.zeros(B, 3, dtype=torch.float64, device=dev); d_cost = torch.zeros(B, dtype=torch.float64, device=dev)
        d_stats = torch.zeros(B, 8, dtype=torch.int32, device=dev)
        out = {}
        for name, seeds_ptr, tab_ptr in (("device generator", d_seeds.data_ptr(), 0), ("host table", 0, d_tabs.data_ptr())):
            lat = []
            for k in range(12):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(), cfg,
                                    seeds_ptr, tab_ptr, capi.SCORE_EXACT, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
                b.record(stream); torch.cuda.synchronize()
                if k >= 2: lat.append(a.elapsed_time(b))
            out[name] = (np.median(lat), d_pose.cpu().numpy().copy())
        same = np.array_equal(out["device generator"][1], out["host table"][1])
        print("%d x %d, %d pair(s), exact: device generator %.3f ms, host table %.3f ms, identical poses %s" % (P, I, B, out["device generator"][0], out["host table"][0], same))
