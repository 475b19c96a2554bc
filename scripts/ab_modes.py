"""A/B timing of the score modes inside ONE process, alternating, inputs resident in HBM (config 3 by default).
usage: python scripts/ab_modes.py [pairs] [rounds] [modes comma separated]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ndtpso_slam_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
names = (sys.argv[3] if len(sys.argv) > 3 else "f32,exact").split(",")
MODES = {"f32": capi.SCORE_F32, "exact": capi.SCORE_EXACT, "f64": capi.SCORE_F64}
P = int(os.environ.get("AB_P", "70")); I = int(os.environ.get("AB_I", "70"))
p = synth.make_pairs(B, seed=2024)
dev = torch.device("cuda", 0)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
grid = capi.Grid(60, 60, 0.5); cfg = capi.PSOConfig.make(I, P)
ctx = capi.Context(0); stream = torch.cuda.current_stream(dev); ctx.set_stream(stream.cuda_stream)
d_ref = torch.from_numpy(p.ref_ranges).to(dev); d_new = torch.from_numpy(p.new_ranges).to(dev)
d_guess = torch.zeros(B, 3, dtype=torch.float64, device=dev)
d_dev = torch.tensor((0.1, 0.1, 3.1415e-3), dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
d_seeds = torch.from_numpy(p.seeds.astype(np.int64)).to(dev).to(torch.int32)
d_pose = torch.zeros(B, 3, dtype=torch.float64, device=dev); d_cost = torch.zeros(B, dtype=torch.float64, device=dev)
d_stats = torch.zeros(B, 8, dtype=torch.int32, device=dev)
def run(mode, steps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(steps):
        ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(), cfg,
                            d_seeds.data_ptr(), 0, mode, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
    b.record(stream); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
for n in names: run(MODES[n], 10)
res = {n: [] for n in names}
for r in range(R):
    for n in names:
        res[n].append(run(MODES[n], 30))
for n in names:
    run(MODES[n], 30)
    st = d_stats.cpu().numpy().view(capi.STATS_DTYPE).reshape(B)
    dur = (st["t_end"] - st["t_start"]).astype(np.uint32) / 100.0
    start = (st["t_start"] - st["t_start"].min()).astype(np.uint32) / 100.0
    end = (st["t_end"] - st["t_start"].min()).astype(np.uint32) / 100.0
    print("%-6s steady state, last launch: wg duration us p50 %.0f p99 %.0f max %.0f; start spread %.0f; last finish %.0f; arbitrated mean %.2f max %d"
          % (n, np.percentile(dur, 50), np.percentile(dur, 99), dur.max(), start.max(), end.max(), st["arbitrated"].mean(), st["arbitrated"].max()))
    if B % 2 == 0:
        h = B // 2
        ea, eb = end[:h], end[h:]
        gap = eb - ea   # later-dispatched partner minus earlier one (assumed partners: b and b + B/2)
        print("       partners (b, b + B/2): finish gap us mean %.0f, |gap| mean %.0f p90 %.0f max %.0f; CU finish (max of the two) p50 %.0f max %.0f; evals corr with duration %.2f"
              % (gap.mean(), np.abs(gap).mean(), np.percentile(np.abs(gap), 90), np.abs(gap).max(), np.percentile(np.maximum(ea, eb), 50), np.maximum(ea, eb).max(),
                 np.corrcoef(dur, st["cost_evals"])[0, 1]))
for n in names:
    v = np.array(res[n])
    print("%-6s ms/launch: median %.4f  min %.4f  max %.4f   -> %.0f align/s" % (n, np.median(v), v.min(), v.max(), B / np.median(v) * 1e3))
