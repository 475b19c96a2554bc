"""A few flagged pairs redone on clusters of workgroups (align_pairs_dev_on_stream, round 6) against the gated launches alone: eight
shapes x {fp64, exact}, ms per 512-pair launch and the results, one process per setting (the outputs go to gpurun_out/):

    NDTPSO_REDO_CLUSTERS=0 python scripts/redo_clusters_ab.py off; python scripts/redo_clusters_ab.py on    # "on" also compares
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndtpso_slam_amd import capi, synth
dev = torch.device("cuda", 0)
P = I = 70
cfg = capi.PSOConfig.make(I, P)
DEV = (0.1, 0.1, 3.1415e-3)
def run(nb, cs, mode, launches=6, env=None, B=512):
    for k, v in (env or {}).items():
        os.environ[k] = v
    ctx = capi.Context(0)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    p = synth.make_pairs(B, n_beams=nb, seed=2024)
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    grid = capi.Grid(60, 60, cs)
    d_ref, d_new = torch.from_numpy(p.ref_ranges).to(dev), torch.from_numpy(p.new_ranges).to(dev)
    d_guess = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_dev = torch.tensor(DEV, dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    d_seeds = torch.from_numpy(p.seeds.astype(np.int64)).to(dev).to(torch.int32)
    d_pose = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_cost = torch.zeros(B, dtype=torch.float64, device=dev)
    d_stats = torch.zeros(B, 8, dtype=torch.int32, device=dev)
    def launch():
        ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(), cfg,
                            d_seeds.data_ptr(), 0, mode, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
    launch()
    torch.cuda.synchronize()
    first = (d_pose.cpu().numpy().copy(), d_cost.cpu().numpy().copy(), d_stats.cpu().numpy().copy())
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / launches
    for k in (env or {}):
        del os.environ[k]
    return ms, (d_pose.cpu().numpy(), d_cost.cpu().numpy(), d_stats.cpu().numpy()), first
out = {}
tag = sys.argv[1]
for nb, cs, mname in ((1081, 0.25, "f64"), (1081, 0.25, "exact"), (1080, 0.25, "f64"), (1441, 0.25, "f64"), (1441, 0.3, "f64"), (721, 0.3, "f64"), (1081, 0.5, "f64"), (1081, 0.5, "exact")):
    mode = {"f64": capi.SCORE_F64, "exact": capi.SCORE_EXACT}[mname]
    ms, r, first = run(nb, cs, mode)
    key = "%d_%.2f_%s" % (nb, cs, mname)
    out[key + "_pose"], out[key + "_cost"], out[key + "_stats"], out[key + "_first"] = r[0], r[1], r[2], first[0]
    print("%s %s: %.3f ms  first call == later calls: %s  status or %x" % (tag, key, ms, np.array_equal(first[0], r[0]), int(np.bitwise_or.reduce(r[2][:, 5]))))
np.savez("gpurun_out/redo_%s.npz" % tag, **out)
if tag == "on":
    a, b = np.load("gpurun_out/redo_off.npz"), np.load("gpurun_out/redo_on.npz")
    for k in a.files:
        if not np.array_equal(a[k], b[k]):
            print("DIFFERENT:", k, np.nonzero(np.any((a[k] != b[k]).reshape(len(a[k]), -1), axis=1))[0][:10])
    print("compared", len(a.files), "arrays")
