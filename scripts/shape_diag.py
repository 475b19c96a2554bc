"""Why a shape of tests/campaigns/shape_sweep.py is slow: the first launch's flags (NDTPSO_NO_REDO=1 keeps the gated redo launches out)
and the time with and without them.   python scripts/shape_diag.py beams cell frame [mode]"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(nb, cs, fr, mode):
    import torch
    from ndtpso_slam_amd import capi, synth
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    B, P, I = 512, 70, 70
    p = synth.make_pairs(B, n_beams=nb, seed=2024)
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    grid, cfg = capi.Grid(fr, fr, cs), capi.PSOConfig.make(I, P)
    m = {"exact": capi.SCORE_EXACT, "f32": capi.SCORE_F32, "f64": capi.SCORE_F64}[mode]
    d_ref, d_new = torch.from_numpy(p.ref_ranges).to(dev), torch.from_numpy(p.new_ranges).to(dev)
    d_guess = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_dev = torch.tensor((0.1, 0.1, 3.1415e-3), dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    d_seeds = torch.from_numpy(p.seeds.astype(np.int64)).to(dev).to(torch.int32)
    d_pose = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_cost = torch.zeros(B, dtype=torch.float64, device=dev)
    d_stats = torch.zeros(B, 8, dtype=torch.int32, device=dev)

    def launch():
        ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(), cfg,
                            d_seeds.data_ptr(), 0, m, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(6):
        launch()
    b.record(stream)
    torch.cuda.synchronize()
    st = d_stats.cpu().numpy().view(capi.STATS_DTYPE).reshape(B)
    flags = st["status"] & 0xffff
    dur = (st["t_end"].astype(np.int64) - st["t_start"].astype(np.int64)) * 0.01
    print(json.dumps(dict(no_redo=os.environ.get("NDTPSO_NO_REDO"), ms=a.elapsed_time(b) / 6, flags={int(k): int((flags == k).sum()) for k in np.unique(flags)},
                          wg_us=dict(mean=float(dur.mean()), p95=float(np.percentile(dur, 95)), max=float(dur.max())),
                          evals=[int(st["cost_evals"].min()), int(st["cost_evals"].max())], n_built=[int(st["n_built"].min()), int(st["n_built"].max())],
                          plan=capi.align_pairs_describe(geom, grid, cfg, m, B)[1])))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    else:
        nb, cs, fr = sys.argv[1:4]
        mode = sys.argv[4] if len(sys.argv) > 4 else "exact"
        for no_redo in ("1", None):
            env = dict(os.environ)
            if no_redo:
                env["NDTPSO_NO_REDO"] = "1"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", nb, cs, fr, mode], env=env, capture_output=True, text=True)
            print(nb, cs, fr, mode, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:])
