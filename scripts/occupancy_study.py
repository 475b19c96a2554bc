"""LDS cell-table sizing / occupancy study (BASELINE config 5 and neighbours), run on the GPU box:
    python scripts/occupancy_study.py > gpurun_out/occupancy.md
For each configuration: the plan the library picks (table form, LDS bytes, workgroups per CU) and the measured
throughput of the fused kernel on synthetic pairs (scans resident in HBM, events on the launch stream)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ndtpso_slam_amd import capi, synth  # noqa: E402

FORM = {0: "bitmap, division", 1: "bitmap, pow2", 2: "dense u16"}
DEV = (0.1, 0.1, 3.1415e-3)


def measure(ctx, n_beams, cs, P, I, B, mode, steps=3):
    p = synth.make_pairs(min(B, 64), n_beams=n_beams, seed=31)
    reps = (B + p.n_pairs - 1) // p.n_pairs
    ref = np.tile(p.ref_ranges, (reps, 1))[:B]
    new = np.tile(p.new_ranges, (reps, 1))[:B]
    seeds = np.tile(p.seeds, reps)[:B]
    dev = torch.device("cuda", 0)
    geom = capi.ScanGeom(n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    grid = capi.Grid(60, 60, cs)
    cfg = capi.PSOConfig.make(I, P)
    d_ref, d_new = torch.from_numpy(ref).to(dev), torch.from_numpy(new).to(dev)
    d_guess = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_dev = torch.tensor(DEV, dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    d_seeds = torch.from_numpy(seeds.astype(np.int64)).to(dev).to(torch.int32)
    d_pose = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_cost = torch.zeros(B, dtype=torch.float64, device=dev)
    d_stats = torch.zeros(B, 8, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)

    def go():
        ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(), cfg,
                            d_seeds.data_ptr(), 0, mode, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
    go()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(steps):
        go()
    b.record(stream)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    st = d_stats.cpu().numpy().view(capi.STATS_DTYPE).reshape(B)
    return ms, float(st["cost_evals"].mean()), int(st["n_built"].max()), int((st["status"] != 0).sum())


def main():
    ctx = capi.Context(0)
    rows = [  # n_beams, cell, P, I, pairs per launch
        (1081, 0.5, 70, 70, 512), (1081, 0.25, 70, 70, 512), (1081, 0.3, 70, 70, 512), (2048, 0.5, 70, 70, 512),
        (2048, 0.25, 70, 70, 512), (1081, 0.5, 256, 70, 256), (1081, 0.5, 512, 70, 256), (2048, 0.25, 2048, 20, 256),
        (2048, 0.25, 2048, 200, 256), (2048, 0.25, 2048, 200, 1),
    ]
    print("| beams | cell m | P x I | pairs | score | table form | window | LDS B/WG | threads/WG | WG/CU | swarm | ms/launch | align/s | evals/align |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for (n, cs, P, I, B) in rows:
        for mode, mname in ((capi.SCORE_EXACT, "exact"), (capi.SCORE_F32, "f32"), (capi.SCORE_F64, "f64")):
            if mode == capi.SCORE_F64 and (P > 512 and I > 20):
                continue
            geom = capi.ScanGeom(n, -2.356194, 4.712389 / (n - 1), 30.0, 0.1)
            rc, pl = capi.align_pairs_describe(geom, capi.Grid(60, 60, cs), capi.PSOConfig.make(I, P), mode, B)
            if rc != 0:
                print(f"| {n} | {cs} | {P}x{I} | {B} | {mname} | does not fit | | | | | | | | |")
                continue
            ms, evals, built, bad = measure(ctx, n, cs, P, I, B, mode, steps=1 if P * I > 100000 else 3)
            print(f"| {n} | {cs} | {P}x{I} | {B} | {mname} | {FORM[pl['table_form']]} | {pl['window_w']}x{pl['window_h']} | "
                  f"{pl['lds_bytes']} | {pl['block_threads']} | {pl['workgroups_per_cu']} | {'HBM' if pl['swarm_in_hbm'] else 'LDS'} | "
                  f"{ms:.2f} | {1e3 * B / ms:.0f} | {evals:.0f} |" + (" status!=0" if bad else ""))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
