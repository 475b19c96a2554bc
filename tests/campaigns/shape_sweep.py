"""Throughput of the fused pairs kernel OFF the benchmark's shapes (512 pairs, 70 x 70 PSO, every score mode asked for).

    python tests/campaigns/shape_sweep.py [--out profiles/r06_shape_sweep.json] [--modes exact,f64] [--quick]

For beams in {361, 541, 721, 1080, 1081, 1441, 2048} x cell side in {0.25, 0.3, 0.5, 1.0} m x frame in {60, 100, 300} m
(100 m: the node's default, include/ndtpso_slam_node.hpp:26; 300 m: launch/scan.launch:14): alignments per second (HIP
events around `--launches` launches, one at a time), the rate per POINT evaluation (alignments/s x evaluations x valid
points) as a fraction of the benchmark shape's (1081 beams, 0.5 m, 60 m), how the launch was planned (table form, LDS,
workgroups per compute unit), and the first 8 pairs' poses against the oracle.  The kernel's straight-line trips exist for 17
and 32 chunks of 64 points only (1081 / 2048 beams); everything else runs the general loop -- this is what that costs.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

BEAMS = [361, 541, 721, 1080, 1081, 1441, 2048]
CELLS = [0.25, 0.3, 0.5, 1.0]
FRAMES = [60, 100, 300]
DEV = (0.1, 0.1, 3.1415e-3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_shape_sweep.json"))
    ap.add_argument("--modes", default="exact")
    ap.add_argument("--pairs", type=int, default=512)
    ap.add_argument("--launches", type=int, default=8)
    ap.add_argument("--oracle-pairs", type=int, default=8)
    ap.add_argument("--beams", default="", help="comma-separated subset of the beam counts")
    ap.add_argument("--cells", default="", help="comma-separated subset of the cell sides")
    ap.add_argument("--frames", default="", help="comma-separated subset of the frame sizes")
    ap.add_argument("--quick", action="store_true", help="beams {541, 1081, 2048} x cells {0.3, 0.5} x frames {60, 100}")
    args = ap.parse_args()
    import torch
    from ndtpso_slam_amd import capi, synth
    from oracle import pyoracle

    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    B, P, I = args.pairs, 70, 70
    cfg = capi.PSOConfig.make(I, P)
    modes = {"exact": capi.SCORE_EXACT, "f32": capi.SCORE_F32, "f64": capi.SCORE_F64}
    beams_l, cells_l, frames_l = (BEAMS, CELLS, FRAMES) if not args.quick else ([541, 1081, 2048], [0.3, 0.5], [60, 100])
    if args.beams:
        beams_l = [int(b) for b in args.beams.split(",")]
    if args.cells:
        cells_l = [float(b) for b in args.cells.split(",")]
    if args.frames:
        frames_l = [int(b) for b in args.frames.split(",")]
    rows = []
    for nb in beams_l:
        p = synth.make_pairs(B, n_beams=nb, seed=2024)
        geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
        d_ref, d_new = torch.from_numpy(p.ref_ranges).to(dev), torch.from_numpy(p.new_ranges).to(dev)
        d_guess = torch.zeros(B, 3, dtype=torch.float64, device=dev)
        d_dev = torch.tensor(DEV, dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
        d_seeds = torch.from_numpy(p.seeds.astype(np.int64)).to(dev).to(torch.int32)
        d_pose = torch.zeros(B, 3, dtype=torch.float64, device=dev)
        d_cost = torch.zeros(B, dtype=torch.float64, device=dev)
        d_stats = torch.zeros(B, 8, dtype=torch.int32, device=dev)
        for cs in cells_l:
            for fr in frames_l:
                grid = capi.Grid(fr, fr, cs)
                k = args.oracle_pairs
                want, _, _ = pyoracle.align_pairs(p.ref_ranges[:k], p.new_ranges[:k], p.angle_min, p.angle_inc, p.range_max, 0.1,
                                                  fr, fr, cs, (0, 0, 0), DEV, pyoracle.PSOConfig.make(I, P), p.seeds[:k])
                for mname in args.modes.split(","):
                    mode = modes[mname]

                    def launch():
                        ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(), cfg,
                                            d_seeds.data_ptr(), 0, mode, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
                    for _ in range(3):
                        launch()
                    torch.cuda.synchronize()
                    # (the median of the launches' own times: one launch in a few hundred takes half as long again -- a host
                    # hiccup, a clock step -- and used to decide a whole row)
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.launches + 1)]
                    ev[0].record(stream)
                    for q in range(args.launches):
                        launch()
                        ev[q + 1].record(stream)
                    torch.cuda.synchronize()
                    ms = float(np.median([ev[q].elapsed_time(ev[q + 1]) for q in range(args.launches)]))
                    st = d_stats.cpu().numpy().view(capi.STATS_DTYPE).reshape(B)
                    pose = d_pose.cpu().numpy()
                    E = 1 + P + P * I
                    pt_evals = float(E) * float(st["n_points"].astype(np.float64).sum())
                    rc, plan = capi.align_pairs_describe(geom, grid, cfg, mode, B)
                    rows.append(dict(beams=nb, cell=cs, frame=fr, mode=mname, ms_per_launch=ms, align_per_s=B / ms * 1e3,
                                     point_evals_per_s=pt_evals / ms * 1e3, mean_valid_points=float(st["n_points"].mean()),
                                     chunks=int((int(st["n_points"].max()) + 63) // 64), flagged=int((st["status"] & 0xffff != 0).sum()),
                                     table_form=plan.get("table_form"), lds_bytes=plan.get("lds_bytes"), workgroups_per_cu=plan.get("workgroups_per_cu"),
                                     max_abs_dpose_vs_oracle=float(np.abs(pose[:k] - want).max())))
                    print(json.dumps(rows[-1]), flush=True)
    out = {"what": "fused pairs kernel, %d pairs x (70 particles x 70 iterations), one launch at a time (HIP events, median of %d launches); "
                   "rel = point evaluations per second relative to the benchmark shape's (1081 beams, 0.5 m cells, 60 m frame) in the same mode"
                   % (B, args.launches), "rows": rows}
    for mname in args.modes.split(","):
        base = [r for r in rows if r["mode"] == mname and r["beams"] == 1081 and r["cell"] == 0.5 and r["frame"] == 60]
        if base:
            for r in rows:
                if r["mode"] == mname:
                    r["rel"] = r["point_evals_per_s"] / base[0]["point_evals_per_s"]
            rel = [r["rel"] for r in rows if r["mode"] == mname]
            out.setdefault("summary", {})[mname] = dict(
                min_rel=min(rel), max_rel=max(rel), below_0_90=[(r["beams"], r["cell"], r["frame"], round(r["rel"], 3)) for r in rows
                                                                 if r["mode"] == mname and r["rel"] < 0.90],
                worst_dpose_vs_oracle=max(r["max_abs_dpose_vs_oracle"] for r in rows if r["mode"] == mname))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out.get("summary", {}), indent=1))


if __name__ == "__main__":
    main()
