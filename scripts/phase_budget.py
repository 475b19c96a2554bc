"""Per-phase time budget of the fused pairs kernel at BASELINE config 3 (512 pairs, 70 x 70, 1081 beams), from a
diagnostic build of the library (-DNDTPSO_PHASE_BUDGET: every workgroup accounts for its own time on the 100 MHz
real-time counter, contiguous marks -- ndtpso_kernels.hpp).

    python scripts/phase_budget.py --build                 # in the build container: hipcc the diagnostic library
    python scripts/phase_budget.py [--score exact|f32] [--out profiles/r06_phase_budget.json]     # on the GPU box

The budget of a launch:  kernel time (events) = dispatch offset (mean start of a workgroup after the first one)
+ mean workgroup duration (= the sum of its phases) + tail (the launch ends with its slowest workgroup).
The shipped library's launch time is measured in a second process for comparison (what the clocks themselves cost).
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUDGET_LIB = os.path.join(ROOT, "ndtpso_slam_amd", "lib", "libndtpso_hip_budget.so")   # ndtpso_slam_amd.build.variant_path("budget")

PHASES = ["setup (scan A, window, table, scan B)", "swarm initialisation (71 evaluations)", "generator at the top of an iteration",
          "proposals + barrier", "evaluation: wave 0's own item", "generator slice inside a round (wave 0)",
          "wait at the round's barrier (wave 0: the other waves' second item)", "arbitration (fp64 tasks)",
          "commits + gbest barriers", "end of iteration (prefetch, barrier)", "final fp64 cost + results"]


def build():
    from ndtpso_slam_amd import build as b
    print(b.build_variant("budget", verbose=True))


CONFIGS = {"config3": dict(B=512, P=70, I=70, beams=1081, cs=0.5, seed=2024, launches=12, warm=6),
           "beams361": dict(B=512, P=70, I=70, beams=361, cs=0.5, seed=2024, launches=12, warm=6),   # (a short scan: where does the time go?)
           "config5": dict(B=256, P=2048, I=200, beams=2048, cs=0.25, seed=21, launches=3, warm=1)}


def timed_launches(score, launches=12, config="config3"):
    """(mean ms per launch by events, last stats) through whatever library NDTPSO_LIB selects."""
    import numpy as np
    import torch
    from ndtpso_slam_amd import capi, synth
    dev = torch.device("cuda", 0)
    cf = CONFIGS[config]
    B, P, I, launches = cf["B"], cf["P"], cf["I"], cf["launches"]
    p = synth.make_pairs(B, n_beams=cf["beams"], seed=cf["seed"])
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    grid, cfg = capi.Grid(60, 60, cf["cs"]), capi.PSOConfig.make(I, P)
    mode = {"exact": capi.SCORE_EXACT, "f32": capi.SCORE_F32, "f64": capi.SCORE_F64}[score]
    ctx = capi.Context(0)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    d_ref, d_new = torch.from_numpy(p.ref_ranges).to(dev), torch.from_numpy(p.new_ranges).to(dev)
    d_guess = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_dev = torch.tensor((0.1, 0.1, 3.1415e-3), dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    d_seeds = torch.from_numpy(p.seeds.astype(np.int64)).to(dev).to(torch.int32)
    d_pose = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_cost = torch.zeros(B, dtype=torch.float64, device=dev)
    d_stats = torch.zeros(B, 8, dtype=torch.int32, device=dev)

    def launch():
        ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(), cfg,
                            d_seeds.data_ptr(), 0, mode, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
    for _ in range(cf["warm"]):
        launch()
    torch.cuda.synchronize()
    rows, ms, spans = [], [], []
    L = capi.load()
    import ctypes as C
    have = hasattr(L, "ndtpso_profile_phase_budget")
    for _ in range(launches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        launch()
        b.record(stream)
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
        st = d_stats.cpu().numpy().view(capi.STATS_DTYPE).reshape(B)
        t0, t1 = st["t_start"].astype(np.int64), st["t_end"].astype(np.int64)
        spans.append(dict(span_us=float((t1.max() - t0.min()) * 0.01), start_offset_us=float((t0 - t0.min()).mean() * 0.01),
                          mean_end_us=float((t1 - t0.min()).mean() * 0.01), dur_us=(t1 - t0) * 0.01))
        if have:
            out = np.zeros((B, 16), dtype=np.uint32)
            rc = L.ndtpso_profile_phase_budget(out.ctypes.data_as(C.c_void_p), B)
            assert rc == 0
            rows.append(out.astype(np.float64) * 0.01)   # microseconds
    return float(np.mean(ms)), spans, (np.stack(rows) if rows else None), st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--score", default="exact", choices=["exact", "f32", "f64"])
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_phase_budget.json"))
    ap.add_argument("--config", default="config3", choices=list(CONFIGS))
    ap.add_argument("--plain", action="store_true", help="(internal) time the shipped library and print the mean launch time")
    args = ap.parse_args()
    if args.build:
        build()
        return
    if args.plain:
        ms, spans, _, _ = timed_launches(args.score, config=args.config)
        print(json.dumps({"ms": ms, "span_us": float(sum(s["span_us"] for s in spans) / len(spans))}))
        return
    import numpy as np
    plain = json.loads(subprocess.check_output([sys.executable, os.path.abspath(__file__), "--plain", "--score", args.score, "--config", args.config],
                                               env={k: v for k, v in os.environ.items() if k != "NDTPSO_LIB"}).decode().strip().splitlines()[-1])
    os.environ["NDTPSO_LIB"] = BUDGET_LIB
    ms, spans, rows, st = timed_launches(args.score, config=args.config)
    assert rows is not None, "the library loaded is not a -DNDTPSO_PHASE_BUDGET build"
    ph = rows[:, :, :11]                       # [launch, workgroup, phase]
    wg_total = rows[:, :, 15]
    covered = ph.sum(axis=2)
    mean_phase = ph.mean(axis=(0, 1))
    mean_wg = float(wg_total.mean())
    offset = float(np.mean([s["start_offset_us"] for s in spans]))
    span = float(np.mean([s["span_us"] for s in spans]))
    mean_end = float(np.mean([s["mean_end_us"] for s in spans]))
    tail = span - mean_end
    kernel_us = 1e3 * ms
    budget = [{"phase": n, "mean_us_per_workgroup": float(v), "share_of_kernel_time": float(v / kernel_us)} for n, v in zip(PHASES, mean_phase)]
    accounted = float(mean_phase.sum()) + offset + tail
    out = {
        "what": "time budget of one launch of the fused pairs kernel, BASELINE %s (%d pairs, %d beams, %d particles x %d iterations, %.2f m cells), "
                "score mode %s; diagnostic build -DNDTPSO_PHASE_BUDGET, %d launches x %d workgroups averaged"
                % (args.config, CONFIGS[args.config]["B"], CONFIGS[args.config]["beams"], CONFIGS[args.config]["P"], CONFIGS[args.config]["I"],
                   CONFIGS[args.config]["cs"], args.score, rows.shape[0], rows.shape[1]),
        "kernel_ms_by_events_instrumented_build": ms,
        "kernel_ms_by_events_shipped_library": plain["ms"],
        "clock_overhead": ms / plain["ms"] - 1.0,
        "launch_span_us_by_device_counter": span,
        "budget_us": budget + [
            {"phase": "dispatch: mean start of a workgroup after the launch's first", "mean_us_per_workgroup": offset, "share_of_kernel_time": offset / kernel_us},
            {"phase": "tail: the launch ends with its slowest workgroup (span - mean end)", "mean_us_per_workgroup": tail, "share_of_kernel_time": tail / kernel_us}],
        "sum_of_budget_us": accounted,
        "sum_over_kernel_time": accounted / kernel_us,
        "workgroup_duration_us": {"mean": mean_wg, "p50": float(np.percentile(wg_total, 50)), "p95": float(np.percentile(wg_total, 95)),
                                  "max": float(wg_total.max()), "phases_cover": float((covered / wg_total).mean())},
        "seen_from_wave_1_two_items_per_round_us": {"own evaluations": float(rows[:, :, 12].mean()), "generator slice": float(rows[:, :, 13].mean()),
                                                    "wait at the round's barrier": float(rows[:, :, 14].mean())},
        "per_alignment": {"cost_evals_mean": float(st["cost_evals"].mean()), "rounds_mean": float(st["rounds"].mean()),
                          "arbitrated_mean": float(st["arbitrated"].mean())},
        "slowest_workgroups": {"corr(duration, arbitrated)": (float(np.corrcoef(wg_total[-1], st["arbitrated"].astype(float))[0, 1])
                                                              if st["arbitrated"].any() else None),
                               "corr(duration, cost_evals)": float(np.corrcoef(wg_total[-1], st["cost_evals"].astype(float))[0, 1])},
    }
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
