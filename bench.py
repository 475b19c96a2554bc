#!/usr/bin/env python
"""bench.py -- scan-pair alignments/second of the fused gfx950 NDT-PSO path.

Workload (BASELINE.json config 3, per GPU): 512 synthetic Hokuyo-like scan pairs (1081 beams, 270 deg,
30 m, sigma 1 cm), reference frame 60 m x 60 m with 0.5 m NDT cells, PSO 70 particles x 70 iterations
replaying the reference's srand(seed) stream and single-thread update order.  One "step" = one pass of
the fused kernel over the rank's 512 pairs, scans already resident in HBM.  N > 1: one process per GPU,
pairs sharded by contiguous index range (weak scaling, 512 pairs/GPU), one RCCL all_gather of the poses
per step.

Score mode: `exact` by default -- fp32 Gaussian, every pbest / gbest comparison it cannot decide arbitrated in
fp64: the poses (and costs) of the all-fp64 mode, which are the oracle's, bit for bit (tests/test_gpu_fullsize.py
checks all 512 pairs of this very workload).  `--score f32` is the plain fp32 score (BASELINE config 2's "fp32":
a tolerance mode, 1 of 4096 pairs leaves the reference's trajectory), `--score f64` the fp64 score throughout;
both are timed as `extra.modes` of the default run.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      bound "valu": the kernel keeps table, points and swarm in LDS, HBM sees 8.7 KB per alignment; what
                bounds it is the chip's vector ALU.  achieved = SURVEY 8(d)'s flops per point evaluation (16 fp64 + 34
                fp32 + 1 exp) x the algorithmic point evaluations of a launch / the launch's duration (events, one
                launch at a time); peak = the same flop mix at the vector peaks of the chip (fp32 157.3, fp64 78.6
                TFLOP/s).  roofline.floor keeps the issue-time figure of the score loop's own instruction mix
                (scripts/isa_mix.py); the 40-byte-per-point-eval "effective bandwidth" of SURVEY 8(d) is under
                roofline.hbm_effective.
  value         steps / time with two batches in flight in the one context (ndtpso_set_pipeline_depth(2): step k + 1 is
                launched while step k's slowest alignments finish); extra.batches_in_flight.one_batch_at_a_time is the
                figure of rounds 1-2 (--pipeline 1 makes it the headline).
  cpu_baseline  the oracle (a port of the reference's algorithm) on this box's host cores, SURVEY 8(d)'s four numbers.
  clock settle  a process that has just started finds the device in a low power state: before the `warmup` steps the same launches
                run, untimed, for --settle-ms of wall time (default 400; extra.clock_settle says how many), and again before the
                one-at-a-time measurement that follows the timed region.  The timed region itself is exactly `steps` steps.
                Without it `--steps 20 --warmup 5` read 263-267 k (one at a time 230 k) where `--steps 200` reads 277 k (248 k)
                on the same box in the same minute; with it 272-274 k (248 k).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMD = 256 * 4               # 256 CUs x 4 SIMDs
BYTES_PER_POINT_EVAL = 40.0    # 16 B fp64 point + 24 B compact cell record
FRAME_M, CELL_SIDE = 60, 0.5
DEVIATION = (0.1, 0.1, 3.1415e-3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 200 x 2.7 ms: a timed region of more than half a second
    ap.add_argument("--warmup", type=int, default=20)   # (the first steps run before the clocks settle)
    ap.add_argument("--pairs", type=int, default=512, help="scan pairs per GPU per step")
    ap.add_argument("--particles", type=int, default=70)
    ap.add_argument("--iterations", type=int, default=70)
    ap.add_argument("--score", choices=["exact", "f32", "f64"], default="exact")
    ap.add_argument("--settle-ms", type=float, default=400., help="untimed launches for this long before the warm-up steps (device clocks)")
    ap.add_argument("--pipeline", type=int, choices=[1, 2], default=2,
                    help="batches in flight in the context (ndtpso_set_pipeline_depth): 2 = step k + 1 is launched while step k's "
                         "slowest alignments finish (the default, what a caller with a queue of batches does); 1 = one at a time")
    ap.add_argument("--cpu-sample", type=int, default=512, help="pairs of this step timed on the host oracle (0 = skip the CPU baseline)")
    ap.add_argument("--no-latency", action="store_true", help="skip the extras (other score modes, config 5, latency, live sequence)")
    ap.add_argument("--identical", action="store_true", help="diagnostic: replicate pair 0 (no load imbalance)")
    ap.add_argument("--sharded-capi", action="store_true",
                    help="ONE process driving --gpus N devices through the C-ABI's ndtpso_align_pairs_sharded_dev (a host thread, "
                         "context and stream per device, ncclCommInitAll, one ncclAllGather per step) instead of one process per GPU")
    args = ap.parse_args()

    # How this process was started decides what it is (launch_plan): a rank of a launcher's group, the parent that has to
    # start the group itself, the one process of --sharded-capi -- or a mistake, which ends here with a reason and exit code 2:
    # a run that asks for N GPUs never prints a number measured on fewer.
    import torch
    kind, detail = launch_plan(args.gpus, args.sharded_capi, os.environ, torch.cuda.device_count() if torch.cuda.is_available() else 0,
                               sys.argv[1:])
    if kind == "fail":
        print("bench.py: " + detail, file=sys.stderr, flush=True)
        raise SystemExit(2)
    if kind == "relaunch":
        print("bench.py: --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(detail)), file=sys.stderr, flush=True)
        os.environ["NDTPSO_BENCH_SELF_LAUNCHED"] = "1"
        os.execv(detail[0], detail)

    # stdout carries exactly one line, the JSON result: everything else that may print there on the way (RCCL's
    # version banner, device-side printf of diagnostic builds) is sent to stderr at the file-descriptor level
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist

    if kind == "sharded":
        out = _main_sharded(args, torch)
        _print_result(out, stdout_fd)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NDTPSO_BENCH_FORCE_DIST=1: take the RCCL path (process group, all_gather, barriers) even with one rank --
    # lets a single-GPU box exercise the code the multi-GPU runs use
    use_dist = world > 1 or os.environ.get("NDTPSO_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from ndtpso_slam_amd import capi, sharding, synth

    B, P, I = args.pairs, args.particles, args.iterations
    mode = {"f32": capi.SCORE_F32, "f64": capi.SCORE_F64, "exact": capi.SCORE_EXACT}[args.score]
    first, last = sharding.shard_range(world * B, rank, world)   # weak scaling: B pairs per GPU
    pairs = synth.make_pairs(last - first, seed=2024, first_pair=first, total_pairs=world * B)
    if args.identical:
        pairs.ref_ranges[:] = pairs.ref_ranges[0]
        pairs.new_ranges[:] = pairs.new_ranges[0]
        pairs.seeds[:] = pairs.seeds[0]
        pairs.delta[:] = pairs.delta[0]
    geom = capi.ScanGeom(pairs.n_beams, float(pairs.angle_min), float(pairs.angle_inc), float(pairs.range_max), 0.1)
    grid = capi.Grid(FRAME_M, FRAME_M, CELL_SIDE)
    cfg = capi.PSOConfig.make(I, P)

    ctx = capi.Context(local_rank)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    # the exact mode's once-per-process self check (ndtpso_exact_check, ~40 ms) now, whatever --warmup is: never inside the timed region
    exact_check = ctx.exact_check() if mode == capi.SCORE_EXACT else None

    d_ref = torch.from_numpy(pairs.ref_ranges).to(dev)
    d_new = torch.from_numpy(pairs.new_ranges).to(dev)
    d_guess = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_dev = torch.tensor(DEVIATION, dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    d_seeds = torch.from_numpy(pairs.seeds.astype(np.int64)).to(dev).to(torch.int32)  # bit pattern of uint32 < 2^31
    d_pose = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d_cost = torch.zeros(B, dtype=torch.float64, device=dev)
    d_stats = torch.zeros(B, 8, dtype=torch.int32, device=dev)

    # Two output sets: with two batches in flight, step k writes set k & 1 while step k - 1's slowest alignments finish.
    outs = [(d_pose, d_cost, d_stats), (torch.zeros_like(d_pose), torch.zeros_like(d_cost), torch.zeros_like(d_stats))]
    depth = args.pipeline
    ctx.set_pipeline_depth(depth)

    def launch(k):
        po, co, st = outs[k & 1] if depth > 1 else outs[0]
        ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(),
                            cfg, d_seeds.data_ptr(), 0, mode, po.data_ptr(), co.data_ptr(), st.data_ptr())

    def run_steps(n, events=None):
        """n steps.  One step = one launch over the rank's B pairs + (N > 1) the single RCCL gather of its poses.  With two
        batches in flight the gather of step k - 1 is enqueued behind the launch of step k (ndtpso_pipeline_flush(1) orders
        step k - 1's poses before it on the stream), the last one after the loop: n launches, n gathers either way."""
        for k in range(n):
            if events is not None:
                events[k][0].record(stream)
            launch(k)
            if events is not None:
                events[k][1].record(stream)
            if use_dist:
                if depth == 1:
                    sharding.gather_poses(outs[0][0], equal_sizes=True, force=True)  # the single RCCL gather of poses over xGMI
                elif k > 0:
                    ctx.pipeline_flush(1)
                    sharding.gather_poses(outs[(k - 1) & 1][0], equal_sizes=True, force=True)
        if depth > 1:
            ctx.pipeline_flush(0)
            if use_dist and n > 0:
                sharding.gather_poses(outs[(n - 1) & 1][0], equal_sizes=True, force=True)

    # Clock settle: a process that has just started finds the device in a low power state, and the first ~0.3 s of launches run
    # 4-7 % slower than the steady state (measured: --steps 20 --warmup 5 read 232 k align/s one at a time where --steps 200
    # read 250 k, same box, same minute).  The same launches, untimed, for --settle-ms of wall time before the warm-up steps;
    # the timed region below is still exactly `steps` steps.  --settle-ms 0 switches it off.
    # Before it, once: what rounds 1-4 measured -- `warmup` + `steps` launches at the configured depth straight after start-up,
    # then min(steps, 60) one at a time -- so that the driver's line stays comparable across rounds (extra.without_clock_settle).
    cold = None
    if args.settle_ms > 0 and world == 1:
        cold = {}
        for k in range(args.warmup):
            launch(k)
        if depth > 1:
            ctx.pipeline_flush(0)
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        for k in range(args.steps):
            launch(k)
        if depth > 1:
            ctx.pipeline_flush(0)
        torch.cuda.synchronize()
        cold["alignments_per_s_at_depth_%d" % depth] = B * args.steps / (time.perf_counter() - t_c)
        ctx.set_pipeline_depth(1)
        for k in range(2):
            launch(0)
        torch.cuda.synchronize()
        nc = max(1, min(args.steps, 60))
        t_c = time.perf_counter()
        for k in range(nc):
            launch(0)
        torch.cuda.synchronize()
        cold["alignments_per_s_one_at_a_time"] = B * nc / (time.perf_counter() - t_c)
        cold["untimed_launches_before"] = args.warmup
        cold["note"] = "the procedure of rounds 1-4: warmup + steps launches right after start-up, no clock settle; then one at a time"
        ctx.set_pipeline_depth(depth)
    cold_launches = (args.warmup + args.steps + 2 + max(1, min(args.steps, 60))) if cold is not None else 0
    settle_launches = 0
    if args.settle_ms > 0:
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < args.settle_ms:
            for k in range(4):   # (launches only: how many fit in the time differs from rank to rank, a collective here would hang)
                launch(k)
            if depth > 1:
                ctx.pipeline_flush(0)
            torch.cuda.synchronize()
            settle_launches += 4
    run_steps(args.warmup)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()

    # the timed region: exactly `steps` steps between two synchronisations
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    run_steps(args.steps, ev if depth == 1 else None)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ctx.synchronize()
    poses_equal_between_sets = bool(torch.equal(outs[0][0], outs[1][0])) if depth > 1 and args.steps > 1 else None

    # Per-launch duration of the kernel, one launch at a time (events on the launch stream): what the rocprofv3 kernel
    # trace of `bench.py --pipeline 1` averages, and what the roofline divides by.  With two batches in flight a
    # launch's own duration says nothing (two launches share the device); the serial figure is measured here, after
    # the timed region, over min(steps, 60) launches.
    ctx.set_pipeline_depth(1)
    if depth == 1:
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else float("nan")
        serial_ms = 1e3 * elapsed / max(args.steps, 1)
    else:
        ns = max(1, min(args.steps, 60))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ns)]
        # (untimed first: the context's own work buffers -- the lanes had theirs -- are allocated by the first serial call, ~2 ms;
        # and the device's clocks follow the change of regime with a lag -- right behind the two-in-flight region the same serial
        # launches measured 3.5 % slower than in a `--pipeline 1` run on the same box)
        t_s = time.perf_counter()
        while True:
            for _ in range(2):
                ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(),
                                    cfg, d_seeds.data_ptr(), 0, mode, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
            torch.cuda.synchronize()
            if (time.perf_counter() - t_s) * 1e3 >= args.settle_ms:
                break
        t1 = time.perf_counter()
        for k in range(ns):
            evs[k][0].record(stream)
            ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(),
                                cfg, d_seeds.data_ptr(), 0, mode, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
            evs[k][1].record(stream)
        torch.cuda.synchronize()
        serial_ms = 1e3 * (time.perf_counter() - t1) / ns
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    stats = d_stats.cpu().numpy().view(capi.STATS_DTYPE).reshape(B)
    pose = d_pose.cpu().numpy()
    n_valid = stats["n_points"].astype(np.float64)
    evals_nominal = 1 + P + P * I                      # cost evaluations the reference performs per alignment
    algo_bytes = float(n_valid.sum()) * evals_nominal * BYTES_PER_POINT_EVAL   # per launch (one step, this rank)
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else float("nan")

    out = None
    if rank == 0:
        value = (B * world * args.steps) / elapsed
        out = {
            "metric": "scan alignments/sec (1081-beam, 70 particles x 70 iters)",
            "value": value,
            "unit": "alignments/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            # every launch this process issued before the timed region: the cold measurement of extra.without_clock_settle, the
            # clock settle (--settle-ms of wall time) and the `warmup` steps
            "untimed_launches_total": cold_launches + settle_launches + args.warmup,
            "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "value_one_batch_at_a_time": B * world / (serial_ms * 1e-3) if world == 1 else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"exact": "f64 transform/index + f32 score, undecidable comparisons arbitrated in f64 (= the f64 mode's poses)",
                      "f32": "f64 transform/index + f32 score", "f64": "f64"}[args.score],
            "data": "synthetic",
            "config": {
                "workload": "BASELINE config 3: %d scan pairs per GPU per step, %d beams, %.2f m cells, %d m frame, "
                            "PSO %d particles x %d iterations, exact reference order + srand(seed) stream"
                            % (B, pairs.n_beams, CELL_SIDE, FRAME_M, P, I),
                "pairs_per_gpu": B, "particles": P, "iterations": I, "score": args.score,
                "parallelism": "pairs sharded by contiguous index range, 1 RCCL all_gather of poses per step"
                               if world > 1 else "single GPU",
                "batches_in_flight": depth,
            },
            "roofline": _roofline(stats, evals_nominal, kern_ms, algo_bytes, achieved, 1e3 * elapsed / max(args.steps, 1), depth, args.score),
            "extra": {
                "mean_cost_evals_per_alignment": float(stats["cost_evals"].mean()),
                "mean_replay_overhead": float(stats["cost_evals"].mean()) / evals_nominal - 1.0,
                "mean_abs_err_vs_truth": np.abs(pose - pairs.delta).mean(axis=0).tolist(),
                "status_nonzero": int((stats["status"] != 0).sum()),
                "cost_evals_min_max": [int(stats["cost_evals"].min()), int(stats["cost_evals"].max())],
                "rounds_min_max": [int(stats["rounds"].min()), int(stats["rounds"].max())],
                "n_built_min_max": [int(stats["n_built"].min()), int(stats["n_built"].max())],
                "n_points_min_max": [int(stats["n_points"].min()), int(stats["n_points"].max())],
                "comparisons_arbitrated_in_f64_per_alignment": float(stats["arbitrated"].mean()),
                "clock_settle": {"ms": args.settle_ms, "untimed_launches_before_the_warmup_steps": settle_launches},
                "without_clock_settle": cold,
                "exact_mode_start_up_check": exact_check,   # state 1: passed (ndtpso_exact_check; before the warm-up)
                "timed_region_s": elapsed,
                "batches_in_flight": {
                    "depth": depth,
                    "note": "value / ms_per_step: `steps` launches of one context issued back to back with ndtpso_set_pipeline_depth(%d); "
                            "one_batch_at_a_time: the same launches one after the other on one stream (what rounds 1-2 reported)" % depth,
                    "one_batch_at_a_time": {"alignments_per_s": B * world / (serial_ms * 1e-3) if world == 1 else None,
                                            "ms_per_step": serial_ms, "kernel_ms_by_events": kern_ms},
                    "poses_equal_between_the_two_output_sets": poses_equal_between_sets,
                },
            },
        }

    # (before rank 0's extras: the other ranks take part in these collectives and must not sit in one for minutes)
    # Did RCCL see N ranks?  Answerable from the record: a real all_gather of the rank numbers (ranks_seen), the gathered
    # poses held against every rank's own (block r of the gather == rank r's poses, checked by rank r, AND over ranks), the
    # library's version, and the median duration of the step's one collective by events on the launch stream.
    if use_dist:
        ids = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([rank], dtype=torch.int64, device=dev))
        ranks_seen = len({int(t.item()) for t in ids})
        mine = outs[(args.steps - 1) & 1][0] if depth > 1 and args.steps > 0 else outs[0][0]
        allp = sharding.gather_poses(mine, equal_sizes=True, force=True)
        own = torch.tensor([1 if torch.equal(allp[rank * B:(rank + 1) * B], mine) and allp.shape[0] == world * B else 0], device=dev)
        dist.all_reduce(own, op=dist.ReduceOp.MIN)
        g_us = []
        for _ in range(25):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            sharding.gather_poses(mine, equal_sizes=True, force=True)
            b.record(stream)
            torch.cuda.synchronize()
            g_us.append(1e3 * a.elapsed_time(b))
        if rank == 0:
            try:
                ver = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:  # noqa: BLE001
                ver = None
            out["rccl"] = {"ranks_seen": ranks_seen, "world_size": dist.get_world_size(), "backend": dist.get_backend(),
                           "version": ver, "gather_us_median": float(np.median(g_us[5:])),
                           "gather_bytes_per_rank": int(B * 24), "every_rank_found_its_poses_in_the_gather": bool(int(own.item())),
                           "launch": "one process per GPU (torch.distributed), %s" % ("started by bench.py itself" if os.environ.get("NDTPSO_BENCH_SELF_LAUNCHED") == "1" else "started by a launcher")}
            assert ranks_seen == world == args.gpus, (ranks_seen, world, args.gpus)
    elif rank == 0:
        out["rccl"] = None   # one rank, no process group: nothing was gathered (NDTPSO_BENCH_FORCE_DIST=1 takes the RCCL path anyway)


    # The extras (single-pair latency, the other score modes, config 5, the live sequences): rank 0 of a ONE-GPU run only.  With
    # N > 1 the other ranks would sit in the closing barrier for as long as rank 0 takes over them -- minutes of a collective in
    # flight on N - 1 devices for figures the N = 1 line of the same driver pass already carries.
    extras = rank == 0 and world == 1 and not args.no_latency
    # single-pair latency (BASELINE config 2)
    if extras:
        torch.cuda.synchronize()
        lat = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            ctx.align_pairs_dev(1, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(),
                                d_dev.data_ptr(), cfg, d_seeds.data_ptr(), 0, mode, d_pose.data_ptr(),
                                d_cost.data_ptr(), d_stats.data_ptr())
            b.record(stream)
            torch.cuda.synchronize()
            lat.append(a.elapsed_time(b))
        out["extra"]["single_pair_latency_ms"] = float(np.median(lat))
        out["extra"]["single_pair_latency_note"] = "one pair of this workload, %s score, spread over a cluster of workgroups" % args.score

    # the other score modes on the same workload and BASELINE config 5, timed here so that they are driver-visible
    if extras:
        def timed(n_pairs, m, steps, geom_, grid_, cfg_, ref_, new_, seeds_):
            for _ in range(2):
                ctx.align_pairs_dev(n_pairs, ref_.data_ptr(), new_.data_ptr(), geom_, grid_, d_guess.data_ptr(), d_dev.data_ptr(),
                                    cfg_, seeds_.data_ptr(), 0, m, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                ctx.align_pairs_dev(n_pairs, ref_.data_ptr(), new_.data_ptr(), geom_, grid_, d_guess.data_ptr(), d_dev.data_ptr(),
                                    cfg_, seeds_.data_ptr(), 0, m, d_pose.data_ptr(), d_cost.data_ptr(), d_stats.data_ptr())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            return {"alignments_per_s": n_pairs * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "pairs": n_pairs}
        def timed_two_in_flight(m, steps):
            """the headline's regime for another score mode: launches issued back to back at pipeline depth 2, alternating
            output sets, flushed before the clock stops"""
            ctx.set_pipeline_depth(2)
            def go(n):
                for k in range(n):
                    po, co, st = outs[k & 1]
                    ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, grid, d_guess.data_ptr(), d_dev.data_ptr(),
                                        cfg, d_seeds.data_ptr(), 0, m, po.data_ptr(), co.data_ptr(), st.data_ptr())
                ctx.pipeline_flush(0)
                torch.cuda.synchronize()
            go(4)
            t1 = time.perf_counter()
            go(steps)
            dt = time.perf_counter() - t1
            ctx.set_pipeline_depth(1)
            return {"alignments_per_s": B * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps}
        modes = {"note": "alignments_per_s / ms_per_step: one 512-pair launch at a time; two_in_flight: the regime of `value`"}
        for name, m, steps in (("f32", capi.SCORE_F32, 300), ("f64", capi.SCORE_F64, 150), ("exact", capi.SCORE_EXACT, 300)):
            if m != mode:
                modes[name] = timed(B, m, steps, geom, grid, cfg, d_ref, d_new, d_seeds)
                modes[name]["two_in_flight"] = timed_two_in_flight(m, steps)
        out["extra"]["modes"] = modes
        try:
            B5 = min(256, B)
            p5 = synth.make_pairs(B5, n_beams=2048, seed=21)
            g5 = capi.ScanGeom(p5.n_beams, float(p5.angle_min), float(p5.angle_inc), float(p5.range_max), 0.1)
            r5, n5 = torch.from_numpy(p5.ref_ranges).to(dev), torch.from_numpy(p5.new_ranges).to(dev)
            s5 = torch.from_numpy(p5.seeds.astype(np.int64)).to(dev).to(torch.int32)
            c5 = {}
            for name, m in (("exact", capi.SCORE_EXACT), ("f32", capi.SCORE_F32)):
                c5[name] = timed(B5, m, 2, g5, capi.Grid(FRAME_M, FRAME_M, 0.25), capi.PSOConfig.make(200, 2048), r5, n5, s5)
            c5["workload"] = "BASELINE config 5: %d pairs per launch, 2048 particles x 200 iterations, 2048 beams, 0.25 m cells" % B5
            out["extra"]["config5"] = c5
        except Exception as e:  # noqa: BLE001 -- an extra, never the reason a bench run fails
            out["extra"]["config5"] = {"error": str(e)}
        try:
            # a short scan (361 beams: the kernels that score two particles per wave), same swarm, same cells
            ps_ = synth.make_pairs(B, n_beams=361, seed=2024)
            gs_ = capi.ScanGeom(ps_.n_beams, float(ps_.angle_min), float(ps_.angle_inc), float(ps_.range_max), 0.1)
            rs_, ns_ = torch.from_numpy(ps_.ref_ranges).to(dev), torch.from_numpy(ps_.new_ranges).to(dev)
            ss_ = torch.from_numpy(ps_.seeds.astype(np.int64)).to(dev).to(torch.int32)
            short = timed(B, mode, 200, gs_, grid, cfg, rs_, ns_, ss_)
            short["workload"] = "%d pairs per launch, %d x %d, 361 beams, %.2f m cells, %s score, one launch at a time" % (B, P, I, CELL_SIDE, args.score)
            short["rate_per_point_evaluation_vs_this_run_one_at_a_time"] = (short["alignments_per_s"] * 361.0) / (B / (serial_ms * 1e-3) * float(n_valid.mean()))
            out["extra"]["short_scans_361_beams"] = short
        except Exception as e:  # noqa: BLE001
            out["extra"]["short_scans_361_beams"] = {"error": str(e)}

    # the node's live sequence: loadLaser -> align -> update against an accumulating resident map,
    # default 30 x 50 PSO, rand() table from the host as the drop-in library passes it; rank 0 only, 60 scans
    if extras:
        try:
            out["extra"]["live_sequence"] = _live_sequence(ctx, capi, synth, mode)
        except Exception as e:  # noqa: BLE001 -- an extra, never the reason a bench run fails
            out["extra"]["live_sequence"] = {"error": str(e)}
        try:
            out["extra"]["live_sequence_cpp_drop_in"] = _live_sequence_cpp(synth, args.score)
            # ... and at the node's own default frame (100 m x 100 m, include/ndtpso_slam_node.hpp:25-26)
            out["extra"]["live_sequence_cpp_drop_in_node_default_frame"] = _live_sequence_cpp(synth, args.score, frame=100)
        except Exception as e:  # noqa: BLE001
            out["extra"]["live_sequence_cpp_drop_in"] = {"error": str(e)}
        try:
            # R replicas of that sequence in one process, a host thread + device context + stream each (the
            # live stream does not shard -- "replicas only"): what one GPU delivers when it serves R robots
            out["extra"]["live_replicas"] = [_live_replicas(synth, args.score, R) for R in (1, 4, 16, 32)]
        except Exception as e:  # noqa: BLE001
            out["extra"]["live_replicas"] = {"error": str(e)}

    # CPU baseline: the oracle -- a port of the reference's algorithm -- on this box's host cores
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        out["cpu_baseline"] = _cpu_baseline(pairs, min(args.cpu_sample, B), P, I)
        opose = out["cpu_baseline"].pop("_poses")
        out["extra"]["parity_max_abs_dpose_vs_oracle"] = np.abs(pose[:len(opose)] - opose).max(axis=0).tolist()
        out["extra"]["parity_pairs_compared"] = int(len(opose))
    elif rank == 0:
        out["cpu_baseline"] = None

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    _print_result(out if rank == 0 else None, stdout_fd)


def _print_result(out, stdout_fd):
    sys.stdout.flush()
    try:  # libc's own stdout buffer (native libraries printf into it) must drain while fd 1 still points at stderr
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    os.dup2(stdout_fd, 1)
    os.close(stdout_fd)
    if out is not None:
        print(json.dumps(out), flush=True)


def _free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return int(so.getsockname()[1])


def relaunch_command(argv, n_gpus, port=None):
    """What `bench.py --gpus N` (N > 1) started WITHOUT a launcher executes in its own place: the launcher line of the
    task's contract, one rank per GPU over RCCL, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port) if port else _free_port()),
            os.path.abspath(__file__)] + list(argv)


def launch_plan(gpus, sharded, environ, n_devices, argv):
    """-> (kind, detail).  kind: 'run' (this process is a rank -- or the only one), 'relaunch' (detail = the command to exec),
    'sharded' (the one process of --sharded-capi), 'fail' (detail = the reason; the caller exits with code 2).
    Rules: --gpus N must be what runs.  WORLD_SIZE set (a launcher started us): it must equal N, and LOCAL_RANK must name a
    visible device.  WORLD_SIZE unset and N > 1: N devices must be visible and this process re-executes itself under
    python -m torch.distributed.run.  --sharded-capi: one process, N devices visible -- unless NDTPSO_SHARD_VIRTUAL=N (the
    shard group's test mode: N shards on the visible devices, gather staged through the host; flagged in the result)."""
    if gpus < 1:
        return "fail", "--gpus %d: at least one" % gpus
    ws = environ.get("WORLD_SIZE")
    if sharded:
        if ws is not None and int(ws) != 1:
            return "fail", "--sharded-capi is ONE process for all devices; started under a launcher with WORLD_SIZE=%s" % ws
        virt = int(environ.get("NDTPSO_SHARD_VIRTUAL") or 0)
        if virt and virt != gpus:
            return "fail", "NDTPSO_SHARD_VIRTUAL=%d but --gpus %d" % (virt, gpus)
        if n_devices < (1 if virt else gpus):
            return "fail", "--sharded-capi --gpus %d but %d HIP device(s) visible" % (gpus, n_devices)
        return "sharded", None
    if ws is None:
        if n_devices < gpus:
            return "fail", "--gpus %d but %d HIP device(s) visible" % (gpus, n_devices)
        if gpus == 1 and environ.get("NDTPSO_BENCH_FORCE_DIST") != "1":
            return "run", None
        # (NDTPSO_BENCH_FORCE_DIST=1 with --gpus 1: the same re-execution with one rank -- a one-GPU box runs the very path, exec
        # included, that `--gpus 8` takes)
        return "relaunch", relaunch_command(argv, gpus)
    if int(ws) != gpus:
        return "fail", "started with WORLD_SIZE=%s but --gpus %d: refusing to report a number for another number of GPUs" % (ws, gpus)
    lr = int(environ.get("LOCAL_RANK", "0"))
    if lr >= n_devices:
        return "fail", "LOCAL_RANK=%d but %d HIP device(s) visible" % (lr, n_devices)
    return "run", None


def _main_sharded(args, torch):
    """--sharded-capi: the SECOND launch convention for N GPUs.  ONE process; ndtpso_shard_group_create(devices 0 .. N-1)
    (ncclCommInitAll inside the library, RCCL dlopen'ed), every device's shard of the weak-scaling workload (B pairs per
    device, the same pairs the per-process ranks would hold) resident on its device, and one step =
    ndtpso_align_pairs_sharded_dev: a host thread per device launches its shard, ONE ncclAllGather of pose + cost, every
    stream waited for.  One batch at a time per device by construction (the call returns with the devices idle), so its
    figure sits below the per-process path's two-in-flight `value`; no torch.distributed anywhere."""
    from ndtpso_slam_amd import capi, sharding, synth
    G, B, P, I = args.gpus, args.pairs, args.particles, args.iterations
    mode = {"f32": capi.SCORE_F32, "f64": capi.SCORE_F64, "exact": capi.SCORE_EXACT}[args.score]
    virt = int(os.environ.get("NDTPSO_SHARD_VIRTUAL") or 0)
    n_dev = torch.cuda.device_count()
    devices = list(range(min(n_dev, G))) if virt else list(range(G))
    grp = capi.ShardGroup(devices)
    info = grp.describe()
    assert info["n_shards"] == G, info
    geom = capi.ScanGeom(synth.N_BEAMS, float(synth.ANGLE_MIN), float(synth.ANGLE_INC), float(synth.RANGE_MAX), 0.1)
    grid = capi.Grid(FRAME_M, FRAME_M, CELL_SIDE)
    cfg = capi.PSOConfig.make(I, P)
    keep, ptrs = [], {k: [] for k in ("ref", "new", "guess", "dev", "seeds")}
    n_valid_total = 0
    for r in range(G):
        a, b = sharding.shard_range(G * B, r, G)
        pr = synth.make_pairs(b - a, seed=2024, first_pair=a, total_pairs=G * B)
        td = torch.device("cuda", info["devices"][r])
        t = {"ref": torch.from_numpy(pr.ref_ranges).to(td), "new": torch.from_numpy(pr.new_ranges).to(td),
             "guess": torch.zeros(b - a, 3, dtype=torch.float64, device=td),
             "dev": torch.tensor(DEVIATION, dtype=torch.float64, device=td).repeat(b - a, 1).contiguous(),
             "seeds": torch.from_numpy(pr.seeds.astype(np.int64)).to(td).to(torch.int32)}
        keep.append(t)
        for k in ptrs:
            ptrs[k].append(t[k].data_ptr())
        torch.cuda.synchronize(td)

    def step(fetch=False):
        return grp.align_pairs_dev(G * B, ptrs["ref"], ptrs["new"], geom, grid, ptrs["guess"], ptrs["dev"], cfg,
                                   d_seeds=ptrs["seeds"], mode=mode, fetch=fetch)

    settle = 0
    t_s = time.perf_counter()
    while args.settle_ms > 0 and (time.perf_counter() - t_s) * 1e3 < args.settle_ms:
        step()
        settle += 1
    for _ in range(args.warmup):
        step()
    g_us, per_call = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        g_us.append(grp.last_gather_device_us())
        per_call.append(grp.last_timing()[1].copy())
    elapsed = time.perf_counter() - t0     # (every call returns with all the devices synchronised)
    ranks_seen = grp.verify_gather()
    pose, cost, stats = step(fetch=True)
    per, _ = grp.last_timing()
    assert (stats["status"] == 0).all(), "alignments flagged: %d" % int((stats["status"] != 0).sum())
    assert ranks_seen == G, "the gather moved %d of %d shards' poses" % (ranks_seen, G)
    # the same poses as the one-device entry on shard 0's pairs (bit for bit: the partition changes nothing)
    c0 = capi.Context(info["devices"][0])
    p0 = synth.make_pairs(B, seed=2024, first_pair=0, total_pairs=G * B)
    want, _, _ = c0.align_pairs(p0.ref_ranges, p0.new_ranges, geom, grid, (0, 0, 0), DEVIATION, cfg, seeds=p0.seeds, mode=mode)
    c0.close()
    same = bool(np.array_equal(want, pose[:B]))
    n_valid = stats["n_points"].astype(np.float64)
    evals_nominal = 1 + P + P * I
    step_ms = 1e3 * elapsed / max(args.steps, 1)
    physical = len(set(info["devices"]))
    flops = float(n_valid.sum()) / G * evals_nominal * (FLOP_FP64 + FLOP_FP32 + FLOP_EXP) if args.score != "f64" else float(n_valid.sum()) / G * evals_nominal * 51.
    t_peak_s = (float(n_valid.sum()) / G * evals_nominal * (51. / (VEC_FP64_TFLOPS * 1e12)) if args.score == "f64" else
                float(n_valid.sum()) / G * evals_nominal * (FLOP_FP64 / (VEC_FP64_TFLOPS * 1e12) + (FLOP_FP32 + FLOP_EXP) / (VEC_FP32_TFLOPS * 1e12)))
    calls = np.array(per_call) if per_call else np.zeros((1, 3))
    grp.close()
    return {
        "metric": "scan alignments/sec (1081-beam, 70 particles x 70 iters)",
        "value": G * B * args.steps / elapsed, "unit": "alignments/s",
        "n_gpus": physical, "steps": args.steps, "warmup": args.warmup, "untimed_launches_total": settle + args.warmup,
        "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"exact": "f64 transform/index + f32 score, undecidable comparisons arbitrated in f64 (= the f64 mode's poses)",
                  "f32": "f64 transform/index + f32 score", "f64": "f64"}[args.score],
        "data": "synthetic",
        "config": {"workload": "BASELINE config 3/4: %d scan pairs per device per step over %d shard(s), %d beams, %.2f m cells, %d m frame, "
                               "PSO %d particles x %d iterations, exact reference order + srand(seed) stream"
                               % (B, G, synth.N_BEAMS, CELL_SIDE, FRAME_M, P, I),
                   "pairs_per_gpu": B, "particles": P, "iterations": I, "score": args.score, "batches_in_flight": 1,
                   "parallelism": "ONE process, ndtpso_align_pairs_sharded_dev: a host thread + context + stream per shard, pairs by "
                                  "contiguous index range, ONE gather of pose + cost per step (%s)" % info["gather"],
                   "shards": G, "shard_devices": info["devices"],
                   "test_mode": ("NDTPSO_SHARD_VIRTUAL=%d: %d shards share %d physical device(s) and the gather is staged through the "
                                 "host -- plumbing only, NOT a multi-GPU measurement" % (virt, G, physical)) if virt else None},
        "rccl": {"ranks_seen": ranks_seen, "world_size": G, "comm_ranks": info["comm_ranks"], "version": info["rccl_version"],
                 "gather": info["gather"], "gather_us_median": float(np.median(g_us)) if g_us else None,
                 "gather_us_what": "events on shard 0's stream: end of its own launch -> end of its share of the collective (the wait "
                                   "for the slowest shard included)",
                 "gather_bytes_per_rank": int(B * 32), "every_rank_found_its_poses_in_the_gather": ranks_seen == G,
                 "launch": "one process (--sharded-capi), ncclCommInitAll"},
        "roofline": {"bound": "valu", "achieved": flops / (step_ms * 1e-3) / 1e12, "peak": flops / t_peak_s / 1e12, "unit": "TFLOP/s",
                     "frac": t_peak_s * 1e3 / step_ms, "traffic": _pmc_traffic_bytes(args.score == "f64"),
                     "regime": "per device: flop floor of one shard's launch / ms_per_step of the sharded call (host dispatch, the launch, "
                               "the gather and the final synchronisation of every device included; one batch at a time)",
                     "kernel": "k_align_pairs (fused scan ingest + cell statistics + PSO)", "kernel_ms": None},
        "cpu_baseline": None,
        "extra": {"shard0_poses_equal_ndtpso_align_pairs": same, "status_nonzero": int((stats["status"] != 0).sum()),
                  "host_us_median": {"until_every_shard_was_enqueued": float(np.median(calls[:, 0])),
                                     "collective_enqueue": float(np.median(calls[:, 1])), "call": float(np.median(calls[:, 2]))},
                  "per_shard_host_us_last_call": {"start_after_entry": per[:, 0].tolist(), "uploads": per[:, 1].tolist(),
                                                  "launches": per[:, 2].tolist()},
                  "clock_settle": {"ms": args.settle_ms, "untimed_calls": settle}, "timed_region_s": elapsed},
    }


def _live_sequence(ctx, capi, synth, mode, n_scans=60):
    """Scans per second of the per-scan sequence of ndtpso_slam_node (ndtpso_slam_node.cpp:177-244) with the map, its
    sliding windows and the scans resident on the device (ndtpso_map_* / ndtpso_points_*)."""
    rng = np.random.default_rng(4)
    s = np.linspace(0.0, 0.6, n_scans)
    poses = np.stack([2.0 + 1.2 * s, -1.0 + 0.8 * np.sin(1.5 * s), 0.3 + 0.25 * s], axis=1)
    clean = synth.raycast(poses)
    ranges = np.where(clean > 0, clean + rng.normal(0, 0.01, clean.shape), 0.0).astype(np.float32)
    geom = capi.ScanGeom(synth.N_BEAMS, float(synth.ANGLE_MIN), float(synth.ANGLE_INC), float(synth.RANGE_MAX), 0.1)
    grid = capi.Grid(FRAME_M, FRAME_M, CELL_SIDE)
    cfg = capi.PSOConfig.make(50, 30)
    n_draw = 3 + 3 * 30 + 6 * 30 * 50
    tables = np.random.default_rng(5).integers(0, 2**31 - 1, size=(n_scans, n_draw), dtype=np.int64).astype(np.int32)
    rmap = capi.ResidentMap(ctx, grid, og_cell_size=0.1, pool_bytes=256 << 20)
    scan = capi.ResidentScan(ctx, 4096)
    prev = np.zeros(3)
    hist = [np.zeros(3), np.zeros(3)]
    t0 = None
    for k in range(n_scans):
        if k == 1:
            ctx.synchronize()
            t0 = time.perf_counter()
        scan.load_scan(ranges[k], geom, clip=grid)
        if k > 0:
            dev = np.array(DEVIATION) if k <= 2 else np.abs(2.0 * (hist[-1] - hist[-2]))   # ndtframe.cpp:253
            prev, _, _ = rmap.align(scan, prev, dev, cfg, rand_table=tables[k], mode=mode)
            hist.append(prev.copy())
        rmap.insert(scan, prev)
        rmap.speculate_build()   # what NDTFrame::update of the drop-in does for a frame that is aligned against
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / (n_scans - 1)
    info = rmap.info()
    rmap.close()
    scan.close()
    return {"scans_per_s": 1.0 / dt, "ms_per_scan": 1e3 * dt, "pso": "30 x 50", "scans": n_scans,
            "map_cells_built": int(info["n_built"]), "through": "ctypes binding (host/replay/node_replay.cpp is the C++ equivalent)"}


def _write_live_scans(synth, path, n_scans):
    rng = np.random.default_rng(4)
    s = np.linspace(0.0, 0.6, n_scans)
    poses = np.stack([2.0 + 1.2 * s, -1.0 + 0.8 * np.sin(1.5 * s), 0.3 + 0.25 * s], axis=1)
    clean = synth.raycast(poses)
    ranges = np.where(clean > 0, clean + rng.normal(0, 0.01, clean.shape), 0.0).astype(np.float32)
    with open(path, "wb") as f:
        np.array([n_scans, synth.N_BEAMS], dtype=np.int32).tofile(f)
        np.array([synth.ANGLE_MIN, synth.ANGLE_INC, synth.RANGE_MAX], dtype=np.float32).tofile(f)
        ranges.tofile(f)


def _live_replicas(synth, score, R, n_scans=120):
    """host/replay/node_replicas: R node sequences (30 x 50 PSO, resident map) on R host threads of one process."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "host", "replay", "node_replicas")
    if not os.path.exists(exe):
        return {"replicas": R, "error": "host/replay/node_replicas is not built (make -C host)"}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "scans.bin")
        _write_live_scans(synth, path, n_scans)
        r = subprocess.run([exe, path, str(FRAME_M), str(CELL_SIDE), "50", "30", "7", str(R)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, NDTPSO_RESIDENT="1", NDTPSO_SCORE=score))
    if r.returncode != 0 or not r.stdout.strip():
        return {"replicas": R, "error": "node_replicas failed: " + r.stderr[-300:]}
    d = json.loads(r.stdout.strip().splitlines()[-1])
    d["score"] = score
    return d


def _live_sequence_cpp(synth, score, n_scans=200, frame=FRAME_M):
    """The same sequence through the C++ drop-in library itself: host/replay/node_replay makes the node's calls
    (NDTFrame::loadLaser / align / update, re-allocation of the per-scan frame; ndtpso_slam_node.cpp:177-244) on
    libndtpso_slam.so, frames resident on the device, and reports the node's own "matching rate" (:239)."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "host", "replay", "node_replay")
    if not os.path.exists(exe):
        return {"error": "host/replay/node_replay is not built (make -C host)"}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "scans.bin")
        _write_live_scans(synth, path, n_scans)
        r = subprocess.run([exe, path, str(frame), str(CELL_SIDE), "50", "30", "7"], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, NDTPSO_RESIDENT="1", NDTPSO_SCORE=score))
    m = re.search(r"matching rate: ([0-9.]+) Hz \(([0-9.]+) ms per scan\)", r.stderr)
    if r.returncode != 0 or not m:
        return {"error": "node_replay failed: " + r.stderr[-300:]}
    return {"scans_per_s": float(m.group(1)), "ms_per_scan": float(m.group(2)), "pso": "30 x 50", "scans": n_scans, "score": score, "frame_m": frame,
            "through": "libndtpso_slam.so (C++ drop-in), host/replay/node_replay; per scan: loadLaser + align + update"}


def _load_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


# Vector-ALU peaks of the chip: fp32 157.3 TFLOP/s (MI355X_MICROARCH.md, "Peak FP32 (vector)"); fp64 vector = half of it
# (16 lanes per SIMD and clock against 32: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz = 78.6 TFLOP/s).
VEC_FP32_TFLOPS = 157.3
VEC_FP64_TFLOPS = 78.6
# flops of one point evaluation as this kernel performs it -- fp64: the transform (2 x 2 FMA = 8) and the
# two subtractions of the cell mean + two conversions to the table index + two to fp32 (8, counted one flop each);
# fp32: the Cholesky-form Gaussian and the fold of the terms (34 by the survey's count, 2 per FMA); one v_exp_f32.
FLOP_FP64, FLOP_FP32, FLOP_EXP = 16.0, 34.0, 1.0


def _roofline(stats, evals_nominal, kern_ms, algo_bytes, hbm_equiv_gbs, step_ms, depth, score="exact"):
    """The kernel keeps table, points and swarm in LDS: HBM sees 8.7 KB per alignment and no matrix instruction applies
    (DESIGN 5), so the roof is the VECTOR ALU of the chip.  Work of a launch = the reference's 1 + P + P*I cost
    evaluations per alignment x the points each scores (replays of the exact-order scheme not counted) x the flops of
    one point evaluation (16 fp64 + 34 fp32 + 1 exp).  achieved = that / the launch's duration (one launch
    at a time, HIP events on the launch stream).  peak = the same flop mix at the chip's vector peaks -- the time
    fp64_flops / 78.6 T + fp32_flops / 157.3 T, expressed as a rate -- so frac = (that time) / (kernel time).  The
    v_exp_f32 is charged as ONE fp32 flop although it issues at a quarter of the fp32 rate (frac_exp_at_quarter_rate
    charges it 4).  `floor` keeps round 2's instruction-level figure: the issue time of the score loop's actual
    instruction mix (scripts/isa_mix.py x scripts/ubench_valu.hip) -- useful for tuning, not a roofline."""
    # (--score f64: the same 51 flops per point evaluation, ALL of them fp64 -- the kernel executes exactly that many: 8
    # transform, 4 index, 2 differences, 9 quadratic form, 27 in the exponential's reduction and polynomial, 1 accumulate)
    f64 = score == "f64"
    FLOP_FP64, FLOP_FP32, FLOP_EXP = (51.0, 0.0, 0.0) if f64 else (16.0, 34.0, 1.0)
    cands = ("r06_isa_mix_f64.json", "r05_isa_mix_f64.json") if f64 else ("r06_isa_mix.json", "r05_isa_mix.json")
    mix_name = next((n for n in cands if _load_json(n)), cands[-1])
    mix = _load_json(mix_name)
    point_evals = float(stats["n_points"].astype(np.float64).sum()) * evals_nominal
    flops = point_evals * (FLOP_FP64 + FLOP_FP32 + FLOP_EXP)
    t_peak_s = point_evals * (FLOP_FP64 / (VEC_FP64_TFLOPS * 1e12) + (FLOP_FP32 + FLOP_EXP) / (VEC_FP32_TFLOPS * 1e12))
    t_peak_q_s = point_evals * (FLOP_FP64 / (VEC_FP64_TFLOPS * 1e12) + (FLOP_FP32 + 4.0 * FLOP_EXP) / (VEC_FP32_TFLOPS * 1e12))
    # ONE regime per figure: `achieved` / `frac` belong to the step time `value` is computed from (with two batches in
    # flight a step is shorter than a launch, because the next batch fills the compute units the slowest alignments of
    # this one leave idle); the per-launch figures (one launch at a time, what a kernel trace averages) carry their name.
    achieved = flops / (step_ms * 1e-3) / 1e12
    achieved_launch = flops / (kern_ms * 1e-3) / 1e12
    peak = flops / t_peak_s / 1e12
    r = {"bound": "valu", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
         "regime": "%d batch(es) in flight: flops of one step / ms_per_step (%.3f ms), the regime of `value`" % (depth, step_ms),
         "achieved_per_launch": achieved_launch, "frac_per_launch": achieved_launch / peak,
         "traffic": _pmc_traffic_bytes(f64), "kernel": "k_align_pairs (fused scan ingest + cell statistics + PSO)",
         "kernel_ms": kern_ms, "algorithmic_point_evals_per_launch": point_evals,
         "flops_per_point_eval": {"fp64": FLOP_FP64, "fp32": FLOP_FP32, "exp": FLOP_EXP},
         "vector_peaks_tflops": {"fp32": VEC_FP32_TFLOPS, "fp64": VEC_FP64_TFLOPS},
         "flop_floor_ms_per_launch": t_peak_s * 1e3,
         "frac_exp_at_quarter_rate": t_peak_q_s * 1e3 / step_ms,
         "note": "frac = flop floor / ms_per_step; frac_per_launch = flop floor / kernel_ms (one launch at a time on one stream, "
                 "HIP events; measured after the timed region when batches overlap, because then a launch's own duration "
                 "includes its wait for the other batch's compute units)",
         "frac_of_step_time": t_peak_s * 1e3 / step_ms}
    if mix:
        ns_chunk = float(mix["valu_issue_ns_per_chunk"])
        issue_peak = N_SIMD * 64.0 / (ns_chunk * 1e-9)
        r["floor"] = {"valu_instructions_per_64_points": mix["valu_per_chunk"], "valu_issue_ns_per_64_points": ns_chunk,
                      "floor_ms_per_launch": point_evals / issue_peak * 1e3,
                      "frac_of_kernel_ms": point_evals / issue_peak * 1e3 / kern_ms,
                      "what": "issue time of the score loop's own instruction mix on 1024 SIMDs (a shorter loop lowers it): a tuning "
                              "figure, not the roofline",
                      "source": "profiles/%s (scripts/isa_mix.py: llvm-objdump of the shipped kernel x scripts/ubench_valu.hip on this chip)" % mix_name}
    r["hbm_effective"] = {"achieved": hbm_equiv_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_equiv_gbs / HBM_PEAK_GBS,
                          "algorithmic_bytes_per_launch": algo_bytes,
                          "note": "SURVEY 8(d)'s streaming-equivalent accounting: 40 B per point evaluation (16 B point + 24 B cell "
                                  "record); the kernel serves them from LDS, so this exceeds the HBM peak by construction -- HBM is "
                                  "not the roof (traffic = measured HBM bytes per launch of the profiled mode: compulsory 8.7 KB per "
                                  "alignment, x 1.3 in the plain fp32 kernel; the exact kernel adds its fp64 table image per workgroup "
                                  "and the scratch traffic of its arbitration calls)"}
    return r


def _pmc_traffic_bytes(f64=False):
    """HBM bytes per launch of the fused kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x 2 per the
    gfx950 correction in MI355X_MICROARCH.md, + WRITE_SIZE; separate --pmc runs, scripts/pmc.sh).  A profile of
    THIS workload measured on MI355X, not collected live (rocprofv3 cannot wrap the timed run); null if absent."""
    names = ("r06_pmc_summary_f64.json", "r05_pmc_summary_f64.json") if f64 else ("r06_pmc_summary.json", "r05_pmc_summary.json")
    for name in names:
        d = _load_json(name)
        try:
            d = d["derived"]
            return float(d["hbm_read_bytes_per_launch_FETCH_SIZE_x2_KiB_units"]) + float(d["hbm_write_bytes_per_launch"])
        except (TypeError, KeyError, ValueError):
            continue
    return None


def _host_cpu_limits():
    """What this process may actually use of the box's CPUs: logical CPUs, the affinity mask, the cgroup's quota, OpenMP's
    environment, the load other tenants put on the box."""
    d = {"logical_cpus": os.cpu_count() or 1, "cpu_model": _cpu_model()}
    try:
        d["affinity_cpus"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        d["affinity_cpus"] = None
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                d["cgroup_cpu_max"] = " ".join(txt)
                if txt and txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = float(f.read().split()[0])
                d["cgroup_cfs_quota_us"] = q
                if q > 0:
                    quota = q / per
            break
        except (OSError, ValueError, IndexError):
            continue
    d["cgroup_quota_cpus"] = quota
    d["OMP_NUM_THREADS"] = os.environ.get("OMP_NUM_THREADS")
    try:
        with open("/proc/loadavg") as f:
            d["loadavg_1min"] = float(f.read().split()[0])
    except (OSError, ValueError):
        d["loadavg_1min"] = None
    try:   # physical cores: distinct (package, core) pairs
        cores, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
        d["physical_cores"] = len(cores) or None
    except OSError:
        d["physical_cores"] = None
    return d


def _cpu_baseline(pairs, S, P, I):
    """the repo's fp64 CPU restatement of the reference (oracle/ndtpso_oracle.c, `kind: "port"` -- the
    reference itself cannot be built in this image) on the GPU box's host cores:
      c1       one scan pair (BASELINE config 1) in the reference's own parallel shape -- OpenMP over the particles of an
               iteration, live rand(), racy gbest (core.cpp:72-109) -- median of 11 alignments, at 1, 8 and all threads
               (the reference's default is all: num_threads = -1, config.h:30);
      scaling  sequential alignments parallelised over PAIRS (no synchronisation at all: the fairest CPU figure) at
               1, 2, 4, ... threads up to what the process may use, on 4 pairs per thread (16 ... S);
      value    the best rate over all S pairs of this step; `cores` = the FEWEST threads that reach 90 % of the best rate
               of the curve -- the cores the box really gives this process, whatever the number of logical CPUs it shows --,
               and `efficiency` = value / (cores x the one-thread rate)."""
    from oracle import pyoracle
    ocfg = pyoracle.PSOConfig.make(I, P)
    host = _host_cpu_limits()
    nproc = host["logical_cpus"]
    usable = host["affinity_cpus"] or nproc
    ref = pyoracle.Frame((0, 0, 0), FRAME_M, FRAME_M, CELL_SIDE)
    ref.load_laser(pairs.ref_ranges[0], pairs.angle_min, pairs.angle_inc, pairs.range_max)
    new = pyoracle.Frame((0, 0, 0), FRAME_M, FRAME_M, float(FRAME_M))
    new.load_laser(pairs.new_ranges[0], pairs.angle_min, pairs.angle_inc, pairs.range_max)
    ref.build()
    c1 = {}
    # ("all": the CPUs the process may really use -- affinity mask cut by the control group's quota --, not the 256 the box shows:
    # 256 OpenMP threads on a quota of 16 CPUs measured 3.3 alignments/s, a figure about the quota, not about the code)
    eff = int(max(1, min(usable, int(host["cgroup_quota_cpus"]) if host["cgroup_quota_cpus"] else usable)))
    for label, nt in (("1_thread", 1), ("8_threads", 8), ("all_usable_threads", eff)):
        ts = []
        for _ in range(11):
            t1 = time.perf_counter()
            ref.pso_omp((0, 0, 0), new, DEVIATION, ocfg, n_threads=nt)
            ts.append(time.perf_counter() - t1)
        c1[label] = {"alignments_per_s": 1.0 / float(np.median(ts)), "median_ms": 1e3 * float(np.median(ts)),
                     "threads": nt, "runs": len(ts)}

    def run(n_pairs, threads):
        t1 = time.perf_counter()
        po, _, used = pyoracle.align_pairs(pairs.ref_ranges[:n_pairs], pairs.new_ranges[:n_pairs], pairs.angle_min, pairs.angle_inc,
                                           pairs.range_max, 0.1, FRAME_M, FRAME_M, CELL_SIDE, (0, 0, 0), DEVIATION, ocfg,
                                           pairs.seeds[:n_pairs], n_threads=threads)
        return n_pairs / (time.perf_counter() - t1), po, int(used)

    curve, t = [], 1
    while True:
        n = int(min(S, max(16, 4 * t)))
        rate, _, used = run(n, t)
        curve.append([used, rate, n])
        if t >= usable:
            break
        t = min(2 * t, usable)
    one = curve[0][1]
    best_curve = max(r for _, r, _ in curve)
    cores = min(u for u, r, _ in curve if r >= 0.9 * best_curve)
    t_best = max(curve, key=lambda c: c[1])[0]
    best, opose = None, None
    for _ in range(2):
        rate, opose, used = run(S, t_best)
        best = rate if best is None else max(best, rate)
    eff = best / (cores * one)
    why = []
    if host["cgroup_quota_cpus"]:
        why.append("cgroup CPU quota of %.1f CPUs" % host["cgroup_quota_cpus"])
    if host["affinity_cpus"] and host["affinity_cpus"] < nproc:
        why.append("affinity mask of %d of %d logical CPUs" % (host["affinity_cpus"], nproc))
    if host["loadavg_1min"] is not None and host["loadavg_1min"] > 0.25 * nproc:
        why.append("1-minute load average %.0f on %d logical CPUs (other tenants)" % (host["loadavg_1min"], nproc))
    if not why:
        why.append("no quota, mask or foreign load visible from inside the container: the curve itself is the evidence (SMT siblings, "
                   "all-core clocks and memory bandwidth shared with the box's other tenants)")
    return {
        "value": best, "unit": "alignments/s", "cores": int(cores), "kind": "port",
        "sample": "all %d scan pairs of this step (BASELINE config 3), sequential alignments of oracle/ndtpso_oracle.c parallelised over "
                  "pairs on %d OpenMP threads (the best of the scaling curve), best of 2 passes; `cores` = %d: the fewest threads that "
                  "reach 90 %% of the curve's best rate (%s); %s, %d logical CPUs" % (S, t_best, cores, "; ".join(why), host["cpu_model"], nproc),
        "seconds": S / best, "threads_of_value": int(t_best),
        "one_thread_alignments_per_s": one,
        "efficiency": eff, "efficiency_what": "value / (cores x one-thread rate)",
        "scaling": [[int(u), float(r)] for u, r, _ in curve], "scaling_pairs": [int(n) for _, _, n in curve],
        "host": host,
        "c1_single_pair_omp_over_particles": c1,
        "c3_pairs": S, "nproc": nproc, "cpu_model": host["cpu_model"],
        "_poses": opose,
    }


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


if __name__ == "__main__":
    main()
