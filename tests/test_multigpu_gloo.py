"""world_size-2 gloo test of the multi-GPU layout: contiguous sharding of a recorded run + the single
all_gather of poses.  The compute on each rank is the CPU oracle (the HIP path needs a GPU); what is under
test is the host-side plumbing bench.py uses on N GPUs: shard ranges, per-pair determinism of the synthetic
run across ranks, gather order."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["NDTPSO_ROOT"])
from ndtpso_slam_amd import synth, sharding
from oracle import pyoracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
TOTAL = 7                                     # uneven on purpose
first, last = sharding.shard_range(TOTAL, rank, world)
p = synth.make_pairs(last - first, n_beams=181, seed=13, first_pair=first, total_pairs=TOTAL)
cfg = O.PSOConfig.make(6, 8)
pose, cost, _ = O.align_pairs(p.ref_ranges, p.new_ranges, p.angle_min, p.angle_inc, p.range_max, 0.1, 60, 60, 0.5,
                              (0, 0, 0), (0.1, 0.1, 3.1415e-3), cfg, p.seeds, n_threads=1)
allp = sharding.gather_poses(torch.from_numpy(pose))
assert allp.shape == (TOTAL, 3)
if rank == 0:
    np.save(os.environ["NDTPSO_OUT"], allp.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_range_partitions():
    from ndtpso_slam_amd.sharding import shard_range
    for n, w in [(4096, 8), (512, 1), (7, 2), (5, 8), (0, 3)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert shard_range(4096, 3, 8) == (1536, 2048)      # BASELINE config 4: 512 contiguous pairs per GPU


def test_two_rank_gather_equals_single_process(tmp_path):
    out = str(tmp_path / "gathered.npy")
    env = dict(os.environ, NDTPSO_ROOT=ROOT, NDTPSO_OUT=out, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               OMP_NUM_THREADS="1")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE="2"))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)

    sys.path.insert(0, ROOT)
    from ndtpso_slam_amd import synth
    from oracle import pyoracle as O
    p = synth.make_pairs(7, n_beams=181, seed=13, first_pair=0, total_pairs=7)
    want, _, _ = O.align_pairs(p.ref_ranges, p.new_ranges, p.angle_min, p.angle_inc, p.range_max, 0.1, 60, 60, 0.5,
                               (0, 0, 0), (0.1, 0.1, 3.1415e-3), O.PSOConfig.make(6, 8), p.seeds, n_threads=1)
    assert np.array_equal(got, want)


# ---- the same through the HIP path: two ranks on the one GPU of the box ------------------------------------------

GPU_WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["NDTPSO_ROOT"])
from ndtpso_slam_amd import capi, synth, sharding
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ["NDTPSO_BACKEND"]
dev = torch.device("cuda", 0)                 # both ranks share the box's single GPU
torch.cuda.set_device(dev)
dist.init_process_group(backend, rank=rank, world_size=world)
TOTAL, P, I = 128, 70, 70                     # 64 pairs per rank, BASELINE's PSO
first, last = sharding.shard_range(TOTAL, rank, world)
p = synth.make_pairs(last - first, seed=5, first_pair=first, total_pairs=TOTAL)
B = last - first
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
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
ctx.align_pairs_dev(B, d_ref.data_ptr(), d_new.data_ptr(), geom, capi.Grid(60, 60, 0.5), d_guess.data_ptr(), d_dev.data_ptr(),
                    capi.PSOConfig.make(I, P), d_seeds.data_ptr(), 0, capi.SCORE_EXACT, d_pose.data_ptr(), d_cost.data_ptr(),
                    d_stats.data_ptr())
local = d_pose if backend == "nccl" else d_pose.cpu()      # gloo gathers host tensors, RCCL device tensors
allp = sharding.gather_poses(local, equal_sizes=True)
assert allp.shape == (TOTAL, 3)
st = d_stats.cpu().numpy().view(capi.STATS_DTYPE).reshape(B)
assert (st["status"] == 0).all()
if rank == 0:
    np.save(os.environ["NDTPSO_OUT"], allp.cpu().numpy())
dist.barrier()
dist.destroy_process_group()
ctx.close()
'''


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_ranks_through_the_hip_path(tmp_path, ctx, backend):
    """SURVEY 8(e) as far as one GPU goes: two processes, each running `ndtpso_align_pairs_dev` (exact mode) on its
    contiguous shard of 128 pairs and the single all_gather of the poses -- over gloo (host tensors) and over RCCL
    (device tensors, both ranks on cuda:0) -- must reproduce the one-process, one-launch result bit for bit."""
    import torch
    if backend == "nccl" and torch.cuda.device_count() < 1:
        pytest.skip("no GPU")
    out = str(tmp_path / "gathered.npy")
    env = dict(os.environ, NDTPSO_ROOT=ROOT, NDTPSO_OUT=out, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541",
               NDTPSO_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = tmp_path / "gpu_worker.py"
    script.write_text(GPU_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE="2"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    if backend == "nccl" and any(p.returncode != 0 for p in procs) and any("same device" in o.lower() or "duplicate gpu" in o.lower() for o in outs):
        pytest.skip("RCCL refuses two ranks on one device on this box: " + outs[0][-300:])
    assert all(p.returncode == 0 for p in procs), outs
    got = np.load(out)
    from ndtpso_slam_amd import capi, synth
    p = synth.make_pairs(128, seed=5)
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    want, _, st = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, 0.5), (0, 0, 0), (0.1, 0.1, 3.1415e-3),
                                  capi.PSOConfig.make(70, 70), seeds=p.seeds, mode=capi.SCORE_EXACT)
    assert (st["status"] == 0).all()
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_bench_multi_rank_path_under_torch_distributed_run(tmp_path):
    """bench.py exactly as the driver launches it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N),
    with the one GPU of this box: N = 1 through the launcher takes the RCCL path (process group on device 0, the
    all_gather of poses every step, barriers, MAX over ranks) and must print the one JSON line."""
    import json
    env = dict(os.environ, NDTPSO_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "10", "--warmup", "3", "--cpu-sample", "0", "--no-latency"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["value"] > 1e4 and d["scaling"] == "weak"
    assert d["roofline"]["bound"] == "valu" and 0 < d["roofline"]["frac"] < 1
    assert d["rccl"]["ranks_seen"] == 1 and d["rccl"]["backend"] == "nccl" and d["rccl"]["every_rank_found_its_poses_in_the_gather"]
    assert d["rccl"]["gather_us_median"] > 0 and "launcher" in d["rccl"]["launch"]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus N` WITHOUT torch.distributed.run around it (how the driver starts the N = 1 line): for N > 1 it
    re-executes itself under the launcher.  On the one-GPU box NDTPSO_BENCH_FORCE_DIST=1 sends --gpus 1 down the same road --
    the exec, the rendezvous on 127.0.0.1, a process group of one rank over RCCL, the all_gather every step."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NDTPSO_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3", "--cpu-sample", "0",
                        "--no-latency"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "without a launcher: starting" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl"]["ranks_seen"] == 1 and "bench.py itself" in d["rccl"]["launch"]


# ---- one process, several devices: the C-ABI's sharded entry point (ndtpso_align_pairs_sharded) --------------------

def test_c_abi_shard_range_is_the_python_partition():
    from ndtpso_slam_amd import capi, sharding
    for n in (0, 1, 5, 7, 512, 4096, 4099):
        for G in (1, 2, 3, 8):
            for r in range(G):
                assert capi.shard_range(n, r, G) == sharding.shard_range(n, r, G)


def test_shard_group_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ndtpso_slam_amd import capi
    with pytest.raises(capi.NdtpsoError):
        capi.ShardGroup([0])
    with pytest.raises(capi.NdtpsoError):
        capi.ShardGroup([0, 0])


def write_batch_file(path, p, grid_m, cell_side, I, P, mode, guess=(0.0, 0.0, 0.0), deviation=(0.1, 0.1, 3.1415e-3)):
    """The batch format host/replay/batch_sharded.cpp reads."""
    import struct
    B = p.n_pairs
    with open(path, "wb") as f:
        f.write(b"NDTB")
        f.write(struct.pack("<III", 1, B, p.n_beams))
        f.write(struct.pack("<ffff", float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1))
        f.write(struct.pack("<IId", grid_m, grid_m, cell_side))
        f.write(struct.pack("<ii", I, P))
        f.write(struct.pack("<dddd", 0.8, 2.0, 2.0, 1.0))
        f.write(struct.pack("<i", mode))
        f.write(np.ascontiguousarray(p.ref_ranges, dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(p.new_ranges, dtype=np.float32).tobytes())
        f.write(np.tile(np.asarray(guess, dtype=np.float64), (B, 1)).tobytes())
        f.write(np.tile(np.asarray(deviation, dtype=np.float64), (B, 1)).tobytes())
        f.write(np.ascontiguousarray(p.seeds, dtype=np.uint32).tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("n_pairs", [3, 130, 301])
def test_sharded_entry_point_equals_align_pairs(ctx, n_pairs):
    """ndtpso_align_pairs_sharded over every device of the box (one on the test box: the n = 1 case -- scatter, launch,
    a one-rank ncclAllGather through RCCL, copy back) returns ndtpso_align_pairs' poses, costs and statistics bit for
    bit; with more devices the same assertion covers the partition and the gather order."""
    import torch
    from ndtpso_slam_amd import capi, synth
    p = synth.make_pairs(n_pairs, seed=31)
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    grid, cfg = capi.Grid(60, 60, 0.5), capi.PSOConfig.make(12, 20)
    dev = (0.1, 0.1, 3.1415e-3)
    want, wcost, wst = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds, mode=capi.SCORE_EXACT)
    for devices in ([0], list(range(torch.cuda.device_count()))):
        g = capi.ShardGroup(devices)
        assert g.size() == len(devices)
        for _ in range(2):   # a group is reusable
            got, cost, st = g.align_pairs(p.ref_ranges, p.new_ranges, geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds, mode=capi.SCORE_EXACT)
            assert np.array_equal(got, want) and np.array_equal(cost, wcost)
            assert np.array_equal(st["status"], wst["status"]) and np.array_equal(st["gbest_updates"], wst["gbest_updates"])
        g.close()


@pytest.mark.gpu
def test_sharded_resident_flavour_and_timing(ctx):
    """ndtpso_align_pairs_sharded_dev: every device's shard already resident (per-device arrays of device pointers).  Same
    poses as the host flavour; the gathered batch can stay on the devices; the call reports where its host time went."""
    import torch
    from ndtpso_slam_amd import capi, synth
    B = 301
    p = synth.make_pairs(B, seed=33)
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    grid, cfg = capi.Grid(60, 60, 0.5), capi.PSOConfig.make(12, 20)
    dev = (0.1, 0.1, 3.1415e-3)
    want, wcost, wst = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds, mode=capi.SCORE_EXACT)
    devices = list(range(torch.cuda.device_count()))
    G = len(devices)
    g = capi.ShardGroup(devices)
    keep, ptrs = [], {k: [] for k in ("ref", "new", "guess", "dev", "seeds")}
    for r, d in enumerate(devices):
        a, b = capi.shard_range(B, r, G)
        td = torch.device("cuda", d)
        t = {"ref": torch.from_numpy(p.ref_ranges[a:b]).to(td), "new": torch.from_numpy(p.new_ranges[a:b]).to(td),
             "guess": torch.zeros(b - a, 3, dtype=torch.float64, device=td),
             "dev": torch.tensor(dev, dtype=torch.float64, device=td).repeat(b - a, 1).contiguous(),
             "seeds": torch.from_numpy(p.seeds[a:b].astype(np.int64)).to(td).to(torch.int32)}
        keep.append(t)
        for k in ptrs:
            ptrs[k].append(t[k].data_ptr())
        torch.cuda.synchronize(td)
    got, cost, st = g.align_pairs_dev(B, ptrs["ref"], ptrs["new"], geom, grid, ptrs["guess"], ptrs["dev"], cfg, d_seeds=ptrs["seeds"])
    assert np.array_equal(got, want) and np.array_equal(cost, wcost) and np.array_equal(st["gbest_updates"], wst["gbest_updates"])
    per, call = g.last_timing()
    assert per.shape == (G, 3) and (per[:, 1] == 0).all() and (per[:, 2] > 0).all()      # nothing uploaded, something launched
    assert 0 < call[0] <= call[2] and call[1] > 0
    # results left on the devices: device 0's copy of the gathered batch, block r = shard r as [pose (M x 3) | cost (M)]
    _, _, st2 = g.align_pairs_dev(B, ptrs["ref"], ptrs["new"], geom, grid, ptrs["guess"], ptrs["dev"], cfg, d_seeds=ptrs["seeds"], fetch=False)
    M = -(-B // G)
    import ctypes
    raw = np.empty(4 * M * G)
    assert g.gathered(0) != 0
    hip = ctypes.CDLL("libamdhip64.so")      # the HIP runtime torch has already loaded: one plain device-to-host copy
    assert hip.hipMemcpy(ctypes.c_void_p(raw.ctypes.data), ctypes.c_void_p(g.gathered(0)), ctypes.c_size_t(raw.nbytes), 2) == 0
    for r in range(G):
        a, b = capi.shard_range(B, r, G)
        assert np.array_equal(raw[4 * M * r:4 * M * r + 3 * (b - a)].reshape(b - a, 3), want[a:b])
        assert np.array_equal(raw[4 * M * r + 3 * M:4 * M * r + 3 * M + (b - a)], wcost[a:b])
    # the host flavour reports its uploads
    g.align_pairs(p.ref_ranges, p.new_ranges, geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds, mode=capi.SCORE_EXACT)
    per, call = g.last_timing()
    assert (per[:, 1] > 0).all()
    g.close()


@pytest.mark.gpu
def test_cpp_batch_driver_on_all_devices(tmp_path, ctx):
    """host/replay/batch_sharded (C++, the C-ABI only) on a batch file: same poses as the Python binding."""
    from ndtpso_slam_amd import capi, synth
    exe = os.path.join(ROOT, "host", "replay", "batch_sharded")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s", "replay/batch_sharded"])
    p = synth.make_pairs(140, seed=52)
    batch, out = str(tmp_path / "batch.bin"), str(tmp_path / "poses.bin")
    write_batch_file(batch, p, 60, 0.5, 10, 16, capi.SCORE_EXACT)
    r = subprocess.run([exe, batch, out, "all", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(out, dtype=np.float64)
    got, gcost = raw[:3 * 140].reshape(140, 3), raw[3 * 140:]
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    want, wcost, _ = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, capi.Grid(60, 60, 0.5), (0, 0, 0), (0.1, 0.1, 3.1415e-3),
                                     capi.PSOConfig.make(10, 16), seeds=p.seeds, mode=capi.SCORE_EXACT)
    assert np.array_equal(got, want) and np.array_equal(gcost, wcost)
    import json
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["pairs"] == 140 and info["devices"] >= 1 and info["flagged"] == 0
    assert info["ranks_seen"] == info["shards"] == info["devices"] and info["gather"].startswith("ncclAllGather")


def test_cpp_batch_driver_refuses_to_run_without_a_device(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ndtpso_slam_amd import capi, synth
    exe = os.path.join(ROOT, "host", "replay", "batch_sharded")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s", "replay/batch_sharded"])
    p = synth.make_pairs(2, n_beams=91, seed=1)
    batch = str(tmp_path / "batch.bin")
    write_batch_file(batch, p, 60, 0.5, 3, 4, capi.SCORE_EXACT)
    r = subprocess.run([exe, batch, str(tmp_path / "o.bin"), "all"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "nothing is computed on the CPU" in r.stderr
    assert not os.path.exists(str(tmp_path / "o.bin"))


# ---- launch conventions of bench.py: a run that asks for N GPUs never reports a number measured on fewer ------------------

def test_bench_launch_plan_rules():
    """bench.launch_plan: --gpus N without a launcher re-executes under torch.distributed.run (the contract's command line);
    under a launcher WORLD_SIZE must equal N; too few devices, a stray launcher around --sharded-capi or a virtual shard count
    that differs from N are refused with a reason."""
    sys.path.insert(0, ROOT)
    import bench
    kind, cmd = bench.launch_plan(8, False, {}, 8, ["--gpus", "8", "--steps", "5"])
    assert kind == "relaunch"
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5"]
    assert bench.launch_plan(1, False, {}, 1, []) == ("run", None)
    assert bench.launch_plan(8, False, {"WORLD_SIZE": "8", "LOCAL_RANK": "7"}, 8, []) == ("run", None)
    for args in [(8, False, {}, 1, []),                                       # eight asked for, one visible
                 (8, False, {"WORLD_SIZE": "1"}, 8, []),                      # `python bench.py --gpus 8` inside a one-rank launcher
                 (1, False, {"WORLD_SIZE": "8", "LOCAL_RANK": "0"}, 8, []),   # eight ranks, --gpus forgotten
                 (8, False, {"WORLD_SIZE": "8", "LOCAL_RANK": "7"}, 4, []),   # a rank without a device
                 (0, False, {}, 1, []),
                 (8, True, {}, 4, []),
                 (8, True, {"WORLD_SIZE": "8"}, 8, []),
                 (8, True, {"NDTPSO_SHARD_VIRTUAL": "4"}, 1, [])]:
        kind, why = bench.launch_plan(*args)
        assert kind == "fail" and isinstance(why, str) and why, args
    assert bench.launch_plan(8, True, {}, 8, []) == ("sharded", None)
    assert bench.launch_plan(8, True, {"NDTPSO_SHARD_VIRTUAL": "8"}, 1, []) == ("sharded", None)


def test_bench_refuses_more_gpus_than_are_visible():
    """`python bench.py --gpus N` on a box with fewer than N devices (here: whatever the box has + 1): exit code 2, a
    one-line reason, no JSON line."""
    import torch
    n = (torch.cuda.device_count() if torch.cuda.is_available() else 0) + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for extra in ([], ["--sharded-capi"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"] + extra,
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 2, (r.returncode, r.stderr[-500:])
        assert "HIP device(s) visible" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1 but --gpus 2" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_are_visible_on_a_gpu_box():
    test_bench_refuses_more_gpus_than_are_visible()


# ---- the G-device call on a one-device box: NDTPSO_SHARD_VIRTUAL (the shard group's test mode) -----------------------------

VIRTUAL_WORKER = r'''
import json, os, sys, time
import numpy as np
import torch   # before the library: the process must load torch's own HIP runtime first (torch finds no device in another one)
sys.path.insert(0, os.environ["NDTPSO_ROOT"])
from ndtpso_slam_amd import capi, synth
G = int(os.environ["NDTPSO_SHARD_VIRTUAL"])
B = int(os.environ["NDTPSO_TEST_PAIRS"])
p = synth.make_pairs(B, n_beams=541, seed=77)
geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
grid, cfg = capi.Grid(60, 60, 0.5), capi.PSOConfig.make(8, 16)
dev = (0.1, 0.1, 3.1415e-3)
ctx = capi.Context(0)
want, wcost, wst = ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds, mode=capi.SCORE_EXACT)
g = capi.ShardGroup([0])
info = g.describe()
assert info["n_shards"] == G and info["gather"] == "host-staged (test)" and info["devices"] == [0] * G and info["comm_ranks"] == 0, info
res = {"info": info}
got, cost, st = g.align_pairs(p.ref_ranges, p.new_ranges, geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds, mode=capi.SCORE_EXACT)
assert np.array_equal(got, want) and np.array_equal(cost, wcost) and np.array_equal(st["gbest_updates"], wst["gbest_updates"])
assert g.verify_gather() == G
per, call = g.last_timing()
sizes = [capi.shard_range(B, r, G) for r in range(G)]
assert (per[:, 1] > 0).sum() == sum(1 for a, b in sizes if b > a)      # every non-empty shard uploaded its own block
# one shard failing fails the call -- with its reason, promptly, nothing hanging -- and the group stays usable
os.environ["NDTPSO_SHARD_TEST_FAIL"] = str(G - 2 if G > 2 else 0)
t0 = time.time()
try:
    g.align_pairs(p.ref_ranges, p.new_ranges, geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds, mode=capi.SCORE_EXACT)
    raise SystemExit("the injected failure was not reported")
except capi.NdtpsoError as e:
    assert "NDTPSO_SHARD_TEST_FAIL" in str(e), str(e)
res["failed_call_s"] = time.time() - t0
try:
    g.verify_gather()
    raise SystemExit("verify_gather after a failed call")
except capi.NdtpsoError:
    pass
del os.environ["NDTPSO_SHARD_TEST_FAIL"]
got, cost, st = g.align_pairs(p.ref_ranges, p.new_ranges, geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds, mode=capi.SCORE_EXACT)
assert np.array_equal(got, want) and np.array_equal(cost, wcost) and g.verify_gather() == G
# a null pointer for a non-empty shard of the resident flavour: an argument error, not a hang
keep, ptrs = [], {k: [] for k in ("ref", "new", "guess", "dev", "seeds")}
td = torch.device("cuda", 0)
for r in range(G):
    a, b = sizes[r]
    t = {"ref": torch.from_numpy(p.ref_ranges[a:b]).to(td), "new": torch.from_numpy(p.new_ranges[a:b]).to(td),
         "guess": torch.zeros(b - a, 3, dtype=torch.float64, device=td),
         "dev": torch.tensor(dev, dtype=torch.float64, device=td).repeat(b - a, 1).contiguous(),
         "seeds": torch.from_numpy(p.seeds[a:b].astype(np.int64)).to(td).to(torch.int32)}
    keep.append(t)
    for k in ptrs:
        ptrs[k].append(t[k].data_ptr() if b > a else 0)
torch.cuda.synchronize()
got, cost, st = g.align_pairs_dev(B, ptrs["ref"], ptrs["new"], geom, grid, ptrs["guess"], ptrs["dev"], cfg, d_seeds=ptrs["seeds"])
assert np.array_equal(got, want) and np.array_equal(cost, wcost)
bad = list(ptrs["new"])
bad[0] = 0
try:
    g.align_pairs_dev(B, ptrs["ref"], bad, geom, grid, ptrs["guess"], ptrs["dev"], cfg, d_seeds=ptrs["seeds"])
    raise SystemExit("a null shard pointer was accepted")
except capi.NdtpsoError as e:
    assert "null device pointer" in str(e), str(e)
g.close()
ctx.close()
print(json.dumps(res))
'''


@pytest.mark.gpu
@pytest.mark.parametrize("G,n_pairs", [(8, 4099), (8, 5), (3, 301)])
def test_virtual_shards_on_one_device(tmp_path, G, n_pairs):
    """What the first G-device call will execute, on the box's one device: G shards (context, stream and host thread each)
    over an UNEVEN partition (4096 + 3 pairs over 8; 5 pairs over 8: three shards empty), every shard's block scattered by
    its own thread, the gathered batch found on every shard, poses bit for bit those of ndtpso_align_pairs; an injected
    failure of one shard fails the call with its reason within seconds and leaves the group usable."""
    import json
    script = tmp_path / "virt.py"
    script.write_text(VIRTUAL_WORKER)
    env = dict(os.environ, NDTPSO_ROOT=ROOT, NDTPSO_SHARD_VIRTUAL=str(G), NDTPSO_TEST_PAIRS=str(n_pairs))
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["failed_call_s"] < 10.0


@pytest.mark.gpu
@pytest.mark.parametrize("virtual", [0, 8])
def test_bench_sharded_capi_mode(virtual):
    """bench.py --sharded-capi: the one-process launch convention (ndtpso_align_pairs_sharded_dev), with the JSON contract
    of the per-process runs.  virtual = 0: the box's one device through real RCCL (ncclCommInitAll of one rank);
    virtual = 8: --gpus 8 as eight shards on the one device, flagged as a test mode and reported as n_gpus 1."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    n = 8 if virtual else 1
    if virtual:
        env["NDTPSO_SHARD_VIRTUAL"] = "8"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sharded-capi", "--gpus", str(n), "--steps", "6", "--warmup", "2",
                        "--settle-ms", "50", "--pairs", "128" if virtual else "512"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 1e4
    assert d["rccl"]["ranks_seen"] == n and d["rccl"]["every_rank_found_its_poses_in_the_gather"]
    assert d["extra"]["shard0_poses_equal_ndtpso_align_pairs"] and d["extra"]["status_nonzero"] == 0
    assert d["config"]["shards"] == n and 0 < d["roofline"]["frac"] < 1
    if virtual:
        assert "NOT a multi-GPU measurement" in d["config"]["test_mode"] and d["rccl"]["gather"] == "host-staged (test)"
    else:
        assert d["config"]["test_mode"] is None and d["rccl"]["gather"].startswith("ncclAllGather") and d["rccl"]["comm_ranks"] == 1
        assert d["rccl"]["version"]


@pytest.mark.gpu
def test_sharded_host_entry_resolves_pairs_the_kernels_leave_flagged(ctx):
    """0.125 m cells in the 60 m frame: a few rooms are beyond the fused kernels (largest table too small, bitmap form beyond LDS)
    and come back flagged from them.  The entries that hold the inputs on the host -- ndtpso_align_pairs and
    ndtpso_align_pairs_sharded -- align those through a resident frame; both return the same poses, nothing flagged."""
    from ndtpso_slam_amd import capi, synth
    p = synth.make_pairs(600, seed=606)
    sel = np.r_[28:44]          # (34, 38, 39 are three of the thirteen pairs of this set the kernels cannot hold)
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    grid, cfg = capi.Grid(60, 60, 0.125), capi.PSOConfig.make(10, 16)
    dev = (0.1, 0.1, 3.1415e-3)
    want, wcost, wst = ctx.align_pairs(p.ref_ranges[sel], p.new_ranges[sel], geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds[sel], mode=capi.SCORE_EXACT)
    assert ((wst["status"] & 0xffff) == 0).all()
    g = capi.ShardGroup([0])
    got, cost, st = g.align_pairs(p.ref_ranges[sel], p.new_ranges[sel], geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds[sel], mode=capi.SCORE_EXACT)
    g.close()
    assert ((st["status"] & 0xffff) == 0).all()
    assert np.array_equal(got, want) and np.array_equal(cost, wcost)
