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
