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
