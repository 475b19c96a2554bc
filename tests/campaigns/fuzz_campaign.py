"""One-off: the resident-map differential fuzz (tests/test_gpu_resident.py) and the speculative-build test on many more
seeds and frame shapes than the suite runs.  usage: python tests/campaigns/fuzz_campaign.py [n] [seed]"""
import sys, os
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
from ndtpso_slam_amd import capi
from oracle import pyoracle
import test_gpu_resident as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx = capi.Context(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
ok = aligns = exact32 = 0
for k in range(n):
    seed = int(rng.integers(10, 10**6))
    fw, fh = int(rng.choice([8, 12, 16, 20, 30])), int(rng.choice([8, 12, 16, 20, 30]))
    cs = float(rng.choice([0.4, 0.5, 0.7, 1.0, 1.3, 2.0]))
    if os.environ.get("FUZZ_WIDE"):   # frames and cells far from the suite's: up to 150 m, down to 0.15 m
        fw, fh = int(rng.choice([8, 20, 40, 60, 100, 150])), int(rng.choice([8, 20, 40, 60, 100, 150]))
        cs = float(rng.choice([0.15, 0.2, 0.25, 0.3, 0.5, 1.0]))
        while (fw / cs) * (fh / cs) > 2000000:   # (a resident frame has fewer than 2^21 cells: DESIGN 7)
            cs *= 2
    ogcs = float(rng.choice([0.0, 0.1, 0.2, 0.25, 0.5]))
    if ogcs > cs: ogcs = 0.0
    na, ne = T.run_operation_sequence(ctx, pyoracle, seed, fw, fh, cs, ogcs)
    aligns += na
    exact32 += ne
    ok += 1
    print(f"case {k}: seed {seed} frame {fw}x{fh} cs {cs} og {ogcs}: ok", flush=True)
print(f"{ok}/{n} sequences identical to the oracle; fp32-score alignments bit-identical: {exact32}/{aligns}")
