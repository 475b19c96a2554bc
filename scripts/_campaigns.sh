cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_campaigns_b; mkdir -p $O
(time NDTPSO_RANDOM_CASES=40000 NDTPSO_RANDOM_SEED=660011 timeout 3000 python -m pytest "tests/test_gpu_exact.py::test_exact_mode_equals_fp64_mode_on_random_configurations" -m gpu -q -s) > $O/random_exact.log 2>&1
(time NDTPSO_RANDOM_CASES=2000 NDTPSO_RANDOM_SEED=660012 timeout 900 python -m pytest "tests/test_gpu_parity.py::test_randomised_configurations" -m gpu -q -s) > $O/random_modes.log 2>&1
(time SOAK_SCORE=exact timeout 3000 python tests/campaigns/soak_replay.py 10000 --oracle) > $O/soak.log 2>&1
(time timeout 900 python tests/campaigns/fuzz_campaign.py 400 660013) > $O/fuzz.log 2>&1
for f in random_exact random_modes soak fuzz; do echo "== $f"; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostn\|^Librccl\|amdgpu.ids\|^$" $O/$f.log | tail -6; done
