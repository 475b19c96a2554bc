cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_campaigns; mkdir -p $O
(time NDTPSO_RANDOM_CASES=10000 NDTPSO_RANDOM_SEED=660001 timeout 1500 python -m pytest "tests/test_gpu_exact.py::test_exact_mode_equals_fp64_mode_on_random_configurations" -m gpu -q -s) > $O/random_exact.log 2>&1
(time NDTPSO_RANDOM_CASES=600 NDTPSO_RANDOM_SEED=660002 timeout 900 python -m pytest "tests/test_gpu_parity.py::test_randomised_configurations" -m gpu -q -s) > $O/random_modes.log 2>&1
(time SOAK_SCORE=exact timeout 1500 python tests/campaigns/soak_replay.py 3000 --oracle) > $O/soak.log 2>&1
(time timeout 900 python tests/campaigns/fuzz_campaign.py 150 660003) > $O/fuzz.log 2>&1
(time timeout 900 python tests/campaigns/host_fuzz_campaign.py 30 660004) > $O/host_fuzz.log 2>&1
(time NDTPSO_TAG_STRESS=200000 timeout 900 python -m pytest tests/test_gpu_cluster_tags.py -m gpu -q) > $O/tags.log 2>&1
for f in random_exact random_modes soak fuzz host_fuzz tags; do echo "== $f"; grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostn\|^Librccl\|amdgpu.ids\|^$" $O/$f.log | tail -6; done
