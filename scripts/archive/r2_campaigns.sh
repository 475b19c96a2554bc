mkdir -p gpurun_out/r2camp
( SOAK_SCORE=exact timeout 1500 python tests/campaigns/soak_replay.py 3000 --oracle ) > gpurun_out/r2camp/soak_exact_3000.log 2>&1
( NDTPSO_RANDOM_CASES=1200 NDTPSO_RANDOM_SEED=777 timeout 1500 python -m pytest "tests/test_gpu_exact.py::test_exact_mode_equals_fp64_mode_on_random_configurations" -x -q -s ) > gpurun_out/r2camp/exact_random_1200.log 2>&1
tail -8 gpurun_out/r2camp/soak_exact_3000.log; tail -5 gpurun_out/r2camp/exact_random_1200.log
