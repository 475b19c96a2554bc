mkdir -p gpurun_out/r2camp
timeout 600 python -m pytest tests/test_gpu_resident.py -x -q 2>&1 | tail -3
( timeout 1500 python tests/campaigns/fuzz_campaign.py 600 9091 ) > gpurun_out/r2camp/fuzz_600.log 2>&1; tail -2 gpurun_out/r2camp/fuzz_600.log
( timeout 1500 python tests/campaigns/host_fuzz_campaign.py 150 4242 ) > gpurun_out/r2camp/host_fuzz_150.log 2>&1; tail -2 gpurun_out/r2camp/host_fuzz_150.log
