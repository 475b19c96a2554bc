mkdir -p gpurun_out/r2t
timeout 2400 python -m pytest tests -m gpu -q -x -s 2>&1 | tail -50 > gpurun_out/r2t/pytest.log
tail -15 gpurun_out/r2t/pytest.log
