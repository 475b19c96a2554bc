set -x
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r2a/pytest.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu > gpurun_out/r2a/ubench_valu.txt 2>&1
python bench.py --steps 200 --warmup 20 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
nproc > gpurun_out/r2a/nproc.txt; lscpu | head -20 >> gpurun_out/r2a/nproc.txt
tail -5 gpurun_out/r2a/pytest.log; cat gpurun_out/r2a/ubench_valu.txt; cat gpurun_out/r2a/bench.json
