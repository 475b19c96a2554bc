set -x
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_fullsize.py -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/r2b/pytest.log
for sc in f32 exact f64; do
python bench.py --steps 100 --warmup 10 --score $sc --cpu-sample 0 --no-latency > gpurun_out/r2b/bench_$sc.json 2> gpurun_out/r2b/bench_$sc.err
done
cat gpurun_out/r2b/pytest.log; for sc in f32 exact f64; do python -c "
import json,sys
d=json.load(open('gpurun_out/r2b/bench_$sc.json')); print('$sc', d['value'], d['roofline']['kernel_ms'], d['extra'])"; done
