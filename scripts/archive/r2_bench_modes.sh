# usage: bash scripts/r2_bench_modes.sh <outdir> [modes...]   -- quick throughput of the score modes (100 steps)
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for sc in ${@:-f32 exact}; do
python bench.py --steps 100 --warmup 10 --score $sc --cpu-sample 0 --no-latency > $OUT/bench_$sc.json 2> $OUT/bench_$sc.err
python -c "
import json,sys
d=json.load(open('$OUT/bench_$sc.json')); print('$sc', round(d['value']), d['roofline']['kernel_ms'])"
done
