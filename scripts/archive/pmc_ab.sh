# quick A/B of two library builds with one PMC pass each: bash scripts/pmc_ab.sh libA.so libB.so
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_$(basename $L .so); mkdir -p $OUT
  NDTPSO_LIB=$L rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-latency > $OUT/log.txt 2>&1
  python - <<PY
import csv,collections
rows=list(csv.DictReader(open("$OUT/p_counter_collection.csv")))
agg=collections.defaultdict(list)
for r in rows:
    if 'k_align_pairs' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print("$L", {k: round(sum(v)/len(v)/1e6,1) for k,v in agg.items()})
PY
done
