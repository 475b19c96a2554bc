# usage: bash scripts/r2_pmc_ab.sh <outdir> lib1 lib2 ... : instruction-count PMC pass of the bench for several library variants
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = cur ]; then unset NDTPSO_LIB; else export NDTPSO_LIB=$GRAFT_REPO_ROOT/ndtpso_slam_amd/lib/variants/$v.so; fi
  CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-latency --score ${SCORE:-f32}"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/$v.p1 -o p -- $CMD > $OUT/$v.p1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/$v.p2 -o p -- $CMD > $OUT/$v.p2.log 2>&1
done
python - <<PY
import csv, glob, collections
for v in "$*".split():
    vals = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s.p*/**/p_counter_collection.csv" % v, recursive=True) + glob.glob("$OUT/%s.p*/p_counter_collection.csv" % v):
        for r in csv.DictReader(open(f)):
            if "k_align_pairs<0, 3, false" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {k: round(sum(x)/len(x)) for k, x in sorted(vals.items())})
PY
