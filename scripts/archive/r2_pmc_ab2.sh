# usage: bash scripts/r2_pmc_ab2.sh <outdir> lib1 lib2 ... : LDS-side PMC pass (after r2_pmc_ab.sh's instruction counts)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = cur ]; then unset NDTPSO_LIB; else export NDTPSO_LIB=$GRAFT_REPO_ROOT/ndtpso_slam_amd/lib/variants/$v.so; fi
  CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-latency --score ${SCORE:-f32}"
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT/$v.p3 -o p -- $CMD > $OUT/$v.p3.log 2>&1
done
python - <<PY
import csv, glob, collections
for v in "$*".split():
    vals = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s.p3/**/p_counter_collection.csv" % v, recursive=True) + glob.glob("$OUT/%s.p3/p_counter_collection.csv" % v):
        for r in csv.DictReader(open(f)):
            if "k_align_pairs<0, 3, false" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {k: round(sum(x)/len(x)) for k, x in sorted(vals.items())})
PY
