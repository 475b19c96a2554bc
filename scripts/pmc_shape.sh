# usage (GPU box): bash scripts/pmc_shape.sh <beams> [cell] -- VALU / LDS busy fractions of the fused pairs kernel on another shape than the
# benchmark's (512 pairs, 70 x 70): is a short scan's launch issue-bound like the benchmark's (VALU busy 0.89) or waiting?
B=$1; C=${2:-0.5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_shape_$B; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tests/campaigns/shape_sweep.py --out $OUT/sweep.json --beams $B --cells $C --frames 60 --launches 3 --oracle-pairs 1"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
python - <<PY
import csv, collections
def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "k_align_pairs<0, 3" in r["Kernel_Name"] and int(float(r["Grid_Size"])) == 512 * int(float(r["Workgroup_Size"])):
            acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
    return {k: sum(sum(v) for v in d.values()) / len(d) for k, d in acc.items()}
a, b = load("$OUT/a/p_counter_collection.csv"), load("$OUT/b/p_counter_collection.csv")
cyc = b["GRBM_GUI_ACTIVE"] / 8.0   # summed over the 8 XCDs
print({"beams": $B, "cell": $C, "valu_busy_frac": a["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024), "valu_cycles_per_instr": a["SQ_ACTIVE_INST_VALU"] * 4 / a["SQ_INSTS_VALU"],
       "lds_active_frac": b["SQ_LDS_IDX_ACTIVE"] * 4 / (cyc * 1024), "waves_per_simd": a["SQ_WAVE_CYCLES"] * 4 / (cyc * 1024),
       "insts_valu": a["SQ_INSTS_VALU"], "kernel_cycles_per_xcd": cyc})
PY
