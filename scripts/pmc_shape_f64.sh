# usage (GPU box): bash scripts/pmc_shape_f64.sh <beams> <cell> -- the fp64-score kernel of the fused pairs path on one shape: VALU and LDS
# busy fractions, LDS bank conflicts, instructions per 64 point evaluations (why 1081 beams at 0.25 m cells run a launch in 5.1 ms
# where 0.5 m cells take 4.3: DESIGN 5)
B=$1; C=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_f64_${B}_$C; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tests/campaigns/shape_sweep.py --out $OUT/sweep.json --modes f64 --beams $B --cells $C --frames 60 --launches 3 --oracle-pairs 1"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -o p -- $CMD > $OUT/b.log 2>&1
python - <<PY
import csv, collections, json
def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    names = collections.Counter()
    for r in csv.DictReader(open(path)):
        if "k_align_pairs<1," in r["Kernel_Name"] and int(float(r["Grid_Size"])) == 512 * int(float(r["Workgroup_Size"])):
            acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
            names[r["Kernel_Name"].split("(")[0]] += 1
    return {k: sum(sum(v) for v in d.values()) / len(d) for k, d in acc.items()}, names
(a, na), (b, nb) = load("$OUT/a/p_counter_collection.csv"), load("$OUT/b/p_counter_collection.csv")
cyc = b["GRBM_GUI_ACTIVE"] / 8.0
sw = json.load(open("$OUT/sweep.json"))["rows"][0]
print(json.dumps({"beams": $B, "cell": $C, "kernel": list(na)[0], "ms_per_launch_incl_redo": sw.get("ms_per_launch"), "align_per_s": sw["align_per_s"],
       "valu_busy_frac": a["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024), "insts_valu_per_launch": a["SQ_INSTS_VALU"], "insts_lds_per_launch": a["SQ_INSTS_LDS"],
       "lds_active_frac": b["SQ_LDS_IDX_ACTIVE"] * 4 / (cyc * 1024), "lds_bank_conflict_frac_of_lds_active": b["SQ_LDS_BANK_CONFLICT"] / b["SQ_LDS_IDX_ACTIVE"],
       "kernel_cycles_per_xcd": cyc}))
PY
