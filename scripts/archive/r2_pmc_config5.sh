# PMC passes for BASELINE config 5 (256 pairs, one launch) and for one clustered pair; usage: bash scripts/r2_pmc_config5.sh <outdir>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for what in "256 exact 2" "1 exact 3"; do
  tag=$(echo $what | cut -d' ' -f1)
  CMD="python $GRAFT_REPO_ROOT/scripts/config5_profile.py $what"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_${tag}_trace -o t -- $CMD > $OUT/c5_${tag}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/c5_${tag}_p1 -o p -- $CMD > $OUT/c5_${tag}_p1.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --output-format csv -d $OUT/c5_${tag}_p2 -o p -- $CMD > $OUT/c5_${tag}_p2.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c5_${tag}_p3 -o p -- $CMD > $OUT/c5_${tag}_p3.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c5_${tag}_p4 -o p -- $CMD > $OUT/c5_${tag}_p4.log 2>&1
done
python - <<PY
import csv, glob, collections, json
res = {}
for tag in ("256", "1"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/c5_%s_p*/p_counter_collection.csv" % tag):
        for r in csv.DictReader(open(f)):
            if "k_align_pairs<0" in r["Kernel_Name"]:
                per[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    stats = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open("$OUT/c5_%s_trace/t_kernel_stats.csv" % tag)) if "k_align_pairs<0" in r["Name"]}
    for name, vals in per.items():
        mean = {k: sum(v) / len(v) for k, v in vals.items()}
        if mean.get("SQ_WAVES", 0) < 64:      # the gated redo launches: every workgroup exits at once
            continue
        d = {"kernel": name, "launch": "%s pair(s) of 2048 x 200, 2048 beams, 0.25 m cells" % tag,
             "kernel_avg_ms": float(stats[name]["AverageNs"]) / 1e6 if name in stats else None, "counters_mean_per_launch": mean}
        if "GRBM_GUI_ACTIVE" in mean and "SQ_ACTIVE_INST_VALU" in mean:
            cyc = mean["GRBM_GUI_ACTIVE"] / 8
            d["derived"] = {"valu_busy_frac_of_all_1024_simds": mean["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024),
                            "mean_waves_per_simd_over_all_1024": mean["SQ_WAVE_CYCLES"] * 4 / (cyc * 1024),
                            "valu_cycles_per_instr": mean["SQ_ACTIVE_INST_VALU"] * 4 / mean["SQ_INSTS_VALU"],
                            "lds_active_frac": mean.get("SQ_LDS_IDX_ACTIVE", 0) * 4 / (cyc * 1024),
                            "hbm_bytes_per_launch": mean.get("FETCH_SIZE", 0) * 2048 + mean.get("WRITE_SIZE", 0) * 1024}
        res["pairs_%s %s" % (tag, name)] = d
json.dump(res, open("$OUT/config5_pmc_summary.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_mean_per_launch"} for k, v in res.items()}, indent=1))
PY
