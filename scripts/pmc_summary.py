"""Summarises the PMC passes of scripts/pmc.sh (gpurun_out/<dir>/p1..p5) for the fused pairs kernel into
profiles/<name>.json: per-launch counter means and the derived figures DESIGN.md and bench.py quote.
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE under-reports by 2x on gfx950 (MI355X_MICROARCH.md), corrected here.
usage: python scripts/pmc_summary.py gpurun_out/pmc_final profiles/r01_pmc_summary.json [pairs P I beams]"""
import csv, json, sys, collections, math

src, dst = sys.argv[1], sys.argv[2]
pairs, P, I, beams = (int(v) for v in (sys.argv[3:7] if len(sys.argv) >= 7 else (512, 70, 70, 1081)))
import os
# dense form with byte-address entries (what config 3 runs): the arbitrating kernel of the exact mode, the plain fp32-score
# kernel, or the general dense form -- whichever the profiled run launched
names = {r["Kernel_Name"] for r in csv.DictReader(open(f"{src}/p1/p_counter_collection.csv"))}
KERNEL = next((k for k in ("k_align_pairs<0, 3, false, true, true>", "k_align_pairs<0, 3, false, false, true>",
                           "k_align_pairs<0, 3, false, true, false>", "k_align_pairs<0, 3, false, false, false>",
                           "k_align_pairs<0, 3, false, true>", "k_align_pairs<0, 3, false, false>", "k_align_pairs<0, 3, false>",
                           "k_align_pairs<0, 2, false") if any(k in n for n in names)), "k_align_pairs")
vals = collections.defaultdict(list)
disp = {}
for p in ("p1", "p2", "p3", "p4", "p5"):
    for r in csv.DictReader(open(f"{src}/{p}/p_counter_collection.csv")):
        if KERNEL not in r["Kernel_Name"]:
            continue
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        disp = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size")}
mean = {k: sum(v) / len(v) for k, v in vals.items()}
stats = [r for r in csv.DictReader(open(f"{src}/trace/t_kernel_stats.csv")) if KERNEL in r["Name"]]
kern_ns = float(stats[0]["AverageNs"]) if stats else float("nan")
kern_calls = int(float(stats[0]["Calls"])) if stats else 0
evals = 1 + P + P * I
# cost evaluations include the replays of the exact-order scheme (measured mean of the bench workload: +2.3 %)
chunks = pairs * evals * 1.0228 * math.ceil(beams / 64)
cyc_xcd = mean["GRBM_GUI_ACTIVE"] / 8
derived = {
    "kernel_avg_ns_from_kernel_trace": kern_ns,
    "kernel_trace_calls": kern_calls,
    "kernel_cycles_per_xcd": cyc_xcd,
    "valu_instr_per_64_point_evals": mean["SQ_INSTS_VALU"] / chunks,
    "valu_cycles_per_instr": mean["SQ_ACTIVE_INST_VALU"] * 4 / mean["SQ_INSTS_VALU"],
    # the SQ_* cycle counters tick once per 4 cycles and are summed over the 1024 SIMDs
    "valu_busy_frac": mean["SQ_ACTIVE_INST_VALU"] * 4 / (cyc_xcd * 1024),
    "lds_active_frac": mean["SQ_LDS_IDX_ACTIVE"] * 4 / (cyc_xcd * 1024),
    "lds_bank_conflict_frac": mean["SQ_LDS_BANK_CONFLICT"] / mean["SQ_LDS_IDX_ACTIVE"],
    "mean_waves_per_simd": mean["SQ_WAVE_CYCLES"] * 4 / (cyc_xcd * 1024),
    "hbm_read_bytes_per_launch_FETCH_SIZE_x2_KiB_units": mean["FETCH_SIZE"] * 1024 * 2,
    "hbm_write_bytes_per_launch": mean["WRITE_SIZE"] * 1024,
}
out = {"kernel": f"{KERNEL} (fp32 score, dense table, per-alignment window), {pairs} pairs, {P}x{I}; the gated redo launches "
                 "exit immediately and are not included", "source": src, "dispatch": disp,
       "counters_mean_per_launch": mean, "derived": derived}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(derived, indent=1))
