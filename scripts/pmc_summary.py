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
# (the launch that did the work: a run's gated redo launches are other instances of k_align_pairs whose workgroups exit at once)
_valu = collections.defaultdict(list)
for r in csv.DictReader(open(f"{src}/p1/p_counter_collection.csv")):
    if "k_align_pairs<" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
        _valu[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
_busiest = max(_valu, key=lambda k: max(_valu[k])) if _valu else None
KERNEL = _busiest if _busiest else next((k for k in ("k_align_pairs<1, 9, false", "k_align_pairs<1, 8, false", "k_align_pairs<0, 3, false, true, true", "k_align_pairs<0, 3, false, false, true",
                           "k_align_pairs<0, 3, false, true, false", "k_align_pairs<0, 3, false, false, false",
                           "k_align_pairs<0, 3, false, true", "k_align_pairs<0, 3, false, false", "k_align_pairs<0, 3, false",
                           "k_align_pairs<0, 2, false") if any(k in n for n in names)), "k_align_pairs")
FULL_NAME = next((n.split("(")[0].replace("void ", "") for n in sorted(names) if KERNEL in n), KERNEL)


def describe(name):
    """What the template arguments of k_align_pairs<MODE, PATH, CLUSTER, ARB, NOCLIP, SWARM> say about the launch."""
    import re
    m = re.search(r"<([^>]*)>", name)
    a = [x.strip() for x in m.group(1).split(",")] if m else []
    a += ["false"] * (5 - len(a)) + (["2"] if len(a) < 6 else [])
    mode = "fp32 Gaussian score" if a[0] == "0" else "fp64 score"
    form = {"0": "bitmap table, true division", "1": "bitmap table, power-of-two cells", "2": "dense u16 table (entries in 16-byte units)",
            "3": "dense u16 table (entries are byte addresses)", "8": "dense u16 table, fp64 records, true division",
            "9": "dense u16 table, fp64 records, power-of-two cells"}.get(a[1], a[1])
    arb = "EXACT mode: near-tie comparisons arbitrated with the fp64 score" if a[3] == "true" else "plain (no arbitration)"
    swarm = {"0": "swarm in LDS", "1": "swarm in its HBM workspace", "2": "both swarm homes compiled in"}.get(a[5], a[5])
    return "%s, %s, %s, %s, %s%s" % (mode, form, arb, "cluster of workgroups per pair" if a[2] == "true" else "one workgroup per pair",
                                     "no frame-clipping trips, " if a[4] == "true" else "", swarm)


def code_object_resources(name):
    """VGPRs / spills / scratch / static LDS of the kernel from the shipped library's code object metadata (rocprofv3's
    dispatch columns report allocation granules and 0 for dynamic LDS, which is what round 2's summary mistook)."""
    try:
        import re, subprocess, tempfile
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, here)
        import isa_mix
        lib = os.environ.get("NDTPSO_LIB", os.path.join(os.path.dirname(here), "ndtpso_slam_amd", "lib", "libndtpso_hip.so"))
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(isa_mix.extract_code_object(lib))
            f.flush()
            out = subprocess.check_output([os.path.join(isa_mix.LLVM, "llvm-readelf"), "--notes", f.name], text=True)
        want = name.replace(" ", "")
        for b in out.split("- .agpr_count")[1:]:
            sym = re.search(r"\.name:\s+(\S+)", b).group(1)
            dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
            if dem.replace(" ", "").replace("ndtpso::", "") == want.replace("ndtpso::", ""):
                g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))  # noqa: E731
                return {"vgpr_count": g("vgpr_count"), "vgpr_spill_count": g("vgpr_spill_count"), "sgpr_count": g("sgpr_count"),
                        "scratch_bytes_per_lane": g("private_segment_fixed_size"), "static_lds_bytes": g("group_segment_fixed_size"),
                        "source": "llvm-readelf --notes of the gfx950 code object in libndtpso_hip.so"}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}
    return {"error": "kernel not found in the code object"}
# A gated redo launch can be the SAME kernel with another workgroup size (the fp64 dense form's largest-table launch: 1024
# threads against 512; every workgroup exits at once): the working launches are the ones of the smaller workgroup size.
rows_by_wg = collections.defaultdict(list)
for p in ("p1", "p2", "p3", "p4", "p5"):
    for r in csv.DictReader(open(f"{src}/{p}/p_counter_collection.csv")):
        # (Grid_Size = threads: the bench's launches are `pairs` workgroups -- the exact mode's start-up check runs the same
        # kernel on 520 small pairs once per process and must not be averaged in)
        if KERNEL in r["Kernel_Name"] and int(float(r["Grid_Size"])) == pairs * int(float(r["Workgroup_Size"])):
            rows_by_wg[int(float(r["Workgroup_Size"]))].append(r)
def _wg_valu(w):
    v = [float(r["Counter_Value"]) for r in rows_by_wg[w] if r["Counter_Name"] == "SQ_INSTS_VALU"]
    return max(v) if v else 0.
WG = max(rows_by_wg, key=_wg_valu) if rows_by_wg else 0
vals = collections.defaultdict(list)
disp = {}
for r in rows_by_wg.get(WG, []):
    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    disp = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size")}
mean = {k: sum(v) / len(v) for k, v in vals.items()}
trace_path = f"{src}/trace/t_kernel_trace.csv"
if os.path.exists(trace_path):   # per dispatch: the working launches only, warm ones (the second half)
    d = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(trace_path))
         if KERNEL in r["Kernel_Name"] and int(float(r["Workgroup_Size_X"])) == WG and int(float(r["Grid_Size_X"])) == pairs * WG]
    d = d[len(d) // 2:] if len(d) >= 4 else d
    kern_ns, kern_calls = (sum(d) / len(d) if d else float("nan")), len(d)
else:
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
out = {"kernel": FULL_NAME, "kernel_is": describe(FULL_NAME), "workload": f"{pairs} pairs, {P} x {I}, {beams} beams; the gated redo "
       "launches exit immediately and are not included", "source": src,
       "code_object": code_object_resources(FULL_NAME),
       "rocprofv3_dispatch_columns": dict(disp, note="allocation granules as rocprofv3 reports them; dynamic LDS shows as 0 -- see code_object and "
                                          "ndtpso_align_pairs_describe for the real figures"),
       "counters_mean_per_launch": mean, "derived": derived}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(derived, indent=1))
