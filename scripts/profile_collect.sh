# usage (build container): bash scripts/profile_collect.sh <dir under gpurun_out written by scripts/profile_run.sh> <prefix, e.g. r03>
# copies the evidence of one profile run from scratch (gpurun_out/) into the tracked profiles/ directory
SRC=gpurun_out/$1; P=profiles/$2
mkdir -p ${P}_pmc
cp $SRC/bench.json ${P}_bench.json
cp $SRC/bench_one_at_a_time.json ${P}_bench_one_at_a_time.json
for sc in f64 f32; do [ -f $SRC/bench_$sc.json ] && cp $SRC/bench_$sc.json ${P}_bench_$sc.json; done
cp $SRC/ubench_valu.txt ${P}_ubench_valu.txt
cp $SRC/isa_mix.json ${P}_isa_mix.json
cp $SRC/score_loop_isa.txt ${P}_score_loop_isa.txt
[ -f $SRC/isa_mix_f64.json ] && cp $SRC/isa_mix_f64.json ${P}_isa_mix_f64.json && cp $SRC/score_loop_isa_f64.txt ${P}_score_loop_isa_f64.txt
cp $SRC/trace1/t_kernel_stats.csv ${P}_kernel_stats.csv                 # one batch at a time
cp $SRC/trace2/t_kernel_stats.csv ${P}_kernel_stats_two_in_flight.csv
for k in 1 2 3 4 5; do
  # the rows of the fused pairs kernels only (the counter files also list torch's own kernels)
  head -1 $SRC/pmc/p$k/p_counter_collection.csv > ${P}_pmc/p${k}_k_align_pairs.csv
  grep "k_align_pairs" $SRC/pmc/p$k/p_counter_collection.csv >> ${P}_pmc/p${k}_k_align_pairs.csv
done
cp $SRC/pmc_summary.json ${P}_pmc_summary.json
if [ -f $SRC/pmc_summary_f64.json ]; then
  cp $SRC/pmc_summary_f64.json ${P}_pmc_summary_f64.json
  mkdir -p ${P}_pmc_f64
  for k in 1 2 3 4 5; do
    head -1 $SRC/pmc_f64/p$k/p_counter_collection.csv > ${P}_pmc_f64/p${k}_k_align_pairs.csv
    grep "k_align_pairs" $SRC/pmc_f64/p$k/p_counter_collection.csv >> ${P}_pmc_f64/p${k}_k_align_pairs.csv
  done
  grep -h "Name\|k_align_pairs" $SRC/pmc_f64/trace/t_kernel_stats.csv > ${P}_kernel_stats_f64.csv
fi
[ -f $SRC/phase_budget_f64.json ] && cp $SRC/phase_budget_f64.json ${P}_phase_budget_f64.json
[ -f $SRC/shard_timing_g1.json ] && cp $SRC/shard_timing_g1.json ${P}_shard_timing_g1.json
cp $SRC/phase_budget.json ${P}_phase_budget.json
cp $SRC/phase_budget_f32.json ${P}_phase_budget_f32.json
cp $SRC/phase_budget_config5.json ${P}_phase_budget_config5.json
[ -f $SRC/live/live_timeline.json ] && cp $SRC/live/live_timeline.json ${P}_live_timeline.json
for f in phase_budget_361 bench_self_launched_rccl bench_sharded_capi_g1 bench_sharded_capi_virtual8 shape_sweep; do [ -f $SRC/$f.json ] && cp $SRC/$f.json ${P}_$f.json; done
[ -f $SRC/live_replicas.txt ] && grep replicas $SRC/live_replicas.txt > ${P}_live_replicas.txt
[ -f $SRC/verify_config3.json ] && python - <<PY
import json
out = {}
for w in ("config3", "config4", "random", "converged", "config5", "short"):
    try:
        d = json.load(open("$SRC/verify_%s.json" % w))
        out[w] = {k: v for k, v in d.items() if k != "runs"}
        out[w]["runs"] = [{k: r[k] for k in ("pairs", "beams", "cs", "P", "I", "evaluations_checked", "points_checked", "max_err", "max_err_over_bound", "max_bound_over_half_tau", "max_err_over_half_tau", "points_binned_differently", "arbitrated_mean")} for r in d["runs"]]
    except Exception as e:
        out[w] = {"error": str(e)}
for w in ("binning_config3_x400", "binning_random_x40"):
    try:
        d = json.load(open("$SRC/verify_%s.json" % w))
        out[w] = {"points_checked": d["points_checked"], "binning": d["binning"], "launches_per_run": d["runs"][0].get("launches"), "runs": len(d["runs"])}
    except Exception as e:
        out[w] = {"error": str(e)}
json.dump(out, open("${P}_margin_verification.json", "w"), indent=1)
PY
ls -la profiles | tail -24
