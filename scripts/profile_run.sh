# Evidence run of one round on the GPU box: microbenchmark, ISA mix, bench line, rocprofv3 kernel traces, PMC passes,
# per-phase time budgets.  Everything lands in gpurun_out/<dir>/; `bash scripts/profile_collect.sh <dir> rNN` (build
# container) copies what is to be judged into profiles/.
#   usage (GPU box): bash scripts/profile_run.sh <dir under gpurun_out>
set -x
D=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$D
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/ubench_valu.hip -o /tmp/ubench_valu 2>/dev/null && /tmp/ubench_valu > $OUT/ubench_valu.txt 2>&1
python scripts/isa_mix.py --kernel "k_align_pairs<0, 3, false, true, true, 0, false>" --ubench $OUT/ubench_valu.txt --out $OUT/isa_mix.json --dump $OUT/score_loop_isa.txt > /dev/null
python scripts/isa_mix.py --kernel "k_align_pairs<1, 9, false, false, false, 2, false>" --marker v_rndne_f64 --marker-span 90 --ubench $OUT/ubench_valu.txt --out $OUT/isa_mix_f64.json --dump $OUT/score_loop_isa_f64.txt > /dev/null
cp $OUT/isa_mix.json profiles/r06_isa_mix.json   # the bench line's roofline.floor reads them
cp $OUT/isa_mix_f64.json profiles/r06_isa_mix_f64.json
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --pipeline 1 --no-latency --cpu-sample 0 > $OUT/bench_one_at_a_time.json 2>> $OUT/bench.err
# the other two score modes of the same workload (fp64 score throughout; plain fp32 score), two batches in flight and one at a time
for sc in f64 f32; do timeout 300 python bench.py --score $sc --no-latency --cpu-sample 0 > $OUT/bench_$sc.json 2>> $OUT/bench.err; done
# kernel traces (own runs): one batch at a time -- the per-launch duration the roofline divides by -- and two in flight
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o t -- python $GRAFT_REPO_ROOT/bench.py --pipeline 1 --steps 40 --warmup 20 --cpu-sample 0 --no-latency > $OUT/trace1.log 2>&1)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -o t -- python $GRAFT_REPO_ROOT/bench.py --pipeline 2 --steps 40 --warmup 20 --cpu-sample 0 --no-latency > $OUT/trace2.log 2>&1)
bash scripts/pmc.sh $D/pmc --pipeline 1 > $OUT/pmc.log 2>&1
python scripts/pmc_summary.py $OUT/pmc $OUT/pmc_summary.json > $OUT/pmc_summary.log 2>&1
bash scripts/pmc.sh $D/pmc_f64 --pipeline 1 --score f64 > $OUT/pmc_f64.log 2>&1
python scripts/pmc_summary.py $OUT/pmc_f64 $OUT/pmc_summary_f64.json > $OUT/pmc_summary_f64.log 2>&1
timeout 300 python scripts/phase_budget.py --config config3 --score f64 --out $OUT/phase_budget_f64.json > $OUT/budget_f64.log 2>&1
timeout 300 python scripts/shard_timing.py --out $OUT/shard_timing_g1.json > $OUT/shard_timing.log 2>&1
timeout 300 python scripts/phase_budget.py --config config3 --score exact --out $OUT/phase_budget.json > $OUT/budget.log 2>&1
timeout 300 python scripts/phase_budget.py --config beams361 --score exact --out $OUT/phase_budget_361.json >> $OUT/budget.log 2>&1
# the two launch conventions of N GPUs as far as one GPU goes: bench.py starting its own rank (RCCL path), the one-process sharded
# entry on the one device through real RCCL, and eight VIRTUAL shards on the one device (plumbing only, flagged in the line)
NDTPSO_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --no-latency --cpu-sample 0 > $OUT/bench_self_launched_rccl.json 2>> $OUT/bench.err
timeout 300 python bench.py --sharded-capi --gpus 1 > $OUT/bench_sharded_capi_g1.json 2>> $OUT/bench.err
NDTPSO_SHARD_VIRTUAL=8 timeout 300 python bench.py --sharded-capi --gpus 8 --steps 40 > $OUT/bench_sharded_capi_virtual8.json 2>> $OUT/bench.err
# replicas of the live sequence in one process, defaults and the round-5 behaviour (spinning waiters, clusters whatever is in flight)
REPLICAS=1,4,8,16,32,64 timeout 600 python scripts/replicas_wait_ab.py "" "NDTPSO_WAIT=spin,NDTPSO_CLUSTER_MAX_INFLIGHT=1000" > $OUT/live_replicas.txt 2>&1
timeout 300 python scripts/phase_budget.py --config config3 --score f32 --out $OUT/phase_budget_f32.json >> $OUT/budget.log 2>&1
timeout 400 python scripts/phase_budget.py --config config5 --score exact --out $OUT/phase_budget_config5.json >> $OUT/budget.log 2>&1
# device timeline of the live sequence (C++ drop-in, node replay): kernel spans folded over the scans
timeout 300 python scripts/live_timeline.py run $OUT/live 200 > $OUT/live_timeline.log 2>&1; rm -rf $OUT/live/trace
[ -n "$SKIP_VERIFY" ] || timeout 600 python scripts/verify_margin.py --workload short > $OUT/verify_short.json 2>> $OUT/verify.err
[ -n "$SKIP_VERIFY" ] || for w in config3 config4 random converged config5; do timeout 600 python scripts/verify_margin.py --workload $w $( [ $w = config5 ] && echo --pairs 130 ) > $OUT/verify_$w.json 2>> $OUT/verify.err; done
# the binning's statistics over 1e12 point evaluations (config 3 x 400 launches of other pairs and seeds) and random configurations
[ -n "$SKIP_VERIFY" ] || timeout 900 python scripts/verify_margin.py --workload config3 --repeat 400 > $OUT/verify_binning_config3_x400.json 2>> $OUT/verify.err
[ -n "$SKIP_VERIFY" ] || timeout 900 python scripts/verify_margin.py --workload random --repeat 40 > $OUT/verify_binning_random_x40.json 2>> $OUT/verify.err
[ -n "$SKIP_SWEEP" ] || timeout 1500 python tests/campaigns/shape_sweep.py --out $OUT/shape_sweep.json --modes exact,f64 > $OUT/shape_sweep.log 2>&1
cat $OUT/bench.json | head -c 600; echo; tail -3 $OUT/bench.err; cat $OUT/pmc_summary.log | tail -14
