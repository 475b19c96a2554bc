# round-2 evidence run: VALU microbenchmark, bench (default = exact mode), rocprofv3 kernel trace + PMC passes
set -x
OUT=gpurun_out/r2prof
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/ubench_valu.hip -o /tmp/ubench_valu 2>/dev/null && /tmp/ubench_valu > $OUT/ubench_valu.txt 2>&1
# the roofline floor of the bench line from THIS box's microbenchmark and the shipped library (collect_profiles.sh redoes it from the same files)
python scripts/isa_mix.py --ubench $OUT/ubench_valu.txt --out profiles/r02_isa_mix.json --dump profiles/r02_score_loop_isa.txt > /dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
bash scripts/pmc.sh r2prof/pmc > $OUT/pmc.log 2>&1
python scripts/pmc_summary.py gpurun_out/r2prof/pmc $OUT/pmc_summary.json > $OUT/pmc_summary.log 2>&1
cat $OUT/ubench_valu.txt | tail -12; cat $OUT/bench.json; tail -20 $OUT/pmc_summary.log; tail -3 $OUT/bench.err
