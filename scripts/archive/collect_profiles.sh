# usage: bash scripts/collect_profiles.sh <gpurun_out subdir of scripts/r2_profile.sh> <prefix, e.g. r02>
# copies the evidence of one profile run from scratch (gpurun_out/) into the tracked profiles/ directory
SRC=gpurun_out/$1; P=profiles/$2
mkdir -p ${P}_pmc
cp $SRC/bench.json ${P}_bench.json
cp $SRC/ubench_valu.txt ${P}_ubench_valu.txt
cp $SRC/pmc/trace/t_kernel_stats.csv ${P}_kernel_stats.csv
for k in 1 2 3 4 5; do
  # the rows of the fused pairs kernels only (the counter files also list torch's own kernels)
  head -1 $SRC/pmc/p$k/p_counter_collection.csv > ${P}_pmc/p${k}_k_align_pairs.csv
  grep "k_align_pairs" $SRC/pmc/p$k/p_counter_collection.csv >> ${P}_pmc/p${k}_k_align_pairs.csv
done
python scripts/pmc_summary.py $SRC/pmc ${P}_pmc_summary.json > /dev/null
python scripts/isa_mix.py --ubench ${P}_ubench_valu.txt --out ${P}_isa_mix.json --dump ${P}_score_loop_isa.txt > /dev/null
ls -la profiles | tail -20
