export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name=${grp%% *}; d=/tmp/prof_f32_$name; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o pmc -- python bench.py --precision f32 --steps 1 --warmup 1 --no-cpu-baseline --no-exact > gpurun_out/r04_f32_pmc_$name.log 2>&1
  for k in kernel_trace counter_collection; do f=$(find $d -name "*_$k.csv" | head -1); [ -n "$f" ] && [ "$f" != "$d/pmc_$k.csv" ] && mv "$f" $d/pmc_$k.csv; done
  python scripts/summarize_prof.py $d pmc gpurun_out/r04_f32_pmc_$name.txt
  grep "k_decoder<1, 0>" gpurun_out/r04_f32_pmc_$name.txt
done
d=/tmp/prof_f32_ks; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python bench.py --precision f32 --steps 1 --warmup 1 --no-cpu-baseline --no-exact > gpurun_out/r04_f32_bench_under_rocprof.log 2>&1
f=$(find $d -name "*_kernel_stats.csv" | head -1); mv "$f" $d/ks_kernel_stats.csv; python scripts/summarize_prof.py $d ks gpurun_out/r04_f32_kernel_stats.txt
grep "k_decoder<1, 0>\|k_decoder<0, 0>\|k_decoder<1, 1>" gpurun_out/r04_f32_kernel_stats.txt | cut -c1-120
