export TMPDIR=/tmp
d=/tmp/prof_lds; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $d -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact > gpurun_out/r01d_pmc_LDS.log 2>&1
for k in kernel_trace counter_collection; do f=$(find $d -name "*_$k.csv" | head -1); [ -n "$f" ] && [ "$f" != "$d/pmc_$k.csv" ] && mv "$f" $d/pmc_$k.csv; done
python scripts/summarize_prof.py $d pmc gpurun_out/r01d_pmc_LDS.txt
grep "k_decoder_h<1, 0>" gpurun_out/r01d_pmc_LDS.txt
tail -3 gpurun_out/r01d_pmc_LDS.log | cut -c1-200
