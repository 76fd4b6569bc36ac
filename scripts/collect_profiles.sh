#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration on the GPU box (run through gpurun):
#   scripts/collect_profiles.sh <tag> [bench args...]
# kernel trace + stats in one run, every PMC group in its own run (never combined with other trace domains).
# Raw CSVs stay in /tmp on the box; the condensed summaries land in gpurun_out/<tag>_*.txt (copy them to profiles/).
set -u
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out; mkdir -p $out
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact $*"
flat() {  # flat <dir> <prefix>: move the per-run CSVs to <dir>/<prefix>_<kind>.csv
  for k in kernel_stats kernel_trace counter_collection; do
    f=$(find $1 -name "*_$k.csv" | head -1); [ -n "$f" ] && mv "$f" $1/$2_$k.csv
  done
}
d=/tmp/prof_$tag/ks; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- $CMD > $out/${tag}_bench_under_rocprof.log 2>&1
flat $d ks; python scripts/summarize_prof.py $d ks $out/${tag}_kernel_stats.txt
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  name=${grp%% *}
  d=/tmp/prof_$tag/$name; rm -rf $d; mkdir -p $d
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o pmc -- $CMD > $out/${tag}_pmc_$name.log 2>&1
  flat $d pmc; python scripts/summarize_prof.py $d pmc $out/${tag}_pmc_$name.txt
done
tail -1 $out/${tag}_bench_under_rocprof.log | cut -c1-300
ls -la $out | grep $tag
python scripts/make_traffic_json.py $out $tag $out/${tag%%_*}_traffic.json
