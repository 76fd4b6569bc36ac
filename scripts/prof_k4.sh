export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=/tmp/prof_k4; rm -rf $d; mkdir -p $d
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python $R/bench.py --groups 1 --steps 2 --warmup 1 --no-cpu-baseline --no-exact --no-shipped > /dev/null 2>&1)
f=$(find $d -name "*kernel_stats.csv" | head -1)
grep -E "k_normal_eq|k_decoder_h|k_solve" $f | cut -d, -f1-4 | cut -c1-160
