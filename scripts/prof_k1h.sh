#!/bin/bash
# usage: prof_k1h.sh <variant tag | -> <n> ...   prints avg ns of k_decoder_h kernels (rocprofv3 kernel trace)
tag=$1; shift
[ "$tag" != "-" ] && cp build/libhortihip_$tag.so hortimapping_amd/libhortihip.so
export TMPDIR=/tmp
for n in "$@"; do
  rm -rf /tmp/pk && mkdir -p /tmp/pk
  HM_PREC=f16x3 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o p -- python scripts/gpu_time_decoder.py 256 64 $n > /tmp/pk/log 2>&1
  f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1)
  echo "variant $tag n=$n:"; grep k_decoder_h "$f" | awk -F, '{printf "   %s calls=%s avg_ns=%s\n", substr($1,1,60), $2, $4}'
done
