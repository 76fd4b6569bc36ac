#!/bin/bash
# build_variant.sh <tag> <extra hipcc flags...>  ->  hortimapping_amd/variants/libhortihip_<tag>.so (git-ignored, travels with gpurun)  (timing experiments only)
set -e
tag=$1; shift
cd "$(dirname "$0")/../hortimapping_amd/csrc"
mkdir -p ../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I . "$@" -o ../variants/libhortihip_$tag.so \
  hm_pack.hip hm_decoder.hip hm_decoder_h.hip hm_decoder_p.hip hm_decoder_any.hip hm_prep.hip hm_debug.hip hm_normal_eq.hip hm_solve.hip hm_render.hip hm_optimize.hip hm_mesh.hip hm_metrics.hip hm_api.hip
