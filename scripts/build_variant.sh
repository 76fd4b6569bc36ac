#!/bin/bash
# build_variant.sh <tag> <extra hipcc flags...>  ->  build/libhortihip_<tag>.so  (timing experiments only)
set -e
tag=$1; shift
cd "$(dirname "$0")/../hortimapping_amd/csrc"
mkdir -p ../../build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I . "$@" -o ../../build/libhortihip_$tag.so \
  hm_pack.hip hm_decoder.hip hm_decoder_h.hip hm_normal_eq.hip hm_solve.hip hm_render.hip hm_optimize.hip hm_mesh.hip hm_metrics.hip hm_api.hip
