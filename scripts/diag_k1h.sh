#!/bin/bash
# Timing-only experiments on the f16x3 decoder's K loop (HM_DIAG variants, results numerically meaningless).
# Build here:  scripts/diag_k1h.sh build      Run on the GPU box:  scripts/diag_k1h.sh run
set -e
SRC="hm_pack.hip hm_decoder.hip hm_decoder_h.hip hm_normal_eq.hip hm_solve.hip hm_render.hip hm_optimize.hip hm_mesh.hip hm_metrics.hip hm_api.hip"
if [ "$1" = build ]; then
  for d in ${DIAGS:-1 2 3}; do
    (cd hortimapping_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I . -DHM_DIAG=$d $DIAGFLAGS \
       -o ../../build/libhortihip_diag$d.so $SRC) &
  done
  wait
else
  cp hortimapping_amd/libhortihip.so build/libhortihip_diag0.so
  for d in 0 ${DIAGS:-1 2 3}; do
    cp build/libhortihip_diag$d.so hortimapping_amd/libhortihip.so
    echo "== HM_DIAG=$d"
    HM_PREC=f16x3 python scripts/gpu_time_decoder.py 256 64 1024
  done
  cp build/libhortihip_diag0.so hortimapping_amd/libhortihip.so
fi
