# builds: for v in NOA NOB NOMFMA; do bash scripts/build_variant.sh abl_$v -DHM_EXPERIMENTAL -DHM_ABL_$v; done; bash scripts/build_variant.sh abl_NOAB -DHM_EXPERIMENTAL -DHM_ABL_NOA -DHM_ABL_NOB
for v in "" abl_NOA abl_NOB abl_NOAB abl_NOMFMA; do
  if [ -n "$v" ]; then export HORTIHIP_LIB=$PWD/hortimapping_amd/variants/libhortihip_$v.so; else unset HORTIHIP_LIB; fi
  echo "== ${v:-product}"
  python scripts/gpu_sweep_k1h.py 256 0 2>&1 | grep -v amdgpu.ids | tail -3
done
