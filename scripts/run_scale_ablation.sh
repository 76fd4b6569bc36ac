# round 5, VERDICT r04 next #5: what would removing the 2^-11 rescale of the weight hi-operand (8 v_pk_mul_f16 per K-step)
# buy at most?  builds: for v in NOSCALE HALFSCALE; do bash scripts/build_variant.sh abl_$v -DHM_EXPERIMENTAL -DHM_ABL_$v; done
for rep in 1 2; do
for v in "" abl_HALFSCALE abl_NOSCALE; do
  if [ -n "$v" ]; then export HORTIHIP_LIB=$PWD/hortimapping_amd/variants/libhortihip_$v.so; else unset HORTIHIP_LIB; fi
  echo "== ${v:-product} (pass $rep)"
  python scripts/gpu_sweep_k1h.py 256 0 2>&1 | grep -v amdgpu.ids | grep "variant 0"
done
done
