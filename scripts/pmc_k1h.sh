#!/bin/bash
# pmc_k1h.sh: SQ / TA / TCP / TCC counter passes for the iteration's main K1h launch (one group per pass; --pmc is never
# combined with other trace domains).  Every pass runs under its own `timeout`: the TA_* group of round 3 hangs rocprofv3
# on this pool in round 4 (it cost one 40-minute gpurun call) and is left out.  Output: gpurun_out/r04_k1h_pmc_deep.txt
export TMPDIR=/tmp
out=gpurun_out; mkdir -p $out
CMD="python bench.py --steps 1 --warmup 1 --iters 40 --groups 1 --no-cpu-baseline --no-exact $*"
res=$out/r04_k1h_pmc_deep.txt; : > $res
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "TCC_REQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1)); d=/tmp/pmc_deep/$i; rm -rf $d; mkdir -p $d
  timeout 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o pmc -- $CMD > $d/log 2>&1 || echo "pass $i ($grp): rocprofv3 exit $? (timeout 240 s)" >> $res
  f=$(find $d -name "*_counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pass $i ($grp): no counter file; tail of log:" >> $res; tail -3 $d/log >> $res; continue; fi
  python - "$f" >> $res <<'PY'
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Kernel_Name", "")
    if "k_decoder_h" not in n: continue
    key = (("main<0>" if "<0, false>" in n else "render<1>" if "<1, false>" in n else "other"), r["Counter_Name"])
    agg[key][0] += 1; agg[key][1] += float(r["Counter_Value"] or 0)
for (k, c), (n, s) in sorted(agg.items()):
    print(f"{k:10s} {c:40s} dispatches {n:5d}  mean {s / n:.6g}")
PY
done
cat $res
