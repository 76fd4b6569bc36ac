#!/bin/bash
# Matrix-pipe-busy counters of the any-architecture decoder kernels (GPU box, through gpurun): one --pmc pass, condensed to
# gpurun_out/r05_arch_pmc_mfma.txt.  Busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs.
export TMPDIR=/tmp
d=/tmp/prof_arch_pmc; rm -rf $d; mkdir -p $d gpurun_out
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $d -o pmc -- python scripts/pmc_arch_decoder.py > gpurun_out/arch_pmc.log 2>&1
for k in kernel_trace counter_collection; do f=$(find $d -name "*_$k.csv" | head -1); [ -n "$f" ] && mv "$f" $d/pmc_$k.csv; done
python scripts/summarize_prof.py $d pmc gpurun_out/r05_arch_pmc_mfma.txt
python - <<'PY'
import re
rows = {}
for line in open("gpurun_out/r05_arch_pmc_mfma.txt"):
    p = line.rstrip("\n").rsplit(",", 4)
    if len(p) == 5 and "k_decoder" in p[0]:
        try: rows.setdefault(p[0], {})[p[1]] = float(p[3])
        except ValueError: pass
with open("gpurun_out/r05_arch_pmc_mfma.txt", "a") as f:
    f.write("== matrix pipe busy per launch (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs) ==\n")
    for k, v in sorted(rows.items()):
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE"):
            s = "%-40s %.3f" % (k, (v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (v["GRBM_GUI_ACTIVE"] / 8.0))
            f.write(s + "\n"); print(s)
PY
