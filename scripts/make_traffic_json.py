#!/usr/bin/env python3
"""Derive profiles/rNN_traffic.json (the per-launch fabric traffic and matrix-pipe-busy figures bench.py quotes) from the
condensed PMC summaries scripts/collect_profiles.sh writes, so the quoted figure cannot go stale:

    python scripts/make_traffic_json.py <dir with <tag>_pmc_*.txt> <tag> <out.json> [kernel substring] [precision]

FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads: MI355X_MICROARCH.md, HBM section);
WRITE_SIZE is taken as is (uncalibrated); both are kilobytes per dispatch.  Busy fraction = SQ_VALU_MFMA_BUSY_CYCLES summed
over the chip's 1024 SIMDs / (GRBM_GUI_ACTIVE summed over 8 XCDs / 8)."""
import json
import os
import sys


def rows(path, kernel):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        p = line.rstrip("\n").rsplit(",", 4)
        if len(p) == 5 and kernel in p[0]:
            try:
                out[p[1]] = (int(p[2]), float(p[3]))
            except ValueError:
                pass
    return out


def main():
    d, tag, outp = sys.argv[1], sys.argv[2], sys.argv[3]
    kernel = sys.argv[4] if len(sys.argv) > 4 else "k_decoder_h<0, false>"
    prec = sys.argv[5] if len(sys.argv) > 5 else "f16x3"
    f = rows(os.path.join(d, f"{tag}_pmc_FETCH_SIZE.txt"), kernel)
    w = rows(os.path.join(d, f"{tag}_pmc_WRITE_SIZE.txt"), kernel)
    b = rows(os.path.join(d, f"{tag}_pmc_SQ_VALU_MFMA_BUSY_CYCLES.txt"), kernel)
    t = rows(os.path.join(d, f"{tag}_pmc_TCC_HIT_sum.txt"), kernel)
    e = {"kernel": kernel, "workload": "c2_joint main launch (65,536 forward+backward queries + the ball-valid ray samples' forward tiles)"}
    if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
        e.update(fetch_size_kb_per_launch=round(f["FETCH_SIZE"][1]), fetch_correction=2.0,
                 write_size_kb_per_launch=round(w["WRITE_SIZE"][1]), launches=f["FETCH_SIZE"][0],
                 bytes_per_launch=int(round((2.0 * f["FETCH_SIZE"][1] + w["WRITE_SIZE"][1]) * 1024)))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in b and "GRBM_GUI_ACTIVE" in b:
        busy = b["SQ_VALU_MFMA_BUSY_CYCLES"][1] / 1024.0
        act = b["GRBM_GUI_ACTIVE"][1] / 8.0
        e.update(mfma_pipe_busy_frac=round(busy / act, 4), mfma_busy_cycles_per_simd=round(busy), active_cycles_per_launch=round(act))
    if "TCC_HIT_sum" in t and "TCC_MISS_sum" in t:
        e["l2_hit_rate"] = round(t["TCC_HIT_sum"][1] / (t["TCC_HIT_sum"][1] + t["TCC_MISS_sum"][1]), 4)
    e["source"] = (f"rocprofv3 --pmc passes of `python bench.py --steps 2` ({tag}_pmc_*.txt under profiles/), one counter group per "
                   "pass; FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, L2 <-> fabric bytes")
    data = {}
    if os.path.exists(outp):
        data = json.load(open(outp))
    data[prec] = e
    json.dump(data, open(outp, "w"), indent=1)
    print(json.dumps(e))


if __name__ == "__main__":
    main()
