"""Build libhortihip.so (gfx950 only) in-tree with hipcc.  No torch extension machinery: the library is a
plain C-ABI shared object loaded through ctypes (`hortimapping_amd._lib`)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhortihip.so")
SOURCES = ["hm_pack.hip", "hm_decoder.hip", "hm_decoder_h.hip", "hm_normal_eq.hip", "hm_solve.hip", "hm_render.hip",
           "hm_optimize.hip", "hm_mesh.hip", "hm_metrics.hip", "hm_api.hip"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", CSRC,
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[hortimapping_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
