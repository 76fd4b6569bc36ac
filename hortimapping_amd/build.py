"""Build libhortihip.so (gfx950 only) in-tree with hipcc.  No torch extension machinery: the library is a
plain C-ABI shared object loaded through ctypes (`hortimapping_amd._lib`).

Each translation unit is compiled to `build/obj/<name>.o` (in parallel, only when it or a header changed) and the
objects are linked into `hortimapping_amd/libhortihip.so`; `build/` is scratch (git- and gpurun-ignored), the `.so`
travels to the GPU box with the tree."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "..", "build", "obj")
LIB = os.path.join(HERE, "libhortihip.so")
SOURCES = ["hm_pack.hip", "hm_decoder.hip", "hm_decoder_h.hip", "hm_decoder_p.hip", "hm_decoder_any.hip", "hm_normal_eq.hip", "hm_solve.hip", "hm_render.hip",
           "hm_optimize.hip", "hm_mesh.hip", "hm_metrics.hip", "hm_prep.hip", "hm_debug.hip", "hm_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", CSRC]


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hs += [os.path.join(HERE, "..", "include", "hortimapping_amd.h"), os.path.abspath(__file__)]
    return max(os.path.getmtime(h) for h in hs if os.path.exists(h))


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _stale_objects(force):
    ht = _headers_mtime()
    out = []
    for s in SOURCES:
        o, p = _obj(s), os.path.join(CSRC, s)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(p), ht):
            out.append(s)
    return out


def build(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    todo = _stale_objects(force)
    if not todo and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(_obj(s)) for s in SOURCES):
        return LIB

    def compile_one(s):
        cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, s), "-o", _obj(s)]
        if verbose:
            print("[hortimapping_amd] " + " ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + [_obj(s) for s in SOURCES]
    if verbose:
        print("[hortimapping_amd] " + " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
