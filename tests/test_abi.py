"""CPU tests of the drop-in boundary: the shared library loads, exports every symbol include/*.h declares, the ctypes
structures mirror the C structs byte for byte, and the product path fails loudly without the library."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hortimapping_amd.h")
DEBUG_HEADER = os.path.join(ROOT, "include", "hortimapping_amd_debug.h")


def declared_functions(headers=(HEADER, DEBUG_HEADER)):
    out = set()
    for h in headers:
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        out |= set(re.findall(r"\b(hm_[a-z0-9_]+)\s*\(", src))
    return sorted(out)


def test_the_product_header_carries_no_debug_hooks():
    """Test / A-B / trace hooks live in include/hortimapping_amd_debug.h; the drop-in ABI declares none of them."""
    assert not [f for f in declared_functions((HEADER,)) if f.startswith("hm_debug") or f == "hm_workspace_set_debug"]
    assert "hm_workspace_set_debug" in declared_functions((DEBUG_HEADER,))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from hortimapping_amd import _lib
    lib = _lib.lib()
    fns = declared_functions()
    assert len(fns) >= 12
    for f in fns:
        assert hasattr(lib, f), f"{f} declared in include/hortimapping_amd.h but not exported"
    assert lib.hm_last_error() is not None


def test_ctypes_structs_match_c_layout():
    """Compile a tiny C program against the public header and compare sizeof/offsetof with the ctypes mirrors."""
    from hortimapping_amd import decoder as HD, optimizer as HO
    fields = {
        "hm_decoder_arch": (HD.HmDecoderArch, ["latent_dim", "use_tanh", "in_dim", "out_dim", "cat", "layer_norm"]),
        "hm_opt_cfg": (HO.HmOptCfg, ["scale_on", "lm_lambda_0", "n_sample_on_ray", "w_codereg", "max_iter",
                                     "epsilon_s", "min_grad_thre"]),
        "hm_limits": (HO.HmLimits, ["max_batch", "max_grad_samples"]),
        "hm_batch": (HO.HmBatch, ["B", "d_points_w", "d_n_frames", "d_pose_known", "d_latent", "d_status"]),
        "hm_debug": (HO.HmDebug, ["d_A", "d_counts"]),
    }
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void){']
    for name, (_, fl) in fields.items():
        prog.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f in fl:
            prog.append(f'printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    prog.append('return 0;}')
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write("\n".join(prog))
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-std=c99", "-o", exe, c])
        out = subprocess.check_output([exe]).decode().split("\n")
    got = dict(l.split() for l in out if l.strip())
    for name, (cls, fl) in fields.items():
        assert int(got[name]) == ctypes.sizeof(cls), name
        for f in fl:
            assert int(got[f"{name}.{f}"]) == getattr(cls, f).offset, f"{name}.{f}"


def test_missing_library_fails_loudly(tmp_path):
    """No CPU fallback: without libhortihip.so every compute entry point raises."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from hortimapping_amd import _lib\n"
        "_lib.LIB_PATH = %r\n"
        "from hortimapping_amd import synthetic as S\n"
        "from hortimapping_amd.decoder import DecoderWeights\n"
        "try:\n"
        "    DecoderWeights.from_params(S.make_synthetic_decoder(32))\n"
        "except _lib.HortiHipError as e:\n"
        "    print('RAISED', e); sys.exit(0)\n"
        "sys.exit(1)\n" % (ROOT, str(tmp_path / "nope.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "RAISED" in out.stdout and "no CPU fallback" in out.stdout.lower() or "There is no CPU fallback" in out.stdout


def test_bad_arguments_return_error_codes():
    """Error convention of the C ABI: negative return + hm_last_error(), never a crash (no GPU needed for these)."""
    from hortimapping_amd import _lib
    lib = _lib.lib()
    h = ctypes.c_void_p()
    rc = lib.hm_decoder_create(48, None, None, ctypes.byref(h))      # null weights
    assert rc < 0 and b"null" in lib.hm_last_error()
    arr = (ctypes.POINTER(ctypes.c_float) * 9)()
    rc = lib.hm_decoder_create(48, arr, arr, ctypes.byref(h))        # latent dim not a multiple of 32
    assert rc < 0 and b"multiple of 32" in lib.hm_last_error()
    assert lib.hm_decoder_destroy(None) == 0
    assert lib.hm_decoder_latent_dim(None) == -1
