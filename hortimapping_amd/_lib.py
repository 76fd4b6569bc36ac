"""ctypes binding of libhortihip.so -- the only way the Python host reaches the GPU kernels.

There is deliberately NO fallback: if the shared library is missing or fails to load, importing any
compute entry point raises (the product path must fail loudly without its HIP extension)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HORTIHIP_LIB: development aid for same-box A/B timing of kernel variants (scripts/build_variant.sh builds them under
# build/); unset = the in-tree library.  Either way a missing file is an error, never a fallback.
LIB_PATH = os.environ.get("HORTIHIP_LIB") or os.path.join(_HERE, "libhortihip.so")
_lib = None

c_float_p = ctypes.POINTER(ctypes.c_float)
c_void_p = ctypes.c_void_p
c_int = ctypes.c_int


class HortiHipError(RuntimeError):
    pass


def _declare(lib):
    lib.hm_last_error.restype = ctypes.c_char_p
    lib.hm_last_error.argtypes = []
    lib.hm_decoder_create.restype = c_int
    lib.hm_decoder_create.argtypes = [c_int, ctypes.POINTER(c_float_p), ctypes.POINTER(c_float_p),
                                      ctypes.POINTER(c_void_p)]
    lib.hm_decoder_create_arch.restype = c_int
    lib.hm_decoder_create_arch.argtypes = [c_void_p, ctypes.POINTER(c_float_p), ctypes.POINTER(c_float_p),
                                           ctypes.POINTER(c_float_p), ctypes.POINTER(c_float_p),
                                           ctypes.POINTER(c_void_p)]
    lib.hm_decoder_destroy.restype = c_int
    lib.hm_decoder_destroy.argtypes = [c_void_p]
    lib.hm_decoder_latent_dim.restype = c_int
    lib.hm_decoder_latent_dim.argtypes = [c_void_p]
    lib.hm_decoder_set_precision.restype = c_int
    lib.hm_decoder_set_precision.argtypes = [c_void_p, c_int]
    lib.hm_decoder_get_precision.restype = c_int
    lib.hm_decoder_get_precision.argtypes = [c_void_p]
    lib.hm_decode_batch.restype = c_int
    lib.hm_decode_batch.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                    c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HortiHipError(
                f"{LIB_PATH} not found: build it with `python -m hortimapping_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # torch bundles its own libamdhip64.so.7 / libhsa-runtime64; load it FIRST so that libhortihip.so binds to the
        # same HIP runtime instance (two runtimes in one process cannot share device pointers or streams)
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().hm_last_error().decode("utf-8", "replace")
        raise HortiHipError(f"{what} failed (rc={rc}): {msg}")
