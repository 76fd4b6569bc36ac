#!/usr/bin/env python3
"""Diagnostic: tests/test_gpu_configs.py::test_frame_turns_invalid_mid_trajectory_L256 in f16x3 with the normal equations of
the f16x3 arithmetics on K4h (default, DIAG_K4=1) or on the fp32-input kernel (DIAG_K4=0).  Prints PASS / the assertion."""
import os
import sys
import ctypes
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["HM_PRECISION"] = "f16x3"
from hortimapping_amd import _lib, optimizer as HO     # noqa: E402

k4 = int(os.environ.get("DIAG_K4", "1"))
_orig = HO.Workspace.__init__


def _init(self, *a, **kw):
    _orig(self, *a, **kw)
    lib = _lib.lib()
    lib.hm_workspace_set_k4_split.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.hm_workspace_set_k4_split(self.handle, k4)


HO.Workspace.__init__ = _init
import test_gpu_configs as TC                           # noqa: E402
try:
    TC.test_frame_turns_invalid_mid_trajectory_L256("f16x3")
    print("DIAG_K4=%d: PASS" % k4)
except AssertionError:
    print("DIAG_K4=%d: FAIL" % k4)
    traceback.print_exc(limit=2)
