"""The round-4 schedule experiments of the f16x3 decoder kernel (hortimapping_amd/csrc/experimental/) are not part of the
product library; when an experimental build is present (scripts/build_variant.sh exp -DHM_EXPERIMENTAL) this test checks,
in a subprocess bound to that build, that every experimental schedule reproduces the product kernel's bits."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "hortimapping_amd", "variants", "libhortihip_exp.so")


@pytest.mark.skipif(not os.path.exists(EXP), reason="no experimental build (scripts/build_variant.sh exp -DHM_EXPERIMENTAL)")
def test_experimental_schedules_reproduce_the_product_bits():
    env = dict(os.environ, HORTIHIP_LIB=EXP)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_k1g_bits.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-2000:]
