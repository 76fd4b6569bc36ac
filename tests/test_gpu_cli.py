"""End-to-end GPU test of the two kept entry points on synthetic data laid out in the reference's folder formats
(BASELINE.json configs[0] "wild_pepper.yaml ... 3 fruit instances" and configs[2] "shape_completion_challenge ...")."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_entry_points_end_to_end(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    data = str(tmp_path / "data")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts/make_synthetic_data.py"), data], env=env)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "test_wild_completion.py"), "-c",
                                   os.path.join(data, "synthetic_wild_pepper.yaml")], env=env, text=True)
    assert "Optimising 3 fruit instances in one batch" in out
    from hortimapping_amd.mesher import read_ply
    for i in (2, 3, 4):                                   # outputs keyed by submap file name (:249-260)
        name = f"{i}_SweetPepper"
        m = read_ply(os.path.join(data, "synthetic_bup20", "submaps_complete", name + ".ply"))
        assert m.faces.shape[0] > 500
        T = np.load(os.path.join(data, "synthetic_bup20", "submaps_pose", name + ".npy"))
        assert T.shape == (4, 4) and 0.5 < np.cbrt(np.linalg.det(T[:3, :3])) < 1.25
        assert abs(T[2, 3] - 0.5) < 0.08                  # fruits sit ~0.5 m in front of the camera
        assert os.path.exists(os.path.join(data, "synthetic_bup20", "submaps_clean", name + ".ply"))
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "run_shape_completion_challenge.py"), "-c",
                                   os.path.join(data, "synthetic_challenge_pepper.yaml")], env=env, text=True)
    vals = {l.split(":")[0].strip(): l.split(":")[1].split()[0] for l in out.splitlines() if ":" in l and "[" in l}
    assert float(vals["CD        [mm]"]) < 5.0 and float(vals["F-score    [%]"]) > 80.0
    assert "calculated over 3 frames" in out
    for i in range(3):
        assert os.path.exists(os.path.join(data, "synthetic_challenge", "results",
                                           "shape_completion_challenge_sweetpepper_homa", "val", f"p{i:03d}.ply"))
