"""Host (numpy / scikit-learn mirrors of the reference) vs device (csrc/hm_prep.hip) timing of the per-instance data
preparation at BUP20-like sizes: 30 fruit instances, 50 frames of 720 x 1280, 2000 surface samples per fruit, a
485k-point background cloud.  GPU box:  python scripts/bench_data_prep.py > gpurun_out/r02_data_prep.txt"""
import os
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hortimapping_amd import data_prep as DP      # noqa: E402

cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "wild_pepper.yaml")))
H, W, F, NI = 720, 1280, 50, 30
rs = np.random.RandomState(0)
yy, xx = np.mgrid[0:H, 0:W]
ids, dep, pose = {}, {}, {}
for f in range(F):
    img = np.zeros((H, W), np.int32)
    for k in range(NI):
        cy, cx = rs.randint(60, H - 60), rs.randint(60, W - 60)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < 40 ** 2] = k + 2
    ids[f], dep[f], pose[f] = img, (0.5 + 0.1 * rs.rand(H, W)).astype(np.float32), np.eye(4)
invK = np.linalg.inv(np.array([[600.0, 0, 640], [0, 600, 360], [0, 0, 1]]))
sids = [k + 2 for k in range(NI)]


def timed(fn, sync=False):
    if sync:
        torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    if sync:
        torch.cuda.synchronize()
    return r, time.perf_counter() - t


np.random.seed(42)
host_rd, t_h = timed(lambda: [DP.get_render_data(s, ids, dep, pose, (H, W), invK, cfg) for s in sids])
frames, t_up = timed(lambda: DP.DeviceFrames(ids, dep), True)
np.random.seed(42)
DP.get_render_data_device(sids[:2], frames, pose, (H, W), invK, cfg)           # warm-up
np.random.seed(42)
dev_rd, t_d = timed(lambda: DP.get_render_data_device(sids, frames, pose, (H, W), invK, cfg), True)
assert all(a["count"] == b["count"] and all(torch.equal(x, y) for x, y in zip(a["rays_fg"], b["rays_fg"]))
           for a, b in zip(dev_rd, host_rd))
print(f"get_render_data, {NI} instances x {F} frames of {H}x{W}: host {t_h * 1e3 / NI:8.1f} ms / instance   device "
      f"{t_d * 1e3 / NI:6.2f} ms / instance (+ {t_up * 1e3:.0f} ms once to upload the sequence)   x{t_h / t_d:.0f}")

# the one full pass over the sequence (hm_prep_stats): HBM-bound, id + depth images read once
import ctypes
lib = DP._prep_lib()
lut = torch.full((NI + 3,), -1, dtype=torch.int32)
lut[2:NI + 2] = torch.arange(NI, dtype=torch.int32)
d_lut = lut.cuda()
d_stats = torch.tensor([0, 2 ** 31 - 1, -1, 2 ** 31 - 1, -1], dtype=torch.int32).repeat(NI * F).cuda()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    lib.hm_prep_stats(frames.ids.data_ptr(), frames.depth.data_ptr(), F, H, W, d_lut.data_ptr(), NI + 3, NI, d_stats.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
nbytes = F * H * W * 8
print(f"hm_prep_stats: {nbytes / 1e6:.0f} MB of id + depth images in {ms:.3f} ms = {nbytes / ms / 1e6:.0f} GB/s "
      f"({nbytes / ms / 1e6 / 8000 * 100:.0f} % of the 8 TB/s HBM peak)")

clouds = []
for k in range(NI):
    d = rs.randn(6000, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d[d[:, 2] < -0.2][:1850]
    clouds.append(np.concatenate([0.04 * d + rs.uniform(-0.3, 0.3, 3), 0.004 * rs.randn(100, 3) + [0.07, 0, 0],
                                  rs.uniform(-0.1, 0.1, (50, 3))]))
host_c, t_h = timed(lambda: [DP.clean_pcd(c, 0.01) for c in clouds])
DP.clean_pcd_device(clouds[:2], 0.01)
dev_c, t_d = timed(lambda: DP.clean_pcd_device(clouds, 0.01), True)
assert all(np.array_equal(a, b) for a, b in zip(host_c, dev_c))
print(f"clean_pcd (DBSCAN), {NI} clouds of 2000 points:          host {t_h * 1e3 / NI:8.1f} ms / instance   device "
      f"{t_d * 1e3 / NI:6.2f} ms / instance   x{t_h / t_d:.0f}")

bg = rs.uniform(-0.6, 0.6, (485000, 3))
host_p, t_h = timed(lambda: [DP.get_pose_init(c, bg) for c in host_c])
cloud, t_up = timed(lambda: DP.DeviceCloud(bg), True)


def dev_pose():
    boxes = [DP.pose_init_box(c) for c in dev_c]
    ok = [i for i, b in enumerate(boxes) if b[2]]
    crops = dict(zip(ok, cloud.crop_boxes([boxes[i][3] for i in ok], [boxes[i][4] for i in ok])))
    return [(b[0], DP.pose_init_rotation(b[0], crops.get(i)) if b[2] else 0.0, b[1], b[2]) for i, b in enumerate(boxes)]
dev_pose()
dev_p, t_d = timed(dev_pose, True)
assert all(np.array_equal(a[0], b[0]) and a[1:] == b[1:] for a, b in zip(host_p, dev_p))
print(f"get_pose_init, 485k-point background cloud:             host {t_h * 1e3 / NI:8.1f} ms / instance   device "
      f"{t_d * 1e3 / NI:6.2f} ms / instance (+ {t_up * 1e3:.0f} ms once to upload the cloud)   x{t_h / t_d:.0f}")
print("every device result above was compared with the host result: identical")
