"""Functional mirrors of the reference's decoder helpers (`wild_completion/utils.py`), GPU-backed.

Same names, argument meaning and return shapes as the reference; `decoder` is a `DecoderWeights` bundle (or the
reference's nn.Module, which is converted once).  No CPU fallback: these call libhortihip.so."""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .decoder import DecoderWeights

_CONVERTED = {}


def _module_fingerprint(module):
    """Identity AND content version of an nn.Module's parameters: `id()` alone would hand back stale packed weights
    after an in-place update (optimizer step, load_state_dict) or after the id is reused by a new module."""
    return tuple((k, v.data_ptr(), v._version, tuple(v.shape)) for k, v in module.state_dict(keep_vars=True).items())


def as_weights(decoder) -> DecoderWeights:
    """`decoder`: a DecoderWeights bundle, or the reference's nn.Module (converted once per parameter version)."""
    if isinstance(decoder, DecoderWeights):
        return decoder
    key, fp = id(decoder), _module_fingerprint(decoder)
    hit = _CONVERTED.get(key)
    if hit is None or hit[0] != fp:
        _CONVERTED[key] = (fp, DecoderWeights.from_module(decoder))
    return _CONVERTED[key][1]


def _pack_points(x: torch.Tensor, device):
    x = x.detach().to(device=device, dtype=torch.float32).reshape(-1, 3)
    n = x.shape[0]
    npad = max(64, (n + 63) // 64 * 64)
    pts4 = torch.zeros(1, npad, 4, device=device, dtype=torch.float32)
    pts4[0, :n, :3] = x
    return pts4, n


def decode_sdf(decoder, lat_vec, x, max_batch=64 ** 3):
    """`wild_completion/utils.py:144-172`: sdf values (N,) of query points x (N,3) under latent `lat_vec` (L,)."""
    dec = as_weights(decoder)
    dev = torch.device("cuda")
    pts4, n = _pack_points(x, dev)
    lat = lat_vec.detach().to(dev, torch.float32).reshape(1, -1).contiguous()
    y, _ = ops.decode_batch(dec, lat, pts4, torch.tensor([n], dtype=torch.int32, device=dev), mode=0)
    return y[0, :n]


def get_batch_sdf_jacobian(decoder, lat_vec, x):
    """`wild_completion/utils.py:175-193`: (y (n,1,1), g (n,1,L+3)) with g = d sdf / d [latent ; xyz]."""
    dec = as_weights(decoder)
    dev = torch.device("cuda")
    L = dec.latent_dim
    pts4, n = _pack_points(x, dev)
    lat = lat_vec.detach().to(dev, torch.float32).reshape(1, -1).contiguous()
    y, J = ops.decode_batch(dec, lat, pts4, torch.tensor([n], dtype=torch.int32, device=dev), mode=1, pose_dim=0)
    g = torch.cat([J[0, :n, :L], J[0, :n, L:L + 3]], dim=1)
    return y[0, :n].reshape(n, 1, 1), g.reshape(n, 1, L + 3)


def get_rays(sampled_pixels, invK):
    """`wild_completion/utils.py:23-37`: camera-frame ray directions (z = 1) of pixels [u, v] (host-side data prep)."""
    n = sampled_pixels.shape[0]
    u_hom = np.concatenate([sampled_pixels, np.ones((n, 1))], axis=-1)
    return (u_hom[:, None, :] * invK).sum(-1).astype(np.float32)


class StageTimer:
    """Wall-clock split of an entry-point script, like the reference's get_time stamps (optimizer.py:91-266) but per stage
    of the whole script.  Off unless the environment names a file (HM_STAGE_TIMES=<path.json>): then `lap(name)`
    synchronises the device, adds the time since the previous lap to `name`, and `write(**extra)` saves the JSON record.
    Off = no synchronisation, no cost (scripts/e2e_cli_timing.py is the user)."""

    def __init__(self):
        import os
        import time
        self.path = os.environ.get("HM_STAGE_TIMES", "")
        self.stages = {}
        self._t = time.perf_counter()
        self._t0 = self._t

    def lap(self, name):
        if not self.path:
            return
        import time
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        now = time.perf_counter()
        self.stages[name] = self.stages.get(name, 0.0) + (now - self._t)
        self._t = now

    def write(self, **extra):
        if not self.path:
            return
        import json
        import time
        rec = {"stages_s": {k: round(v, 4) for k, v in self.stages.items()},
               "total_s": round(time.perf_counter() - self._t0, 4), **extra}
        with open(self.path, "w") as f:
            json.dump(rec, f)
