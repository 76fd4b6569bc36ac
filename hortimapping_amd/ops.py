"""Thin torch-tensor wrappers over the C ABI (device pointers + current stream; no torch types cross the ABI)."""
from __future__ import annotations

import torch

from . import _lib
from .decoder import DecoderWeights

TQ = 64
POSE_PAD = 8


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def decode_batch(dec: DecoderWeights, latent: torch.Tensor, pts4: torch.Tensor, n_q: torch.Tensor,
                 mode: int = 0, pose_dim: int = 0):
    """latent (B,L) f32 cuda; pts4 (B,n_stride,4) f32 cuda (object frame); n_q (B,) int32 cuda.
    Returns y (B,n_stride) and, for mode 1, J (B,n_stride,L+8) rows [d/dz | pose-or-xyz | pad]."""
    assert latent.is_cuda and latent.dtype == torch.float32 and latent.is_contiguous()
    assert pts4.is_cuda and pts4.dtype == torch.float32 and pts4.is_contiguous() and pts4.shape[-1] == 4
    assert n_q.is_cuda and n_q.dtype == torch.int32
    B, n_stride = pts4.shape[0], pts4.shape[1]
    L = dec.latent_dim
    assert n_stride % TQ == 0 and latent.shape == (B, L)
    cb = torch.empty(2, B, 512, device=latent.device, dtype=torch.float32)
    y = torch.zeros(B, n_stride, device=latent.device, dtype=torch.float32)
    ldJ = L + POSE_PAD
    J = torch.zeros(B, n_stride, ldJ, device=latent.device, dtype=torch.float32) if mode == 1 else None
    rc = _lib.lib().hm_decode_batch(dec.handle, B, _ptr(latent), L, _ptr(pts4), _ptr(n_q), n_stride, _ptr(cb),
                                    _ptr(y), _ptr(J), ldJ, pose_dim, mode, _stream())
    _lib.check(rc, "hm_decode_batch")
    return y, J
