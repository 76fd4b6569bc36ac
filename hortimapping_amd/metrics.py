"""Parity metrics of BASELINE.json: Chamfer distance (as the reference defines it) and pose error.

`chamfer_distance` restates `metrics_3d/chamfer_distance.py:16-26`: mean unsquared nearest-neighbour distance in
both directions, halved.  Point sets are drawn from completed shapes with ONE sampler for every party (GPU result,
oracle result, ground truth): the zero level set of the decoder along fixed Fibonacci directions from the object
origin (the synthetic fruits are star-shaped), mapped to the world by T_wo = inverse(T_ow)."""
from __future__ import annotations

import math

import numpy as np
import torch


def fibonacci_dirs(n: int) -> np.ndarray:
    i = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * i / n)
    theta = math.pi * (1 + 5 ** 0.5) * i
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)


def sample_level_set(sdf_fn, n_dirs: int = 2000, r_max: float = 0.08, n_bisect: int = 24) -> np.ndarray:
    """Object-frame points with sdf = 0 along `n_dirs` rays from the origin (first crossing by bisection)."""
    d = fibonacci_dirs(n_dirs)
    lo = np.zeros(n_dirs)
    hi = np.full(n_dirs, r_max)
    for _ in range(n_bisect):
        mid = 0.5 * (lo + hi)
        v = np.asarray(sdf_fn(d * mid[:, None]), dtype=np.float64)
        inside = v < 0
        lo = np.where(inside, mid, lo)
        hi = np.where(inside, hi, mid)
    return d * (0.5 * (lo + hi))[:, None]


def completed_points_world(sdf_fn, T_ow: np.ndarray, n_dirs: int = 2000, r_max: float = 0.08) -> np.ndarray:
    """Level-set samples mapped to the world frame (the reference writes the completed mesh transformed by T_wo,
    test_wild_completion.py:249-252)."""
    p_o = sample_level_set(sdf_fn, n_dirs, r_max)
    T_wo = np.linalg.inv(np.asarray(T_ow, dtype=np.float64))
    return p_o @ T_wo[:3, :3].T + T_wo[:3, 3]


def chamfer_distance(A: np.ndarray, B: np.ndarray) -> float:
    from scipy.spatial import cKDTree
    da = cKDTree(B).query(A)[0]
    db = cKDTree(A).query(B)[0]
    return 0.5 * (float(da.mean()) + float(db.mean()))


def pose_error(T_ow: np.ndarray, T_wo_true: np.ndarray):
    """(translation error [m], rotation error [deg], scale ratio) of inverse(T_ow) against the true T_wo."""
    T_wo = np.linalg.inv(np.asarray(T_ow, dtype=np.float64))
    Tt = np.asarray(T_wo_true, dtype=np.float64)
    s = np.cbrt(np.linalg.det(T_wo[:3, :3]))
    st = np.cbrt(np.linalg.det(Tt[:3, :3]))
    R = T_wo[:3, :3] / s
    Rt = Tt[:3, :3] / st
    c = (np.trace(R @ Rt.T) - 1) / 2
    return (float(np.linalg.norm(T_wo[:3, 3] - Tt[:3, 3])), float(np.degrees(np.arccos(np.clip(c, -1, 1)))),
            float(s / st))
