"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (same rules as hm_oracle.py: imported by tests/ only).

Independent check of the GPU iso-surface extraction (`hm_extract_surface`, marching tetrahedra) against what the
reference's `convert_sdf_voxels_to_mesh` (wild_completion/utils.py:565-588: skimage `marching_cubes(volume, level=0,
spacing=voxel_size)` + origin shift -1 + scale by cube_radius) produces.  scikit-image is not in this image, and a
triangle-for-triangle comparison would be between two DIFFERENT valid triangulations anyway; what every marching-cubes
variant (classic Lorensen, skimage's Lewiner, marching tetrahedra) agrees on is

  * the vertices on the grid's axis-aligned edges: one per edge whose end values straddle the level, at the LINEAR
    interpolation point  t = (level - a) / (b - a)  -- `edge_crossings` below; skimage's mesh has exactly this vertex
    set (Lewiner adds no other vertices), marching tetrahedra has it plus vertices on face / body diagonals;
  * the surface those vertices span, up to the cell size h: `chamfer_to_crossings` measures the mesh against the
    crossing cloud in both directions.
"""
import numpy as np


def edge_crossings(vol: np.ndarray, level: float = 0.0, cube_radius: float = 1.0) -> np.ndarray:
    """(M, 3) points where the level set crosses the axis-aligned edges of the regular n^3 grid `vol[ix, iy, iz]`, in the
    coordinates convert_sdf_voxels_to_mesh returns: (-1 + index * 2 / (n - 1)) * cube_radius (utils.py:573-586)."""
    vol = np.asarray(vol, dtype=np.float64)
    n = vol.shape[0]
    h = 2.0 / (n - 1)
    idx = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), axis=-1).astype(np.float64)
    pts = []
    for ax in range(3):
        sl_a = [slice(None)] * 3
        sl_b = [slice(None)] * 3
        sl_a[ax], sl_b[ax] = slice(0, n - 1), slice(1, n)
        a, b = vol[tuple(sl_a)], vol[tuple(sl_b)]
        cross = ((a < level) & (b >= level)) | ((a >= level) & (b < level))      # one end strictly below the level
        t = (level - a[cross]) / (b[cross] - a[cross])
        p = idx[tuple(sl_a)][cross].copy()
        p[:, ax] += t
        pts.append(p)
    p = np.concatenate(pts, axis=0)
    return (-1.0 + p * h) * cube_radius


def chamfer_to_crossings(mesh_points: np.ndarray, crossings: np.ndarray):
    """(mean, max) nearest-neighbour distance mesh samples -> crossings, and crossings -> mesh samples."""
    from scipy.spatial import cKDTree
    d_mc = cKDTree(crossings).query(mesh_points)[0]
    d_cm = cKDTree(mesh_points).query(crossings)[0]
    return (float(d_mc.mean()), float(d_mc.max())), (float(d_cm.mean()), float(d_cm.max()))
