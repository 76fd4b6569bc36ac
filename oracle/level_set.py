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


# ---- marching-cubes base cases --------------------------------------------------------------------------------------
# The reference's mesh comes from scikit-image's marching_cubes (utils.py:573), whose default method is Lewiner's MC33.
# Classic tables and Lewiner's agree on the VERTEX SET (the edge crossings above) except in the sub-cases where Lewiner
# resolves a face / body ambiguity with an extra vertex at the cell centre (6.1.2, 7.3, 10.2, 12.2, 13.2-13.4).  Those
# sub-cases only occur in cells whose sign configuration belongs to one of the ambiguous base cases, so counting the
# cells of every base case bounds where the two vertex sets can differ.  Numbering: Chernyaev / Lewiner (cases 11 and 14
# are mirror images and share a signature here).
AMBIGUOUS_CASES = (3, 4, 6, 7, 10, 12, 13)          # face-ambiguous: 3, 6, 7, 10, 12, 13; body-ambiguous: 4, 6, 7, 10, 12, 13
CENTRE_VERTEX_CASES = (6, 7, 10, 12, 13)            # base cases with a Lewiner sub-case that adds the centre vertex


def mc_case_table() -> np.ndarray:
    """(256,) base case of every corner configuration (bit c set <=> corner c inside; corner c = (c&1, c>>1&1, c>>2&1)),
    from rotation / reflection / complement invariants: number of inside corners n (complemented when > 4) and the
    numbers (e, f, b) of inside pairs at Hamming distance 1 / 2 / 3 (cube edges, face diagonals, body diagonals)."""
    sig = {(0, 0, 0, 0): 0, (1, 0, 0, 0): 1, (2, 1, 0, 0): 2, (2, 0, 1, 0): 3, (2, 0, 0, 1): 4, (3, 2, 1, 0): 5,
           (3, 1, 1, 1): 6, (3, 0, 3, 0): 7, (4, 4, 2, 0): 8, (4, 3, 3, 0): 9, (4, 2, 2, 2): 10, (4, 3, 2, 1): 11,
           (4, 2, 3, 1): 12, (4, 0, 6, 0): 13}
    out = np.zeros(256, np.int32)
    for cfg in range(256):
        c = cfg if bin(cfg).count("1") <= 4 else (~cfg) & 255
        idx = [i for i in range(8) if (c >> i) & 1]
        d = [bin(a ^ b).count("1") for k, a in enumerate(idx) for b in idx[k + 1:]]
        out[cfg] = sig[(len(idx), d.count(1), d.count(2), d.count(3))]
    return out


def mc_case_histogram(vol: np.ndarray, level: float = 0.0) -> np.ndarray:
    """(15,) number of cells of the n^3 grid `vol[ix, iy, iz]` per base case (index 14 unused: merged with 11).  A corner
    counts as inside when its value is below the level -- the convention of `edge_crossings`."""
    ins = (np.asarray(vol) < level)
    cfg = np.zeros(tuple(s - 1 for s in ins.shape), np.int32)
    for c in range(8):
        dx, dy, dz = c & 1, (c >> 1) & 1, (c >> 2) & 1
        cfg |= ins[dx:ins.shape[0] - 1 + dx, dy:ins.shape[1] - 1 + dy, dz:ins.shape[2] - 1 + dz].astype(np.int32) << c
    return np.bincount(mc_case_table()[cfg].ravel(), minlength=15)
