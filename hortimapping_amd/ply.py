"""Triangle meshes on the host and their PLY io (numpy only: the dataset readers of the two entry points import this
without pulling in torch, so that file reading can start before the heavy imports -- run_shape_completion_challenge.py).
`hortimapping_amd.mesher` re-exports everything here."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class TriangleMesh:
    vertices: np.ndarray            # (V, 3) float32
    faces: np.ndarray               # (F, 3) int32
    color: Optional[np.ndarray] = None

    def transform(self, T) -> "TriangleMesh":
        T = np.asarray(T, dtype=np.float64)
        v = self.vertices.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
        return TriangleMesh(v.astype(np.float32), self.faces, self.color)

    def area(self) -> float:
        a, b, c = (self.vertices[self.faces[:, k]].astype(np.float64) for k in range(3))
        return float(0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum())

    def sample_points_uniformly(self, n: int, seed: int = 0) -> np.ndarray:
        """Area-weighted uniform samples (what Metrics3D.convert_to_pcd asks Open3D for, metrics_3d/metric.py:41)."""
        rs = np.random.RandomState(seed)
        a, b, c = (self.vertices[self.faces[:, k]].astype(np.float64) for k in range(3))
        w = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
        f = rs.choice(len(w), size=n, p=w / w.sum())
        r1, r2 = np.sqrt(rs.rand(n)), rs.rand(n)
        return (1 - r1)[:, None] * a[f] + (r1 * (1 - r2))[:, None] * b[f] + (r1 * r2)[:, None] * c[f]


def write_ply(mesh: TriangleMesh, path: str):
    """Binary little-endian PLY with the element layout of `write_mesh_to_ply` (utils.py:591-611)."""
    v = np.ascontiguousarray(mesh.vertices, dtype="<f4")
    f = np.ascontiguousarray(mesh.faces, dtype="<i4")
    header = ("ply\nformat binary_little_endian 1.0\n"
              f"element vertex {v.shape[0]}\nproperty float x\nproperty float y\nproperty float z\n"
              f"element face {f.shape[0]}\nproperty list uchar int vertex_indices\nend_header\n")
    rec = np.zeros(f.shape[0], dtype=[("n", "u1"), ("idx", "<i4", (3,))])
    rec["n"] = 3
    rec["idx"] = f
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(v.tobytes())
        fh.write(rec.tobytes())


def read_ply(path: str) -> TriangleMesh:
    """Minimal PLY reader (ascii or binary little-endian; vertex x,y,z [+ extra float/uchar props], triangle faces)."""
    with open(path, "rb") as fh:
        data = fh.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    lines = data[:end].decode("ascii", "replace").split("\n")
    fmt = [l.split()[1] for l in lines if l.startswith("format")][0]
    elems, cur = [], None
    for l in lines:
        t = l.split()
        if not t:
            continue
        if t[0] == "element":
            cur = {"name": t[1], "count": int(t[2]), "props": []}
            elems.append(cur)
        elif t[0] == "property" and cur is not None:
            cur["props"].append(t[1:])
    np_t = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
            "char": "i1", "int": "<i4", "int32": "<i4", "uint": "<u4", "short": "<i2", "ushort": "<u2"}
    verts, faces = None, np.zeros((0, 3), np.int32)
    if fmt == "ascii":
        body = data[end:].decode("ascii").split("\n")
        pos = 0
        for e in elems:
            rows = [body[pos + i].split() for i in range(e["count"])]
            pos += e["count"]
            if e["name"] == "vertex":
                names = [p[-1] for p in e["props"]]
                arr = np.array(rows, dtype=np.float64)
                verts = arr[:, [names.index("x"), names.index("y"), names.index("z")]].astype(np.float32)
            elif e["name"] == "face" and e["count"]:
                faces = np.array([[int(r[1]), int(r[2]), int(r[3])] for r in rows], dtype=np.int32)
    else:
        off = end
        for e in elems:
            if e["name"] == "vertex":
                dt = np.dtype([(p[-1], np_t[p[0]]) for p in e["props"]])
                arr = np.frombuffer(data, dtype=dt, count=e["count"], offset=off)
                off += dt.itemsize * e["count"]
                verts = np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float32)
            elif e["name"] == "face":
                p = e["props"][0]          # list <count type> <index type> vertex_indices
                dt = np.dtype([("n", np_t[p[1]]), ("idx", np_t[p[2]], (3,))])
                arr = np.frombuffer(data, dtype=dt, count=e["count"], offset=off)
                off += dt.itemsize * e["count"]
                faces = arr["idx"].astype(np.int32)
    return TriangleMesh(verts, faces)
