"""Deterministic synthetic meshes for benchmarks and parity tests.

The reference ships no large-mesh generator on this path (its only sphere
builder is benchmarks/benchmark_volume_optimization.py:16-94, a subdivided
octahedron).  BASELINE.json quotes the metric on a class-I geodesic
icosphere, so that is what is generated here:

    frequency f  ->  nv = 10 f^2 + 2,  nf = 20 f^2
    f = 81  -> 65 612 / 131 220     (config 2)
    f = 320 -> 1 024 002 / 2 048 000 (configs 3 and 4)

Vertex row order is ``np.unique`` over coordinates rounded to 1e-9 (the order
the survey's oracle script used, i.e. lexicographic in x,y,z), facets are
oriented outward.  All arrays: positions (nv,3) float64 C-order, triangle
rows (nf,3) int32 C-order -- the layouts of ``Mesh.positions_view`` /
``Mesh.triangle_row_cache`` in the reference (geometry/mesh.py:372-389,
:597-624).
"""

from __future__ import annotations

import numpy as np

_T = (1.0 + 5.0**0.5) / 2.0
_ICO_V = np.array(
    [
        [-1, _T, 0], [1, _T, 0], [-1, -_T, 0], [1, -_T, 0],
        [0, -1, _T], [0, 1, _T], [0, -1, -_T], [0, 1, -_T],
        [_T, 0, -1], [_T, 0, 1], [-_T, 0, -1], [-_T, 0, 1],
    ],
    dtype=np.float64,
)
_ICO_F = np.array(
    [
        [0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
        [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
        [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
        [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1],
    ],
    dtype=np.int64,
)


def icosphere(freq: int, radius: float = 1.0) -> tuple[np.ndarray, np.ndarray]:
    """Class-I geodesic icosphere of frequency ``freq`` on a sphere.

    Returns ``(positions (nv,3) float64, tri_rows (nf,3) int32)``.
    """
    f = int(freq)
    if f < 1:
        raise ValueError("freq must be >= 1")
    # Barycentric lattice of one face: rows i = 0..f, columns j = 0..f-i.
    ii, jj = np.meshgrid(np.arange(f + 1), np.arange(f + 1), indexing="ij")
    keep = (ii + jj) <= f
    li = ii[keep]
    lj = jj[keep]
    # Flat index of lattice point (i, j) inside one face.
    row_start = np.concatenate([[0], np.cumsum(f + 1 - np.arange(f + 1))])[:-1]

    def lidx(i, j):
        return row_start[i] + j

    # "Up" triangles (i,j),(i+1,j),(i,j+1) and "down" triangles.
    ui, uj = np.meshgrid(np.arange(f), np.arange(f), indexing="ij")
    um = (ui + uj) <= f - 1
    ui, uj = ui[um], uj[um]
    up = np.stack([lidx(ui, uj), lidx(ui + 1, uj), lidx(ui, uj + 1)], axis=1)
    dm = (ui + uj) <= f - 2
    di, dj = ui[dm], uj[dm]
    down = np.stack(
        [lidx(di + 1, dj), lidx(di + 1, dj + 1), lidx(di, dj + 1)], axis=1
    )
    local_tris = np.concatenate([up, down], axis=0)
    npts = li.shape[0]

    pts = np.empty((20 * npts, 3), dtype=np.float64)
    tris = np.empty((20 * local_tris.shape[0], 3), dtype=np.int64)
    a = (li / f)[:, None]
    b = (lj / f)[:, None]
    for k, (ia, ib, ic) in enumerate(_ICO_F):
        A, B, C = _ICO_V[ia], _ICO_V[ib], _ICO_V[ic]
        pts[k * npts : (k + 1) * npts] = A + (B - A) * a + (C - A) * b
        tris[k * local_tris.shape[0] : (k + 1) * local_tris.shape[0]] = (
            local_tris + k * npts
        )
    pts /= np.linalg.norm(pts, axis=1)[:, None]
    # rows in lexicographic (x, y, z) order of the rounded coordinates, duplicates merged: what
    # np.unique(key, axis=0, return_index=True, return_inverse=True) returns, by a stable three-key sort (an order of
    # magnitude faster on the 16-million-point lattices of the scaling runs)
    key = np.round(pts, 9)
    order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))
    ks = key[order]
    new = np.empty(ks.shape[0], dtype=bool)
    new[0] = True
    np.any(ks[1:] != ks[:-1], axis=1, out=new[1:])
    first = order[new]  # (stable sort: the first row of a group is its earliest occurrence)
    inv = np.empty(ks.shape[0], dtype=np.int64)
    inv[order] = np.cumsum(new) - 1
    positions = np.ascontiguousarray(pts[first] * float(radius))
    tri_rows = inv[tris].astype(np.int32)
    # Orient outward (positive enclosed volume).
    v0 = positions[tri_rows[:, 0]]
    v1 = positions[tri_rows[:, 1]]
    v2 = positions[tri_rows[:, 2]]
    flip = np.einsum("ij,ij->i", np.cross(v1 - v0, v2 - v0), v0 + v1 + v2) < 0.0
    if np.any(flip):
        tri_rows[flip] = tri_rows[flip][:, [0, 2, 1]]
    return positions, np.ascontiguousarray(tri_rows)


def smooth_displace(positions: np.ndarray, amplitude: float = 0.05) -> np.ndarray:
    """Displace radially by a fixed smooth function so gradients are non-trivial.

    x <- x * (1 + amplitude * (x*y + 0.5*z^2))   
    """
    p = np.asarray(positions, dtype=np.float64)
    y = p[:, 0] * p[:, 1] + 0.5 * p[:, 2] ** 2
    return np.ascontiguousarray(p * (1.0 + amplitude * y)[:, None])


def disk_patch(n_rings: int, radius: float = 1.0, bulge: float = 0.3,
               jitter: float = 0.0, seed: int = 0):
    """Open triangulated cap with a boundary (exercises the boundary branches).

    Concentric rings, ring r has 6r vertices; z = bulge * (1 - rho^2).  With
    ``jitter`` > 0 the in-plane positions are perturbed (seeded), which
    produces obtuse triangles.  Returns (positions, tri_rows, is_boundary).
    """
    pts = [(0.0, 0.0)]
    ring_start = [0]
    for r in range(1, n_rings + 1):
        ring_start.append(len(pts))
        m = 6 * r
        for k in range(m):
            ang = 2.0 * np.pi * k / m
            rho = radius * r / n_rings
            pts.append((rho * np.cos(ang), rho * np.sin(ang)))
    xy = np.array(pts, dtype=np.float64)
    tris = []
    for r in range(1, n_rings + 1):
        m_out = 6 * r
        m_in = 6 * (r - 1)
        so = ring_start[r]
        si = ring_start[r - 1]
        for sector in range(6):
            for k in range(r):
                o0 = so + (sector * r + k) % m_out
                o1 = so + (sector * r + k + 1) % m_out
                if r == 1:
                    tris.append((0, o0, o1))
                    continue
                i0 = si + (sector * (r - 1) + k) % m_in
                if k < r - 1:
                    i1 = si + (sector * (r - 1) + k + 1) % m_in
                    tris.append((i0, o0, o1))
                    tris.append((i0, o1, i1))
                else:
                    tris.append((i0, o0, o1))
    tri_rows = np.array(tris, dtype=np.int32)
    rng = np.random.default_rng(seed)
    if jitter > 0.0:
        h = radius / n_rings
        interior = np.ones(len(xy), dtype=bool)
        interior[ring_start[n_rings]:] = False
        xy[interior] += jitter * h * rng.normal(size=(int(interior.sum()), 2))
    rho2 = (xy[:, 0] ** 2 + xy[:, 1] ** 2) / radius**2
    z = bulge * (1.0 - rho2)
    positions = np.ascontiguousarray(np.column_stack([xy, z]))
    is_boundary = np.zeros(len(xy), dtype=bool)
    is_boundary[ring_start[n_rings]:] = True
    return positions, tri_rows, is_boundary


def boundary_mask_from_triangles(nv: int, tri_rows: np.ndarray) -> np.ndarray:
    """Vertices on edges that belong to fewer than two facets.

    Array restatement of ``Mesh.boundary_vertex_ids`` (geometry/mesh.py:304-319).
    """
    t = np.asarray(tri_rows, dtype=np.int64)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]], axis=0)
    e.sort(axis=1)
    key = e[:, 0] * np.int64(nv) + e[:, 1]
    uniq, counts = np.unique(key, return_counts=True)
    b = uniq[counts < 2]
    mask = np.zeros(nv, dtype=bool)
    mask[(b // nv).astype(np.int64)] = True
    mask[(b % nv).astype(np.int64)] = True
    return mask
