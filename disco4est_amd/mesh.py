"""Synthetic meshes in d4est's data layout (host side, numpy).

Produces exactly the arrays the reference's ``d4est_mesh_update`` hands to the
operator (src/Mesh/d4est_mesh.c:2790, :2395-2470, :2544-2700, :2757-2776):
per element ``deg, deg_quad, nodal_stride, quad_stride`` and the SoA geometric
factors ``J_quad[local_nodes_quad]``, ``rst_xyz_quad[(3*i+j)*local_nodes_quad +
quad_stride + n] = d r_i / d x_j``.  Elements are ordered along p4est's Morton
(z-order) curve of a single-tree brick [0,1]^3 (x is the fastest bit).
"""
import numpy as np

from .capi import table

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64_uniform(seed, n, offset=0):
    """Repo-local fixed-seed generator: U[0,1) doubles from splitmix64(seed + index).
    (Stands in for the reference test's srand(102321)/rand(),
    src/Tests/Unit/d4est_test_laplacian_speedup.c:429, so CPU and GPU see identical data.)"""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def morton_order(level):
    """Integer coordinates (ix, iy, iz) of the 8^level elements in z-order (x fastest bit)."""
    n = 1 << level
    ix, iy, iz = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    ix, iy, iz = ix.ravel(), iy.ravel(), iz.ravel()
    key = np.zeros(ix.size, dtype=np.int64)
    for b in range(level):
        key |= ((ix >> b) & 1) << (3 * b)
        key |= ((iy >> b) & 1) << (3 * b + 1)
        key |= ((iz >> b) & 1) << (3 * b + 2)
    order = np.argsort(key, kind="stable")
    return np.stack([ix[order], iy[order], iz[order]], axis=1)


def quad_nodes(quad_type, deg_quad):
    return table("gauss_nodes" if quad_type == 0 else "lobatto_nodes", deg_quad)


class BrickMesh:
    """Uniform single-tree brick [0,1]^3 at refinement ``level`` with per-element degrees.

    ``deg`` may be an int or an array of length 8**level (mixed p);
    ``deg_quad_inc`` mirrors [initial_mesh] regionX_deg_quad_inc (src/Mesh/d4est_mesh.c:303-330).
    ``first``/``count`` select a contiguous Morton range (the shard of one rank).
    """

    def __init__(self, level, deg, deg_quad_inc=0, quad_type=0, first=0, count=None):
        self.level = level
        self.quad_type = quad_type
        ijk = morton_order(level)
        total = ijk.shape[0]
        deg_all = np.full(total, deg, dtype=np.int32) if np.isscalar(deg) else np.asarray(deg, dtype=np.int32)
        assert deg_all.size == total
        count = total - first if count is None else count
        self.global_elements = total
        self.first = first
        self.ijk = ijk[first:first + count]
        self.deg = deg_all[first:first + count].copy()
        self.deg_quad = (self.deg + deg_quad_inc).astype(np.int32)
        self.n_elements = count
        self.h = 1.0 / (1 << level)
        n3 = (self.deg.astype(np.int64) + 1) ** 3
        q3 = (self.deg_quad.astype(np.int64) + 1) ** 3
        self.nodal_stride = np.concatenate([[0], np.cumsum(n3)[:-1]]).astype(np.int32)
        self.quad_stride = np.concatenate([[0], np.cumsum(q3)[:-1]]).astype(np.int32)
        self.local_nodes = int(n3.sum())
        self.local_nodes_quad = int(q3.sum())

    # -- coordinates ---------------------------------------------------------
    def _ref_coords(self, e, nodes_1d):
        """brick coordinates X,Y,Z (each [n^3], x fastest) of tensor nodes in element e"""
        n = nodes_1d.size
        x0 = self.ijk[e] * self.h
        t = 0.5 * self.h * (nodes_1d + 1.0)
        X = np.broadcast_to(x0[0] + t[None, None, :], (n, n, n)).ravel()
        Y = np.broadcast_to(x0[1] + t[None, :, None], (n, n, n)).ravel()
        Z = np.broadcast_to(x0[2] + t[:, None, None], (n, n, n)).ravel()
        return X, Y, Z

    def nodal_coords(self, mapping=None):
        """physical x,y,z at the Lobatto nodes, element-ordered [local_nodes]"""
        out = [np.empty(self.local_nodes) for _ in range(3)]
        cache = {}
        for e in range(self.n_elements):
            p = int(self.deg[e])
            if p not in cache:
                cache[p] = table("lobatto_nodes", p)
            X, Y, Z = self._ref_coords(e, cache[p])
            if mapping is not None:
                X, Y, Z = mapping.x(X, Y, Z)
            s = self.nodal_stride[e]
            n3 = (p + 1) ** 3
            out[0][s:s + n3], out[1][s:s + n3], out[2][s:s + n3] = X, Y, Z
        return out

    def geometry(self, mapping=None):
        """(J_quad, rst_xyz_quad) in the reference SoA layout for the affine brick or a smooth map."""
        nq = self.local_nodes_quad
        J = np.empty(nq)
        rst = np.zeros((9, nq))
        if mapping is None:
            J[:] = (0.5 * self.h) ** 3
            for i in range(3):
                rst[3 * i + i, :] = 2.0 / self.h
            return J, rst.reshape(-1)
        cache = {}
        for e in range(self.n_elements):
            pq = int(self.deg_quad[e])
            if pq not in cache:
                cache[pq] = quad_nodes(self.quad_type, pq)
            X, Y, Z = self._ref_coords(e, cache[pq])
            DF = mapping.jacobian(X, Y, Z)           # [n,3,3] dx_i/dX_j
            dxdr = DF * (0.5 * self.h)               # dX_j/dr_j = h/2
            s = self.quad_stride[e]
            q3 = (pq + 1) ** 3
            J[s:s + q3] = np.linalg.det(dxdr)
            inv = np.linalg.inv(dxdr)                # dr_i/dx_j
            for i in range(3):
                for j in range(3):
                    rst[3 * i + j, s:s + q3] = inv[:, i, j]
        return J, rst.reshape(-1)

    def field(self, mapping=None, seed=102321, noise=1.0):
        """u = x^2 + y^2 + z^2 + noise * U[0,1) at the Lobatto nodes
        (the reference speed-up test's input, d4est_test_laplacian_speedup.c:429-432)."""
        x, y, z = self.nodal_coords(mapping)
        u = x * x + y * y + z * z
        if noise:
            u = u + noise * splitmix64_uniform(seed, self.local_nodes, offset=int(self.first) * 7919)
        return u


class SineMap:
    """Smooth invertible map of the unit cube, x = X + a * sin(pi X) sin(pi Y) sin(pi Z) * c,
    giving every element a full, spatially varying 3x3 dr/dx (stand-in for curved geometries)."""

    def __init__(self, amplitude=0.05, c=(1.0, -0.7, 0.4)):
        self.a = amplitude
        self.c = np.asarray(c, dtype=np.float64)

    def x(self, X, Y, Z):
        s = self.a * np.sin(np.pi * X) * np.sin(np.pi * Y) * np.sin(np.pi * Z)
        return X + self.c[0] * s, Y + self.c[1] * s, Z + self.c[2] * s

    def jacobian(self, X, Y, Z):
        sx, sy, sz = np.sin(np.pi * X), np.sin(np.pi * Y), np.sin(np.pi * Z)
        cx, cy, cz = np.cos(np.pi * X), np.cos(np.pi * Y), np.cos(np.pi * Z)
        g = self.a * np.pi * np.stack([cx * sy * sz, sx * cy * sz, sx * sy * cz], axis=-1)  # grad s
        DF = np.zeros(X.shape + (3, 3))
        for i in range(3):
            DF[..., i, :] = self.c[i] * g
            DF[..., i, i] += 1.0
        return DF
