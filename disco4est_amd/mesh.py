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
    ``first``/``count`` select a contiguous Morton range (the shard of one rank); ``elements`` (global ids, any order) selects an
    arbitrary element list instead, e.g. a shard followed by its ghost layer (the Schwarz smoother's extended mesh).
    """

    def __init__(self, level, deg, deg_quad_inc=0, quad_type=0, first=0, count=None, elements=None, domain=None):
        """domain: the computational domain is the first `domain` elements of the level's Morton sequence (default: the whole cube);
        faces towards the rest of the cube are domain boundary -- a box of 1, 2, 4 ... level-(L-1) sub-cubes for weak-scaling runs"""
        self.level = level
        self.domain = domain
        self.quad_type = quad_type
        ijk = morton_order(level)
        total = ijk.shape[0]
        deg_all = np.full(total, deg, dtype=np.int32) if np.isscalar(deg) else np.asarray(deg, dtype=np.int32)
        assert deg_all.size == total
        if elements is None:
            count = total - first if count is None else count
            elements = np.arange(first, first + count, dtype=np.int64)
        else:
            elements = np.asarray(elements, dtype=np.int64)
            count = int(elements.size)
            first = int(elements[0]) if count else 0
        self.elements = elements                       # global id of every local element
        self._g2l = -np.ones(total, dtype=np.int64)    # global id -> local index (-1: not on this mesh)
        self._g2l[elements] = np.arange(count)
        self.global_elements = total
        self.first = first
        self.ijk = ijk[elements]
        self.deg = deg_all[elements].copy()
        self.deg_quad = (self.deg + deg_quad_inc).astype(np.int32)
        self.n_elements = count
        self.h = 1.0 / (1 << level)
        n3 = (self.deg.astype(np.int64) + 1) ** 3
        q3 = (self.deg_quad.astype(np.int64) + 1) ** 3
        self.nodal_stride = np.concatenate([[0], np.cumsum(n3)[:-1]]).astype(np.int32)
        self.quad_stride = np.concatenate([[0], np.cumsum(q3)[:-1]]).astype(np.int32)
        self.local_nodes = int(n3.sum())
        self.local_nodes_quad = int(q3.sum())
        # global bookkeeping (for shards: where this rank's DoFs sit in the global element-ordered vector)
        self.deg_global = deg_all
        self.deg_quad_global = (deg_all + deg_quad_inc).astype(np.int32)
        g3 = (deg_all.astype(np.int64) + 1) ** 3
        self.global_nodal_stride = np.concatenate([[0], np.cumsum(g3)[:-1]])
        self.global_nodes = int(g3.sum())
        self.global_nodal_offset = int(self.global_nodal_stride[first]) if count > 0 else 0
        self._ijk_all = ijk

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
            u = u + noise * splitmix64_uniform(seed, self.local_nodes, offset=self.global_nodal_offset)
        return u


    # -- faces ---------------------------------------------------------------
    def build_sides(self, mapping=None, geometry=True):
        """Flat (element, face) side list + mortar geometric factors in the reference's layout (geometry=False: the side list and the
        strides only -- for plans whose mortar factors are generated on the device, Plan.set_faces(..., brick=...)).

        Mirrors what d4est's face iteration and d4est_mesh_compute_mortar_quadrature_quantities produce
        (src/Mesh/d4est_mortars.c:601-803, src/Mesh/d4est_mesh.c:868-1110): for side s = 6*e + f the (+)
        neighbour (local id, -1 = domain boundary, <= -2 = ghost element g = -(v+2)), its face, the face
        re-orientation code (always 0 inside one tree) and the offset of the side's mortar quadrature data:
        sj[S+k], n[3S+d*T+k], drst[9S+(i+3j)*T+k] = d r_i/d x_j, hm/hp = J/sj (FACE_H_EQ_J_DIV_SJ_QUAD).
        Ghost elements are the off-rank face neighbours, ordered by global Morton index.
        """
        n = 1 << self.level
        lookup = -np.ones((n, n, n), dtype=np.int64)
        lookup[self._ijk_all[:, 0], self._ijk_all[:, 1], self._ijk_all[:, 2]] = np.arange(self.global_elements)
        ne = self.n_elements
        side_nbr = np.full(6 * ne, -1, dtype=np.int32)
        side_nbr_face = np.zeros(6 * ne, dtype=np.int32)
        side_reorder = np.zeros(6 * ne, dtype=np.int32)
        nbr_global = np.full(6 * ne, -1, dtype=np.int64)
        for f in range(6):
            d, sgn = f // 2, (1 if f % 2 else -1)
            c = self.ijk.copy()
            c[:, d] += sgn
            inside = (c[:, d] >= 0) & (c[:, d] < n)
            g = np.where(inside, lookup[np.clip(c[:, 0], 0, n - 1), np.clip(c[:, 1], 0, n - 1), np.clip(c[:, 2], 0, n - 1)], -1)
            if self.domain is not None:
                g = np.where(g >= self.domain, -1, g)   # outside the domain: a boundary face
            nbr_global[f::6] = g
            side_nbr_face[f::6] = f ^ 1
        local = (nbr_global >= 0) & (self._g2l[np.clip(nbr_global, 0, None)] >= 0)
        ghost_ids = np.unique(nbr_global[(nbr_global >= 0) & ~local])
        ghost_pos = {int(g): i for i, g in enumerate(ghost_ids)}
        for s_ in range(6 * ne):
            g = int(nbr_global[s_])
            if g < 0:
                side_nbr[s_] = -1
            elif local[s_]:
                side_nbr[s_] = self._g2l[g]
            else:
                side_nbr[s_] = -(ghost_pos[g] + 2)
        ghost_deg = self.deg_global[ghost_ids].astype(np.int32)
        ghost_deg_quad = self.deg_quad_global[ghost_ids].astype(np.int32)
        gn3 = (ghost_deg.astype(np.int64) + 1) ** 3
        ghost_nodal_stride = np.concatenate([[0], np.cumsum(gn3)[:-1]]).astype(np.int32) if len(ghost_ids) else np.zeros(0, np.int32)
        # per-side sizes / strides
        deg_p = np.where(side_nbr >= 0, self.deg[np.clip(side_nbr, 0, None)], 0)
        degq_p = np.where(side_nbr >= 0, self.deg_quad[np.clip(side_nbr, 0, None)], 0)
        gh = side_nbr <= -2
        if gh.any():
            gi = -(side_nbr[gh] + 2)
            deg_p[gh] = ghost_deg[gi]
            degq_p[gh] = ghost_deg_quad[gi]
        deg_m = np.repeat(self.deg, 6)
        degq_m = np.repeat(self.deg_quad, 6)
        bnd = side_nbr == -1
        degq_mortar = np.where(bnd, degq_m, np.maximum(degq_m, degq_p))
        T = (degq_mortar.astype(np.int64) + 1) ** 2
        side_mortar_stride = np.concatenate([[0], np.cumsum(T)[:-1]]).astype(np.int32)
        total = int(T.sum())
        nb = np.where(bnd, (deg_m.astype(np.int64) + 1) ** 2, 0)
        side_bndry_stride = np.concatenate([[0], np.cumsum(nb)[:-1]]).astype(np.int32)
        total_bndry = int(nb.sum())
        if not geometry:
            return dict(side_nbr=side_nbr, side_nbr_face=side_nbr_face, side_reorder=side_reorder,
                        side_mortar_stride=side_mortar_stride, side_bndry_stride=side_bndry_stride,
                        total_mortar_nodes=total, total_bndry_nodes=total_bndry,
                        ghost_global_ids=ghost_ids, ghost_deg=ghost_deg, ghost_deg_quad=ghost_deg_quad,
                        ghost_nodal_stride=ghost_nodal_stride, ghost_nodes=int(gn3.sum()))
        sj = np.empty(total); hm = np.empty(total); hp = np.empty(total)
        nrm = np.zeros(3 * total); drst_m = np.zeros(9 * total); drst_p = np.zeros(9 * total)
        bndry_xyz = np.zeros((3, total_bndry))
        h = self.h
        qcache, lcache = {}, {}
        for s_ in range(6 * ne):
            e, f = divmod(s_, 6)
            d, sgn = f // 2, (1.0 if f % 2 else -1.0)
            pq = int(degq_mortar[s_])
            if pq not in qcache:
                qcache[pq] = quad_nodes(self.quad_type, pq)
            t = qcache[pq]
            S, Tn = int(side_mortar_stride[s_]), int(T[s_])
            x0 = self.ijk[e] * h
            ax = [a for a in range(3) if a != d]           # tangential axes, increasing; first is fastest
            ref = np.zeros((Tn, 3))
            ref[:, d] = sgn
            ref[:, ax[0]] = np.tile(t, pq + 1)
            ref[:, ax[1]] = np.repeat(t, pq + 1)
            X = x0[None, :] + 0.5 * h * (ref + 1.0)
            if mapping is None:
                DF = np.broadcast_to(np.eye(3), (Tn, 3, 3))
            else:
                DF = mapping.jacobian(X[:, 0], X[:, 1], X[:, 2])
            dxdr = DF * (0.5 * h)
            Jm = np.linalg.det(dxdr)
            inv = np.linalg.inv(dxdr)
            v = sgn * Jm[:, None] * inv[:, d, :]
            sjv = np.linalg.norm(v, axis=1)
            sj[S:S + Tn] = sjv
            for j in range(3):
                nrm[3 * S + j * Tn:3 * S + (j + 1) * Tn] = v[:, j] / sjv
            for i in range(3):
                for j in range(3):
                    drst_m[9 * S + (i + 3 * j) * Tn:9 * S + (i + 3 * j + 1) * Tn] = inv[:, i, j]
                    drst_p[9 * S + (i + 3 * j) * Tn:9 * S + (i + 3 * j + 1) * Tn] = inv[:, i, j]  # same size, same tree
            hm[S:S + Tn] = Jm / sjv
            hp[S:S + Tn] = Jm / sjv
            if bnd[s_]:
                p = int(self.deg[e])
                if p not in lcache:
                    lcache[p] = table("lobatto_nodes", p)
                tl = lcache[p]
                nbn = (p + 1) ** 2
                refl = np.zeros((nbn, 3))
                refl[:, d] = sgn
                refl[:, ax[0]] = np.tile(tl, p + 1)
                refl[:, ax[1]] = np.repeat(tl, p + 1)
                XL = x0[None, :] + 0.5 * h * (refl + 1.0)
                if mapping is not None:
                    XL = np.stack(mapping.x(XL[:, 0], XL[:, 1], XL[:, 2]), axis=1)
                B0 = int(side_bndry_stride[s_])
                bndry_xyz[:, B0:B0 + nbn] = XL.T
        return dict(side_nbr=side_nbr, side_nbr_face=side_nbr_face, side_reorder=side_reorder,
                    side_mortar_stride=side_mortar_stride, side_bndry_stride=side_bndry_stride,
                    total_mortar_nodes=total, total_bndry_nodes=total_bndry, bndry_xyz=bndry_xyz,
                    sj=sj, n=nrm, drst_m=drst_m, drst_p=drst_p, hm=hm, hp=hp,
                    ghost_global_ids=ghost_ids, ghost_deg=ghost_deg, ghost_deg_quad=ghost_deg_quad,
                    ghost_nodal_stride=ghost_nodal_stride, ghost_nodes=int(gn3.sum()))

    def gather_ghost(self, sides, u_global):
        """ghost-element data (whole elements, as d4est_ghost_data_exchange delivers them,
        src/Mesh/d4est_ghost_data.c:143-256) taken from a GLOBAL element-ordered vector"""
        out = np.empty(sides["ghost_nodes"])
        for i, g in enumerate(sides["ghost_global_ids"]):
            n3 = (int(sides["ghost_deg"][i]) + 1) ** 3
            s0 = int(self.global_nodal_stride[g])
            out[sides["ghost_nodal_stride"][i]:sides["ghost_nodal_stride"][i] + n3] = u_global[s0:s0 + n3]
        return out


class HangingBrickMesh(BrickMesh):
    """Single-tree brick with ONE level of local refinement: the base cells flagged in ``refine`` (bool array over the
    8**level base cells in Morton order) are replaced, in place, by their 8 children (z-order).  Neighbouring levels
    differ by at most one, so the mesh is 2:1 balanced and has hanging (1 <-> 4) faces wherever a refined cell touches an
    unrefined one -- the non-conforming mortars of src/Mesh/d4est_mortars.c:601-803.
    ``deg`` is an int or an array over the GLOBAL elements; ``first``/``count`` select a contiguous range of the global
    (Morton) element list = the shard of one rank; off-rank face neighbours become ghost elements."""

    def __init__(self, level, refine, deg, deg_quad_inc=0, quad_type=0, first=0, count=None, elements=None):
        self.level = level
        self.quad_type = quad_type
        base = morton_order(level)
        refine = np.asarray(refine, dtype=bool)
        assert refine.size == base.shape[0]
        org, size = [], []          # origin in units of the FINE grid (2^(level+1) per side), size 1 or 2
        for b in range(base.shape[0]):
            o = 2 * base[b]
            if refine[b]:
                for c in range(8):
                    org.append(o + np.array([c & 1, (c >> 1) & 1, (c >> 2) & 1]))
                    size.append(1)
            else:
                org.append(o)
                size.append(2)
        self._org_all = np.asarray(org, dtype=np.int64)
        self._size_all = np.asarray(size, dtype=np.int64)
        total = self._org_all.shape[0]
        deg_all = np.full(total, deg, dtype=np.int32) if np.isscalar(deg) else np.asarray(deg, dtype=np.int32)
        assert deg_all.size == total
        if elements is None:
            count = total - first if count is None else count
            elements = np.arange(first, first + count, dtype=np.int64)
        else:                                   # arbitrary element list (a shard followed by its ghost layer)
            elements = np.asarray(elements, dtype=np.int64)
            count = int(elements.size)
            first = int(elements[0]) if count else 0
        self.elements = elements
        self._g2l = -np.ones(total, dtype=np.int64)
        self._g2l[elements] = np.arange(count)
        self.global_elements = total
        self.first = first
        self.n_elements = count
        self.org = self._org_all[elements]
        self.size = self._size_all[elements]
        self.deg = deg_all[elements].copy()
        self.deg_quad = (self.deg + deg_quad_inc).astype(np.int32)
        self.hf = 1.0 / (1 << (level + 1))      # fine grid spacing
        self.h_elem = self.size * self.hf
        self._h_all = self._size_all * self.hf
        n3 = (self.deg.astype(np.int64) + 1) ** 3
        q3 = (self.deg_quad.astype(np.int64) + 1) ** 3
        self.nodal_stride = np.concatenate([[0], np.cumsum(n3)[:-1]]).astype(np.int32) if count else np.zeros(0, np.int32)
        self.quad_stride = np.concatenate([[0], np.cumsum(q3)[:-1]]).astype(np.int32) if count else np.zeros(0, np.int32)
        self.local_nodes = int(n3.sum())
        self.local_nodes_quad = int(q3.sum())
        self.deg_global = deg_all
        self.deg_quad_global = (deg_all + deg_quad_inc).astype(np.int32)
        g3 = (deg_all.astype(np.int64) + 1) ** 3
        self.global_nodal_stride = np.concatenate([[0], np.cumsum(g3)[:-1]])
        self.global_nodes = int(g3.sum())
        self.global_nodal_offset = int(self.global_nodal_stride[first]) if count > 0 else 0

    def _ref_coords(self, e, nodes_1d):
        n = nodes_1d.size
        x0 = self.org[e] * self.hf
        t = 0.5 * self.h_elem[e] * (nodes_1d + 1.0)
        X = np.broadcast_to(x0[0] + t[None, None, :], (n, n, n)).ravel()
        Y = np.broadcast_to(x0[1] + t[None, :, None], (n, n, n)).ravel()
        Z = np.broadcast_to(x0[2] + t[:, None, None], (n, n, n)).ravel()
        return X, Y, Z

    def geometry(self, mapping=None):
        nq = self.local_nodes_quad
        J = np.empty(nq)
        rst = np.zeros((9, nq))
        cache = {}
        for e in range(self.n_elements):
            pq = int(self.deg_quad[e])
            if pq not in cache:
                cache[pq] = quad_nodes(self.quad_type, pq)
            X, Y, Z = self._ref_coords(e, cache[pq])
            DF = mapping.jacobian(X, Y, Z) if mapping is not None else np.broadcast_to(np.eye(3), (X.size, 3, 3))
            dxdr = DF * (0.5 * self.h_elem[e])
            s = self.quad_stride[e]
            q3 = (pq + 1) ** 3
            J[s:s + q3] = np.linalg.det(dxdr)
            inv = np.linalg.inv(dxdr)
            for i in range(3):
                for j in range(3):
                    rst[3 * i + j, s:s + q3] = inv[:, i, j]
        return J, rst.reshape(-1)

    def _mortar_geom(self, x0, hm, f, pq, mapping):
        """geometric factors on one mortar face: the face f of a (virtual) cell with origin x0 and size hm
        (src/Mesh/d4est_mortars.c:19-190: the mortar is its own reference square, dq = mortar_dq)"""
        d, sgn = f // 2, (1.0 if f % 2 else -1.0)
        t = quad_nodes(self.quad_type, pq)
        Tn = (pq + 1) ** 2
        ax = [a for a in range(3) if a != d]
        ref = np.zeros((Tn, 3))
        ref[:, d] = sgn
        ref[:, ax[0]] = np.tile(t, pq + 1)
        ref[:, ax[1]] = np.repeat(t, pq + 1)
        X = x0[None, :] + 0.5 * hm * (ref + 1.0)
        DF = mapping.jacobian(X[:, 0], X[:, 1], X[:, 2]) if mapping is not None else np.broadcast_to(np.eye(3), (Tn, 3, 3))
        dxdr = DF * (0.5 * hm)
        Jm = np.linalg.det(dxdr)
        inv = np.linalg.inv(dxdr)
        v = sgn * Jm[:, None] * inv[:, d, :]
        sjv = np.linalg.norm(v, axis=1)
        return sjv, v / sjv[:, None], inv, Jm / sjv

    def build_sides(self, mapping=None):
        """Side list with hanging faces.  In addition to BrickMesh.build_sides():
          side_hang[s]        0 conforming / boundary, 1 big side (this face is split: faces_m = 1, faces_p = 4),
                              2 small side (this element is one of the 4 hanging elements: faces_m = 4, faces_p = 1)
          side_sub[s]         small side: index c of this element among the 4 (z-order of the face children)
          side_nbr4[4s..4s+3] big side: the 4 (+) elements in (-) order; small side: the 4 members of its own group
          side_orientation[s] p4est orientation (0 inside one tree)
        Element references (side_nbr, side_nbr4) are local ids, or -(g+2) for ghost element g (off-rank, ordered by global id).
        Mortar data in the reference's layout (src/Mesh/d4est_mesh.c:868-1110): a big side owns one block of its 4
        sub-mortars; the LOCAL small sides of a hanging face share one block (same side_mortar_stride) holding the 4
        sub-mortars one after another; vector/matrix components are strided by the block's total node count."""
        ne = self.n_elements
        nf = 1 << (self.level + 1)
        owner = -np.ones((nf, nf, nf), dtype=np.int64)
        for g in range(self.global_elements):
            o, sz = self._org_all[g], self._size_all[g]
            owner[o[0]:o[0] + sz, o[1]:o[1] + sz, o[2]:o[2] + sz] = g
        first = self.first
        is_local = lambda g: self._g2l[g] >= 0
        # pass 1: global neighbour ids per side
        nbr_g = np.full(6 * ne, -1, dtype=np.int64)
        nbr4_g = np.full(4 * 6 * ne, -1, dtype=np.int64)
        side_nbr_face = np.zeros(6 * ne, dtype=np.int32)
        side_hang = np.zeros(6 * ne, dtype=np.int32)
        side_sub = np.zeros(6 * ne, dtype=np.int32)
        for e in range(ne):
            o, sz = self.org[e], int(self.size[e])
            for f in range(6):
                s_ = 6 * e + f
                d, pos = f // 2, f % 2
                ax = [a for a in range(3) if a != d]
                side_nbr_face[s_] = f ^ 1
                c0 = o.copy()
                c0[d] = o[d] + sz if pos else o[d] - 1
                if c0[d] < 0 or c0[d] >= nf:
                    continue
                g = int(owner[c0[0], c0[1], c0[2]])
                gs = int(self._size_all[g])
                if gs == sz:
                    nbr_g[s_] = g
                    nbr4_g[4 * s_] = g
                elif gs < sz:  # this element is the big one
                    side_hang[s_] = 1
                    for i in range(4):
                        ci = c0.copy()
                        ci[ax[0]] += i & 1
                        ci[ax[1]] += i >> 1
                        nbr4_g[4 * s_ + i] = int(owner[ci[0], ci[1], ci[2]])
                    nbr_g[s_] = nbr4_g[4 * s_]
                else:          # this element is one of the four small ones
                    side_hang[s_] = 2
                    nbr_g[s_] = g
                    go = self._org_all[g]
                    ia, ib = int(o[ax[0]] - go[ax[0]]), int(o[ax[1]] - go[ax[1]])
                    side_sub[s_] = ia + 2 * ib
                    for i in range(4):   # own group: the children of the big neighbour's face, z-order
                        ci = o.copy()
                        ci[ax[0]] = go[ax[0]] + (i & 1)
                        ci[ax[1]] = go[ax[1]] + (i >> 1)
                        nbr4_g[4 * s_ + i] = int(owner[ci[0], ci[1], ci[2]])
        # ghosts: every off-rank element referenced
        refd = np.concatenate([nbr_g, nbr4_g])
        refd = refd[refd >= 0]
        ghost_ids = np.unique(refd[self._g2l[refd] < 0])
        ghost_pos = {int(g): i for i, g in enumerate(ghost_ids)}
        enc = lambda g: -1 if g < 0 else (int(self._g2l[g]) if is_local(g) else -(ghost_pos[int(g)] + 2))
        side_nbr = np.array([enc(g) for g in nbr_g], dtype=np.int32)
        side_nbr4 = np.array([enc(g) if g >= 0 else -1 for g in nbr4_g], dtype=np.int32)
        side_reorder = np.zeros(6 * ne, dtype=np.int32)
        ghost_deg = self.deg_global[ghost_ids].astype(np.int32)
        ghost_deg_quad = self.deg_quad_global[ghost_ids].astype(np.int32)
        gn3 = (ghost_deg.astype(np.int64) + 1) ** 3
        ghost_nodal_stride = np.concatenate([[0], np.cumsum(gn3)[:-1]]).astype(np.int32) if len(ghost_ids) else np.zeros(0, np.int32)
        degq_g = self.deg_quad_global
        # ---- mortar blocks (local (-) sides only)
        side_mortar_stride = np.zeros(6 * ne, dtype=np.int32)
        blocks = []   # (stride S, [(x0, hm, f, pq)] per sub-mortar)
        total = 0
        done_group = {}
        for s_ in range(6 * ne):
            e, f = divmod(s_, 6)
            ge = int(self.elements[e])
            hang = side_hang[s_]
            if hang == 0:
                g = int(nbr_g[s_])
                pq = int(degq_g[ge]) if g < 0 else int(max(degq_g[ge], degq_g[g]))
                side_mortar_stride[s_] = total
                blocks.append((total, [(self.org[e] * self.hf, self.h_elem[e], f, pq)]))
                total += (pq + 1) ** 2
            elif hang == 1:
                d = f // 2
                ax = [a for a in range(3) if a != d]
                subs = []
                side_mortar_stride[s_] = total
                S0 = total
                for i in range(4):
                    g = int(nbr4_g[4 * s_ + i])
                    pq = int(max(degq_g[ge], degq_g[g]))
                    hm = 0.5 * self.h_elem[e]
                    x0 = (self.org[e] * self.hf).copy()
                    x0[ax[0]] += (i & 1) * hm
                    x0[ax[1]] += (i >> 1) * hm
                    if f % 2:
                        x0[d] += hm   # the virtual child touching the +face
                    subs.append((x0, hm, f, pq))
                    total += (pq + 1) ** 2
                blocks.append((S0, subs))
            else:
                grp = tuple(int(x) for x in nbr4_g[4 * s_:4 * s_ + 4])
                key = (grp, f)
                if key in done_group:          # the block was created by the group's first LOCAL member (d4est_mesh.c:956-962)
                    side_mortar_stride[s_] = done_group[key]
                    continue
                g = int(nbr_g[s_])
                subs = []
                S0 = total
                for i in range(4):
                    em = grp[i]
                    pq = int(max(degq_g[em], degq_g[g]))
                    subs.append((self._org_all[em] * self.hf, self._h_all[em], f, pq))
                    total += (pq + 1) ** 2
                done_group[key] = S0
                side_mortar_stride[s_] = S0
                blocks.append((S0, subs))
        sj = np.empty(total); hm_a = np.empty(total); hp_a = np.empty(total)
        nrm = np.zeros(3 * total); drst_m = np.zeros(9 * total); drst_p = np.zeros(9 * total)
        for S0, subs in blocks:
            Ttot = sum((pq + 1) ** 2 for (_, _, _, pq) in subs)
            off = 0
            for (x0, hm, f, pq) in subs:
                Tn = (pq + 1) ** 2
                sjv, nv, inv, hv = self._mortar_geom(np.asarray(x0, dtype=np.float64), hm, f, pq, mapping)
                sj[S0 + off:S0 + off + Tn] = sjv
                hm_a[S0 + off:S0 + off + Tn] = hv
                hp_a[S0 + off:S0 + off + Tn] = hv   # same mortar seen from the other side (continuous map, orientation 0)
                for j in range(3):
                    nrm[3 * S0 + j * Ttot + off:3 * S0 + j * Ttot + off + Tn] = nv[:, j]
                for i in range(3):
                    for j in range(3):
                        a0 = 9 * S0 + (i + 3 * j) * Ttot + off
                        drst_m[a0:a0 + Tn] = inv[:, i, j]
                        drst_p[a0:a0 + Tn] = inv[:, i, j]   # (+) side, (+) order == (-) order inside one tree
                off += Tn
        # boundary sides: Dirichlet values live on the Lobatto face nodes
        bnd = nbr_g == -1
        deg_m = np.repeat(self.deg, 6)
        nb = np.where(bnd, (deg_m.astype(np.int64) + 1) ** 2, 0)
        side_bndry_stride = np.concatenate([[0], np.cumsum(nb)[:-1]]).astype(np.int32) if ne else np.zeros(0, np.int32)
        total_bndry = int(nb.sum())
        bndry_xyz = np.zeros((3, total_bndry))
        for s_ in np.nonzero(bnd)[0]:
            e, f = divmod(int(s_), 6)
            d, sgn = f // 2, (1.0 if f % 2 else -1.0)
            ax = [a for a in range(3) if a != d]
            p = int(self.deg[e])
            tl = table("lobatto_nodes", p)
            nbn = (p + 1) ** 2
            refl = np.zeros((nbn, 3))
            refl[:, d] = sgn
            refl[:, ax[0]] = np.tile(tl, p + 1)
            refl[:, ax[1]] = np.repeat(tl, p + 1)
            XL = (self.org[e] * self.hf)[None, :] + 0.5 * self.h_elem[e] * (refl + 1.0)
            if mapping is not None:
                XL = np.stack(mapping.x(XL[:, 0], XL[:, 1], XL[:, 2]), axis=1)
            B0 = int(side_bndry_stride[s_])
            bndry_xyz[:, B0:B0 + nbn] = XL.T
        return dict(side_nbr=side_nbr, side_nbr_face=side_nbr_face, side_reorder=side_reorder,
                    side_mortar_stride=side_mortar_stride, side_bndry_stride=side_bndry_stride,
                    total_mortar_nodes=total, total_bndry_nodes=total_bndry, bndry_xyz=bndry_xyz,
                    sj=sj, n=nrm, drst_m=drst_m, drst_p=drst_p, hm=hm_a, hp=hp_a,
                    side_hang=side_hang, side_sub=side_sub, side_nbr4=side_nbr4,
                    side_orientation=np.zeros(6 * ne, dtype=np.int32),
                    ghost_global_ids=ghost_ids, ghost_deg=ghost_deg, ghost_deg_quad=ghost_deg_quad,
                    ghost_nodal_stride=ghost_nodal_stride, ghost_nodes=int(gn3.sum()))


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
