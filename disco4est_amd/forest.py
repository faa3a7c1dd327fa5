"""Multi-tree (multi-block) synthetic meshes in d4est's data layout (host side, numpy).

``mesh.BrickMesh`` / ``HangingBrickMesh`` cover one p4est tree, where every face pair is (f, f^1) with orientation 0.  This module
builds the same arrays -- element lists, geometric factors, the flat side list with its mortar data -- for a FOREST of trees glued
through a p8est connectivity, so that faces between trees carry p4est's (face, face, orientation) triple and the engine's
``side_reorder`` / ``side_orientation`` inputs take every value (config 5: the reference's cubed-sphere multi-block meshes,
src/Geometry/d4est_connectivity_cubed_sphere.c:5-66, src/Problems/TwoPunctures/multi_options.input:62-68).

What it mirrors:
  * p8est connectivity conventions (p4est-2.8 src/p8est_connectivity.h:125-140: corners in z-order, faces -x +x -y +y -z +z,
    tree_to_face = face + 6 * orientation, orientation = the face corner of the higher-numbered face that the first corner of the
    lower-numbered face meets) and the face transform between neighbouring trees (p4est_expand_face_transform,
    p4est_quadrant_transform_face; src/p4est_connectivity.c:2877-2944, src/p4est_bits.c:1838-1925) -- restated, p4est is a
    third-party dependency of the reference;
  * the face iteration of the reference (src/Mesh/d4est_mortars.c:601-803): per (-) side the (+) element(s), their face, the
    orientation; hanging quadrants listed in the z-order of their OWN side's face;
  * the mortar factors (src/Mesh/d4est_mesh.c:858-1108): sj, n, drst_dxyz_m on the (-) side's mortar nodes in (-) order,
    drst_dxyz_p_porder evaluated in the (+) side's tree in the (+) side's own node and sub-face order, hp = J/sj of the (+) side
    re-oriented into (-) order.
"""
import numpy as np

from . import capi
from .mesh import morton_order, quad_nodes, splitmix64_uniform
from .capi import table

# p8est_face_corners (p4est-2.8 src/p8est_connectivity.c:29-35): corners of face f in the face's z-order
FACE_CORNERS = np.array([[0, 2, 4, 6], [1, 3, 5, 7], [0, 1, 4, 5], [2, 3, 6, 7], [0, 1, 2, 3], [4, 5, 6, 7]])
_PERM_REFS = np.array([[0, 1, 1, 0, 0, 1], [2, 0, 0, 1, 1, 0], [2, 0, 0, 1, 1, 0],
                       [0, 2, 2, 0, 0, 1], [0, 2, 2, 0, 0, 1], [2, 0, 0, 2, 2, 0]])   # p8est_face_permutation_refs


def expand_face_transform(iface, target_face, orientation):
    """p4est_expand_face_transform (p4est-2.8 src/p4est_connectivity.c:2877-2944): my_axis[3], target_axis[3], edge_reverse[3]"""
    ft = [0] * 9
    ft[0] = 1 if iface < 2 else 0
    ft[1] = 2 if iface < 4 else 1
    ft[2] = iface // 2
    rev = int(_PERM_REFS[0][iface]) ^ int(_PERM_REFS[0][target_face]) ^ int(orientation in (0, 3))
    ft[3 + rev] = 1 if target_face < 2 else 0
    ft[3 + (1 - rev)] = 2 if target_face < 4 else 1
    ft[5] = target_face // 2
    rev = int(_PERM_REFS[iface][target_face] == 1)
    ft[6 + rev] = orientation & 1
    ft[6 + (1 - rev)] = orientation >> 1
    ft[8] = 2 * (iface & 1) + (target_face & 1)
    return ft


def transform_quadrant(q, h, root, ft):
    """p4est_quadrant_transform_face (src/p4est_bits.c:1838-1925): a quadrant (corner q, side h) lying OUTSIDE its tree across the
    face the transform was made for, expressed in the neighbouring tree's coordinates (root = P4EST_ROOT_LEN)."""
    my_axis, target_axis, rev = ft[0:3], ft[3:6], ft[6:9]
    mh = -h
    Rmh = root + mh
    tRmh = root + Rmh
    r = [0, 0, 0]
    r[target_axis[0]] = q[my_axis[0]] if not rev[0] else Rmh - q[my_axis[0]]
    r[target_axis[1]] = q[my_axis[1]] if not rev[1] else Rmh - q[my_axis[1]]
    a = q[my_axis[2]]
    r[target_axis[2]] = (mh - a, a + root, a - root, tRmh - a)[rev[2]]
    return r


class Connectivity:
    """p8est connectivity: ``tree_to_tree[t, f]``, ``tree_to_face[t, f]`` (= face + 6 * orientation); a boundary face points
    at itself.  ``vertices`` / ``tree_to_vertex`` are optional (needed only by TrilinearMap)."""

    def __init__(self, tree_to_tree, tree_to_face, vertices=None, tree_to_vertex=None):
        self.tree_to_tree = np.asarray(tree_to_tree, dtype=np.int64).reshape(-1, 6)
        self.tree_to_face = np.asarray(tree_to_face, dtype=np.int64).reshape(-1, 6)
        self.num_trees = self.tree_to_tree.shape[0]
        self.vertices = None if vertices is None else np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
        self.tree_to_vertex = None if tree_to_vertex is None else np.asarray(tree_to_vertex, dtype=np.int64).reshape(-1, 8)

    @classmethod
    def from_vertices(cls, vertices, tree_to_vertex):
        """Derive tree_to_tree / tree_to_face from shared vertices with p8est's orientation rule (p8est_connectivity.h:133-140)."""
        ttv = np.asarray(tree_to_vertex, dtype=np.int64).reshape(-1, 8)
        nt = ttv.shape[0]
        ttt = np.repeat(np.arange(nt)[:, None], 6, axis=1)
        ttf = np.tile(np.arange(6), (nt, 1))
        faces = {}
        for t in range(nt):
            for f in range(6):
                faces.setdefault(tuple(sorted(ttv[t, FACE_CORNERS[f]].tolist())), []).append((t, f))
        for key, lst in faces.items():
            if len(lst) == 1:
                continue
            assert len(lst) == 2, "a face is shared by at most two trees"
            (t0, f0), (t1, f1) = lst
            (tl, fl), (th, fh) = ((t0, f0), (t1, f1)) if f0 <= f1 else ((t1, f1), (t0, f0))
            v0 = ttv[tl, FACE_CORNERS[fl][0]]
            o = int(np.nonzero(ttv[th, FACE_CORNERS[fh]] == v0)[0][0])
            ttt[t0, f0], ttf[t0, f0] = t1, f1 + 6 * o
            ttt[t1, f1], ttf[t1, f1] = t0, f0 + 6 * o
        return cls(ttt, ttf, vertices, ttv)

    @classmethod
    def brick(cls, nx, ny, nz):
        """nx x ny x nz unit cubes, every tree in the same orientation (all inter-tree faces have orientation 0)"""
        vid = lambda i, j, k: i + (nx + 1) * (j + (ny + 1) * k)
        verts = np.array([[i, j, k] for k in range(nz + 1) for j in range(ny + 1) for i in range(nx + 1)], dtype=np.float64)
        ttv = []
        for k in range(nz):
            for j in range(ny):
                for i in range(nx):
                    ttv.append([vid(i + (c & 1), j + ((c >> 1) & 1), k + ((c >> 2) & 1)) for c in range(8)])
        return cls.from_vertices(verts, ttv)

    @classmethod
    def rotated_pair(cls, rot_a, rot_b):
        """Two unit cubes A = [0,1]^3 and B = [1,2]x[0,1]^2 glued along x = 1, each numbered through a proper rotation of the cube
        (``cube_rotations()[k]``): the tree's reference axes are a rotated copy of the physical ones.  Over the 24 x 24 choices every
        (f_m, f_p, orientation) triple p4est allows between two right-handed trees occurs."""
        verts = np.array([[i, j, k] for k in range(2) for j in range(2) for i in range(3)], dtype=np.float64)
        vid = lambda i, j, k: i + 3 * (j + 2 * k)
        R = cube_rotations()
        ttv = []
        for x0, rot in ((0, R[rot_a]), (1, R[rot_b])):
            row = []
            for c in range(8):
                ref = np.array([c & 1, (c >> 1) & 1, (c >> 2) & 1]) * 2 - 1        # corner in [-1,1]^3 reference coordinates
                phys = (rot @ ref + 1) // 2                                        # rotated corner of the physical unit cube
                row.append(vid(int(x0 + phys[0]), int(phys[1]), int(phys[2])))
            ttv.append(row)
        return cls.from_vertices(verts, ttv)


def reference_reorientation_is_consistent(f_m, f_p, orientation):
    """d4est_operators_reorient_face_data (dGMath/d4est_operators.c:2031-2081) always expands the transform of the pair
    (lower face, higher face) and applies flip0, flip1, transpose in that order, whichever of the two is the (-) side.  That is the
    geometric map from the (+) face ordering to the (-) one whenever the face is not transposed, or both flips are equal, or the
    (-) face is the higher-numbered one; for the remaining triples (transpose with exactly one flip, seen from the lower face: 36 of
    the 144) out(a, b) = in(flip0(b), flip1(a)) is applied where in(flip1(b), flip0(a)) would be geometric.  The reference's own
    connectivities (cubed sphere: codes 0, 1, 2, 3, 7) never produce such a pair.  The engine follows the reference either way."""
    return not (face_reorder_code(f_m, f_p, orientation) in (5, 6) and f_m <= f_p)


def face_reorder_code(f_m, f_p, orientation):
    """host-side twin of d4est_hip_face_reorder_code (dGMath/d4est_operators.c:2031-2050) for code that must not load the library
    (test collection); tests/test_forest.py checks the three implementations (this, the library's, the oracle's) against each other"""
    ft = expand_face_transform(min(f_m, f_p), max(f_m, f_p), orientation)
    aligned = (ft[1] - ft[0]) * (ft[4] - ft[3]) > 0
    return ft[6] | (ft[7] << 1) | ((0 if aligned else 1) << 2)


def cube_rotations():
    """the 24 proper rotations of the cube as signed permutation matrices (det = +1), in a fixed order"""
    import itertools
    out = []
    for perm in itertools.permutations(range(3)):
        for signs in itertools.product((1, -1), repeat=3):
            M = np.zeros((3, 3), dtype=np.int64)
            for i in range(3):
                M[i, perm[i]] = signs[i]
            if round(np.linalg.det(M)) == 1:
                out.append(M)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# tree maps: x(tree, xi) and dx/dxi(tree, xi) for tree coordinates xi in [0,1]^3
# ---------------------------------------------------------------------------------------------------------------------
class TrilinearMap:
    """The vertex map p4est itself uses (trilinear interpolation of the tree's 8 vertices), optionally followed by a smooth warp of
    physical space ``warp.x(X,Y,Z)``, ``warp.jacobian(X,Y,Z)`` (e.g. mesh.SineMap) that makes every element curved."""

    def __init__(self, conn, warp=None):
        assert conn.vertices is not None
        self.V = conn.vertices[conn.tree_to_vertex]      # [nt, 8, 3]
        self.warp = warp

    def _tri(self, tree, xi):
        V = self.V[tree]
        a, b, c = xi[:, 0], xi[:, 1], xi[:, 2]
        X = np.zeros((xi.shape[0], 3))
        D = np.zeros((xi.shape[0], 3, 3))
        for k in range(8):
            s = [(k >> d) & 1 for d in range(3)]
            w = [np.where(s[0], a, 1 - a), np.where(s[1], b, 1 - b), np.where(s[2], c, 1 - c)]
            dw = [1.0 if s[d] else -1.0 for d in range(3)]
            X += (w[0] * w[1] * w[2])[:, None] * V[k][None, :]
            D[:, :, 0] += (dw[0] * w[1] * w[2])[:, None] * V[k][None, :]
            D[:, :, 1] += (w[0] * dw[1] * w[2])[:, None] * V[k][None, :]
            D[:, :, 2] += (w[0] * w[1] * dw[2])[:, None] * V[k][None, :]
        return X, D

    def x(self, tree, xi):
        X, _ = self._tri(tree, xi)
        if self.warp is not None:
            X = np.stack(self.warp.x(X[:, 0], X[:, 1], X[:, 2]), axis=1)
        return X

    def jacobian(self, tree, xi):
        X, D = self._tri(tree, xi)
        if self.warp is not None:
            D = self.warp.jacobian(X[:, 0], X[:, 1], X[:, 2]) @ D
        return D


class CubedSphere7Map:
    """The reference's 7-tree cubed sphere: six wedges around a centre cube (d4est_geometry_cubed_sphere_7tree_X,
    src/Geometry/d4est_geometry_cubed_sphere.c:498-580; parameters R0, R1, compactify_inner_shell of
    [geometry] name = cubed_sphere_7tree).  Tree coordinates -> (a, b, c) in [-1,1]^2 x [1,2]; with p = 2 - c,
    x = p a + (1-p) tan(pi a / 4), y likewise, q = R / sqrt(1 + (1-p)(tan^2 + tan^2) + 2p); the wedge index picks the
    axis permutation.  The centre cube is abc * Clength with Clength = R0 / sqrt(3) so that the surfaces meet.
    The Jacobian is formed here by the chain rule (the reference carries machine-generated closed forms, :1294-1360)."""

    def __init__(self, R0=1.0, R1=2.0, compactify=False):
        self.R0, self.R1, self.compactify = float(R0), float(R1), bool(compactify)
        self.Clength = self.R0 / np.sqrt(3.0)

    def _wedge(self, xi):
        a, b, c = 2 * xi[:, 0] - 1, 2 * xi[:, 1] - 1, xi[:, 2] + 1
        da, db, dc = 2.0, 2.0, 1.0
        R0, R1 = self.R0, self.R1
        if self.compactify:
            m = 1.0 / (1.0 / R1 - 1.0 / R0)
            t = (R0 - 2.0 * R1) / (R0 - R1)
            R = m / (c - t)
            dR = -m / (c - t) ** 2
        else:
            R = R0 * (2.0 - c) + R1 * (c - 1.0)
            dR = np.full_like(c, R1 - R0)
        p = 2.0 - c
        tx, ty = np.tan(a * np.pi / 4), np.tan(b * np.pi / 4)
        dtx, dty = (np.pi / 4) * (1 + tx * tx), (np.pi / 4) * (1 + ty * ty)
        x = p * a + (1 - p) * tx
        y = p * b + (1 - p) * ty
        S = 1.0 + (1 - p) * (tx * tx + ty * ty) + 2 * p
        q = R / np.sqrt(S)
        # derivatives w.r.t. (a, b, c); dp/dc = -1
        dx = [p + (1 - p) * dtx, np.zeros_like(a), -a + tx]
        dy = [np.zeros_like(a), p + (1 - p) * dty, -b + ty]
        dS = [(1 - p) * 2 * tx * dtx, (1 - p) * 2 * ty * dty, (tx * tx + ty * ty) - 2.0]
        dq = [-0.5 * R * S ** -1.5 * dS[0], -0.5 * R * S ** -1.5 * dS[1], dR / np.sqrt(S) - 0.5 * R * S ** -1.5 * dS[2]]
        sc = [da, db, dc]
        qx = [(dq[k] * x + q * dx[k]) * sc[k] for k in range(3)]
        qy = [(dq[k] * y + q * dy[k]) * sc[k] for k in range(3)]
        qq = [dq[k] * sc[k] for k in range(3)]
        return q * x, q * y, q, qx, qy, qq

    # wedge -> (x, y, z) as signed picks of (q x, q y, q), src/Geometry/d4est_geometry_cubed_sphere.c:543-577
    _PICK = [((0, 1), (2, -1), (1, 1)), ((0, 1), (1, 1), (2, 1)), ((0, 1), (2, 1), (1, -1)),
             ((2, 1), (0, -1), (1, -1)), ((1, -1), (0, -1), (2, -1)), ((2, -1), (0, -1), (1, 1))]

    def x(self, tree, xi):
        if tree == 6:
            return (2 * xi - 1) * self.Clength
        qx, qy, q, _, _, _ = self._wedge(xi)
        comp = (qx, qy, q)
        return np.stack([s * comp[i] for (i, s) in self._PICK[tree]], axis=1)

    def jacobian(self, tree, xi):
        if tree == 6:
            return np.broadcast_to(2 * self.Clength * np.eye(3), (xi.shape[0], 3, 3)).copy()
        _, _, _, dqx, dqy, dq = self._wedge(xi)
        comp = (dqx, dqy, dq)
        D = np.zeros((xi.shape[0], 3, 3))
        for row, (i, s) in enumerate(self._PICK[tree]):
            for k in range(3):
                D[:, row, k] = s * comp[i][k]
        return D


def cubed_sphere_7tree_connectivity():
    """tree_to_tree / tree_to_face of d4est_connectivity_new_sphere_7tree from the committed fixture (numbers extracted from
    src/Geometry/d4est_connectivity_cubed_sphere.c:41-58 by tests/golden/make_reference_tables.py)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cubed_sphere_7tree_connectivity.json")
    with open(path) as fh:
        d = json.load(fh)
    return Connectivity(d["tree_to_tree"], d["tree_to_face"])


# ---------------------------------------------------------------------------------------------------------------------
class ForestMesh:
    """Forest of trees at a uniform base ``level`` with optional ONE level of local refinement (``refine``: bool over the
    num_trees * 8**level base cells, tree-major Morton order; the caller keeps it 2:1 balanced across faces).  Elements are in
    p4est order (tree by tree, Morton inside a tree).  ``deg`` is an int or an array over the GLOBAL elements; ``first``/``count``
    or ``elements`` select one rank's shard as in mesh.BrickMesh; off-rank face neighbours become ghost elements."""

    def __init__(self, conn, level, deg, mapping, refine=None, deg_quad_inc=0, quad_type=0, first=0, count=None, elements=None):
        self.conn, self.level, self.quad_type, self.mapping = conn, level, quad_type, mapping
        base = morton_order(level)
        nb = base.shape[0]
        nt = conn.num_trees
        refine = np.zeros(nt * nb, dtype=bool) if refine is None else np.asarray(refine, dtype=bool)
        assert refine.size == nt * nb
        tree, org, size = [], [], []          # origin in units of the FINE grid (2^(level+1) per tree side), size 1 or 2
        for t in range(nt):
            for b in range(nb):
                o = 2 * base[b]
                if refine[t * nb + b]:
                    for c in range(8):
                        tree.append(t); org.append(o + np.array([c & 1, (c >> 1) & 1, (c >> 2) & 1])); size.append(1)
                else:
                    tree.append(t); org.append(o); size.append(2)
        self._tree_all = np.asarray(tree, dtype=np.int64)
        self._org_all = np.asarray(org, dtype=np.int64)
        self._size_all = np.asarray(size, dtype=np.int64)
        self.nf = 1 << (level + 1)
        total = self._org_all.shape[0]
        deg_all = np.full(total, deg, dtype=np.int32) if np.isscalar(deg) else np.asarray(deg, dtype=np.int32)
        assert deg_all.size == total
        if elements is None:
            count = total - first if count is None else count
            elements = np.arange(first, first + count, dtype=np.int64)
        else:
            elements = np.asarray(elements, dtype=np.int64)
            count = int(elements.size)
            first = int(elements[0]) if count else 0
        self.elements = elements
        self._g2l = -np.ones(total, dtype=np.int64)
        self._g2l[elements] = np.arange(count)
        self.global_elements, self.first, self.n_elements = total, first, count
        self.tree, self.org, self.size = self._tree_all[elements], self._org_all[elements], self._size_all[elements]
        self.deg = deg_all[elements].copy()
        self.deg_quad = (self.deg + deg_quad_inc).astype(np.int32)
        n3 = (self.deg.astype(np.int64) + 1) ** 3
        q3 = (self.deg_quad.astype(np.int64) + 1) ** 3
        self.nodal_stride = np.concatenate([[0], np.cumsum(n3)[:-1]]).astype(np.int32) if count else np.zeros(0, np.int32)
        self.quad_stride = np.concatenate([[0], np.cumsum(q3)[:-1]]).astype(np.int32) if count else np.zeros(0, np.int32)
        self.local_nodes, self.local_nodes_quad = int(n3.sum()), int(q3.sum())
        self.deg_global = deg_all
        self.deg_quad_global = (deg_all + deg_quad_inc).astype(np.int32)
        g3 = (deg_all.astype(np.int64) + 1) ** 3
        self.global_nodal_stride = np.concatenate([[0], np.cumsum(g3)[:-1]])
        self.global_nodes = int(g3.sum())
        self.global_nodal_offset = int(self.global_nodal_stride[first]) if count > 0 else 0
        self._owner = -np.ones((nt, self.nf, self.nf, self.nf), dtype=np.int64)
        for g in range(total):
            t, o, sz = self._tree_all[g], self._org_all[g], self._size_all[g]
            self._owner[t, o[0]:o[0] + sz, o[1]:o[1] + sz, o[2]:o[2] + sz] = g

    # -- coordinates ---------------------------------------------------------
    def _cell_xi(self, org, size, ref):
        """tree coordinates in [0,1]^3 of reference points ref[n,3] in [-1,1]^3 of the cell (org, size) on the fine grid"""
        return (np.asarray(org, dtype=np.float64)[None, :] + 0.5 * size * (ref + 1.0)) / self.nf

    @staticmethod
    def _tensor_ref(nodes_1d):
        n = nodes_1d.size
        ref = np.empty((n, n, n, 3))
        ref[..., 0] = nodes_1d[None, None, :]
        ref[..., 1] = nodes_1d[None, :, None]
        ref[..., 2] = nodes_1d[:, None, None]
        return ref.reshape(-1, 3)

    @staticmethod
    def _face_ref(f, nodes_1d):
        """reference points of face f, tangential axes in increasing order, the first fastest"""
        d, sgn = f // 2, (1.0 if f % 2 else -1.0)
        ax = [a for a in range(3) if a != d]
        n = nodes_1d.size
        ref = np.zeros((n * n, 3))
        ref[:, d] = sgn
        ref[:, ax[0]] = np.tile(nodes_1d, n)
        ref[:, ax[1]] = np.repeat(nodes_1d, n)
        return ref

    def nodal_coords(self, mapping=None):
        mp = mapping or self.mapping
        out = [np.empty(self.local_nodes) for _ in range(3)]
        for e in range(self.n_elements):
            p = int(self.deg[e])
            X = mp.x(int(self.tree[e]), self._cell_xi(self.org[e], self.size[e], self._tensor_ref(table("lobatto_nodes", p))))
            s, n3 = self.nodal_stride[e], (p + 1) ** 3
            for d in range(3):
                out[d][s:s + n3] = X[:, d]
        return out

    def geometry(self, mapping=None):
        """(J_quad, rst_xyz_quad) in the reference SoA layout (src/Mesh/d4est_mesh.c:2544-2700)"""
        mp = mapping or self.mapping
        nq = self.local_nodes_quad
        J = np.empty(nq)
        rst = np.zeros((9, nq))
        for e in range(self.n_elements):
            pq = int(self.deg_quad[e])
            xi = self._cell_xi(self.org[e], self.size[e], self._tensor_ref(quad_nodes(self.quad_type, pq)))
            dxdr = mp.jacobian(int(self.tree[e]), xi) * (0.5 * self.size[e] / self.nf)
            s, q3 = self.quad_stride[e], (pq + 1) ** 3
            J[s:s + q3] = np.linalg.det(dxdr)
            inv = np.linalg.inv(dxdr)
            for i in range(3):
                for j in range(3):
                    rst[3 * i + j, s:s + q3] = inv[:, i, j]
        return J, rst.reshape(-1)

    def field(self, mapping=None, seed=102321, noise=1.0):
        x, y, z = self.nodal_coords(mapping)
        u = x * x + y * y + z * z
        if noise:
            u = u + noise * splitmix64_uniform(seed, self.local_nodes, offset=self.global_nodal_offset)
        return u

    # -- topology ------------------------------------------------------------
    def _across(self, t, q, h, f):
        """the cell (corner q, side h, fine units) just outside face f of tree t, in the neighbouring tree's coordinates:
        (tree', q', f', orientation) or None on a domain boundary"""
        tp, code = int(self.conn.tree_to_tree[t, f]), int(self.conn.tree_to_face[t, f])
        if tp == t and code == f:
            return None
        fp, o = code % 6, code // 6
        ft = expand_face_transform(f, fp, o)
        return tp, transform_quadrant(q, h, self.nf, ft), fp, o

    def _neighbour_cell(self, g, f, h=None, shift=(0, 0)):
        """cell of side h across face f of global element g, shifted by `shift` half-steps along the face's tangential axes;
        returns (tree', q', f', orientation) with q' inside tree', or None"""
        t, o, sz = int(self._tree_all[g]), self._org_all[g], int(self._size_all[g])
        h = sz if h is None else h
        d, pos = f // 2, f % 2
        ax = [a for a in range(3) if a != d]
        q = [int(v) for v in o]
        q[d] = o[d] + sz if pos else o[d] - h
        q[ax[0]] += shift[0] * h
        q[ax[1]] += shift[1] * h
        if 0 <= q[d] < self.nf:
            return t, q, f ^ 1, 0
        return self._across(t, q, h, f)

    def face_neighbours(self, g, f):
        """What the reference's face iteration reports for the (-) side (g, f):
        ('boundary',) | ('full', g_p, f_p, o) | ('big', [4 x g_p in (-) order], f_p, o) | ('small', g_big, f_p, o, sub, [own group])"""
        sz = int(self._size_all[g])
        nb = self._neighbour_cell(g, f)
        if nb is None:
            return ("boundary",)
        tp, qp, fp, o = nb
        gp = int(self._owner[tp, qp[0], qp[1], qp[2]])
        gs = int(self._size_all[gp])
        if gs == sz:
            return ("full", gp, fp, o)
        d = f // 2
        ax = [a for a in range(3) if a != d]
        if gs < sz:
            four = []
            for i in range(4):
                tq, qq, _, _ = self._neighbour_cell(g, f, h=sz // 2, shift=(i & 1, i >> 1))
                four.append(int(self._owner[tq, qq[0], qq[1], qq[2]]))
            return ("big", four, fp, o)
        og = self._org_all[g]
        sub = int(og[ax[0]] & 1) + 2 * int(og[ax[1]] & 1)
        t = int(self._tree_all[g])
        grp = []
        for i in range(4):
            c = [int(v) for v in og]
            c[ax[0]] = (int(og[ax[0]]) & ~1) + (i & 1)
            c[ax[1]] = (int(og[ax[1]]) & ~1) + (i >> 1)
            grp.append(int(self._owner[t, c[0], c[1], c[2]]))
        return ("small", gp, fp, o, sub, grp)

    def has_hanging(self):
        return bool(np.any(self._size_all == 1) and np.any(self._size_all == 2))

    # -- geometry on a face ---------------------------------------------------
    def _mortar_geom(self, mp, t, org, size, f, pq):
        """sj, n, dr/dx (inverse Jacobian), J/sj at the mortar quadrature nodes of face f of the (virtual) cell (org, size) of tree t
        (src/Mesh/d4est_mortars.c:19-190, COMPUTE_NORMAL_USING_JACOBIAN) + the physical coordinates of those nodes"""
        d, sgn = f // 2, (1.0 if f % 2 else -1.0)
        xi = self._cell_xi(org, size, self._face_ref(f, quad_nodes(self.quad_type, pq)))
        dxdr = mp.jacobian(t, xi) * (0.5 * size / self.nf)
        Jm = np.linalg.det(dxdr)
        inv = np.linalg.inv(dxdr)
        v = sgn * Jm[:, None] * inv[:, d, :]
        sjv = np.linalg.norm(v, axis=1)
        return sjv, v / sjv[:, None], inv, Jm / sjv, mp.x(t, xi)

    @staticmethod
    def reorient(arr, code, n):
        """d4est_operators_reorient_face_data with the (flip0, flip1, transpose) code on an n x n face array (first index fastest)"""
        a = np.asarray(arr).reshape(n, n)       # a[b, a]
        if code & 1:
            a = a[:, ::-1]
        if code & 2:
            a = a[::-1, :]
        if code & 4:
            a = a.T
        return np.ascontiguousarray(a).reshape(-1)

    def build_sides(self, mapping=None):
        """Flat side list + mortar factors (same keys as mesh.HangingBrickMesh.build_sides; hanging arrays only when the mesh has
        hanging faces).  Extra key ``mortar_xyz_mismatch``: the largest distance between a mortar node's physical position computed
        on the (-) side and on the (+) side after the reference's re-orientation -- a self-check of the topology (0 to rounding;
        NOT 0 where the reference's re-orientation itself is geometrically inconsistent, see ``reference_reorientation_is_consistent``);
        ``hanging_order_mismatch``: number of hanging sub-faces whose (+) element, found geometrically, is not where
        d4est_reference_reorient_face_order says it is."""
        mp = mapping or self.mapping
        lib = capi.load_library()
        ne = self.n_elements
        hanging = self.has_hanging()
        kinds = [[self.face_neighbours(int(self.elements[e]), f) for f in range(6)] for e in range(ne)]
        # ghosts: every off-rank element referenced
        refd = []
        for e in range(ne):
            for f in range(6):
                k = kinds[e][f]
                if k[0] == "full":
                    refd.append(k[1])
                elif k[0] == "big":
                    refd.extend(k[1])
                elif k[0] == "small":
                    refd.append(k[1]); refd.extend(k[5])
        refd = np.asarray(refd, dtype=np.int64)
        ghost_ids = np.unique(refd[self._g2l[refd] < 0]) if refd.size else np.zeros(0, dtype=np.int64)
        ghost_pos = {int(g): i for i, g in enumerate(ghost_ids)}
        enc = lambda g: int(self._g2l[g]) if self._g2l[g] >= 0 else -(ghost_pos[int(g)] + 2)
        side_nbr = np.full(6 * ne, -1, dtype=np.int32)
        side_nbr_face = np.zeros(6 * ne, dtype=np.int32)
        side_reorder = np.zeros(6 * ne, dtype=np.int32)
        side_orientation = np.zeros(6 * ne, dtype=np.int32)
        side_hang = np.zeros(6 * ne, dtype=np.int32)
        side_sub = np.zeros(6 * ne, dtype=np.int32)
        side_nbr4 = np.full(4 * 6 * ne, -1, dtype=np.int32)
        side_mortar_stride = np.zeros(6 * ne, dtype=np.int32)
        degq_g = self.deg_quad_global
        blocks = []        # (S0, [(-) sub-mortar descriptors], [(+) descriptors in (+) order] or None, f_m, f_p, o, code)
        total = 0
        done_group = {}
        order_mismatch = 0
        for e in range(ne):
            ge = int(self.elements[e])
            t, o_e, sz = int(self.tree[e]), self.org[e], int(self.size[e])
            for f in range(6):
                s_ = 6 * e + f
                k = kinds[e][f]
                d = f // 2
                ax = [a for a in range(3) if a != d]
                side_nbr_face[s_] = f ^ 1
                if k[0] == "boundary":
                    pq = int(degq_g[ge])
                    side_mortar_stride[s_] = total
                    blocks.append((total, [(t, o_e, sz, f, pq)], None, f, f, 0, 0))
                    total += (pq + 1) ** 2
                    continue
                fp, o = k[2], k[3]
                code = int(lib.d4est_hip_face_reorder_code(f, fp, o))      # 0 for (f, f^1, 0): the faces inside a tree
                side_nbr_face[s_], side_reorder[s_], side_orientation[s_] = fp, code, o
                if k[0] == "full":
                    gp = k[1]
                    side_nbr[s_] = enc(gp)
                    side_nbr4[4 * s_] = side_nbr[s_]
                    pq = int(max(degq_g[ge], degq_g[gp]))
                    side_mortar_stride[s_] = total
                    blocks.append((total, [(t, o_e, sz, f, pq)],
                                   [(int(self._tree_all[gp]), self._org_all[gp], int(self._size_all[gp]), fp, pq)], f, fp, o, code))
                    total += (pq + 1) ** 2
                elif k[0] == "big":
                    side_hang[s_] = 1
                    four = k[1]
                    for i in range(4):
                        side_nbr4[4 * s_ + i] = enc(four[i])
                    side_nbr[s_] = side_nbr4[4 * s_]
                    # (-) sub-mortars: the half-size virtual children of this element that touch the face, z-order of the face
                    subs_m, pqs = [], []
                    for i in range(4):
                        pq = int(max(degq_g[ge], degq_g[four[i]]))
                        oc = np.array(o_e, dtype=np.int64).copy()
                        oc[ax[0]] += (i & 1) * (sz // 2)
                        oc[ax[1]] += (i >> 1) * (sz // 2)
                        if f % 2:
                            oc[d] += sz // 2
                        subs_m.append((t, oc, sz // 2, f, pq))
                        pqs.append(pq)
                    # (+) side in its OWN order: position j holds the element that is i = inverse(j) in (-) order
                    subs_p = [None] * 4
                    for i in range(4):
                        j = int(lib.d4est_hip_reorient_face_order(f, fp, o, i))
                        gp = four[i]
                        subs_p[j] = (int(self._tree_all[gp]), self._org_all[gp], int(self._size_all[gp]), fp, pqs[i])
                        # p4est lists the hanging quadrants of a side in the z-order of that side's own face: the element found
                        # geometrically across sub-face i must sit at position j of its own side
                        axp = [a for a in range(3) if a != fp // 2]
                        own = int(self._org_all[gp][axp[0]] & 1) + 2 * int(self._org_all[gp][axp[1]] & 1)
                        order_mismatch += int(own != j)
                    side_mortar_stride[s_] = total
                    blocks.append((total, subs_m, subs_p, f, fp, o, code))
                    total += sum((pq + 1) ** 2 for pq in pqs)
                else:
                    side_hang[s_] = 2
                    gp, sub, grp = k[1], k[4], k[5]
                    side_nbr[s_] = enc(gp)
                    side_sub[s_] = sub
                    for i in range(4):
                        side_nbr4[4 * s_ + i] = enc(grp[i])
                    key = (tuple(grp), f)
                    if key in done_group:          # the block was created by the group's first LOCAL member (d4est_mesh.c:956-962)
                        side_mortar_stride[s_] = done_group[key]
                        continue
                    subs_m, pqs = [], []
                    for i in range(4):
                        gm = grp[i]
                        pq = int(max(degq_g[gm], degq_g[gp]))
                        subs_m.append((int(self._tree_all[gm]), self._org_all[gm], int(self._size_all[gm]), f, pq))
                        pqs.append(pq)
                    # (+) side = the big element: its half-size virtual children in ITS own face z-order; child j faces the small
                    # element that is i in (-) order with j = reorient_face_order(i)
                    tpb, opb, szb = int(self._tree_all[gp]), self._org_all[gp], int(self._size_all[gp])
                    dp = fp // 2
                    axp = [a for a in range(3) if a != dp]
                    subs_p = [None] * 4
                    for i in range(4):
                        j = int(lib.d4est_hip_reorient_face_order(f, fp, o, i))
                        oc = np.array(opb, dtype=np.int64).copy()
                        oc[axp[0]] += (j & 1) * (szb // 2)
                        oc[axp[1]] += (j >> 1) * (szb // 2)
                        if fp % 2:
                            oc[dp] += szb // 2
                        subs_p[j] = (tpb, oc, szb // 2, fp, pqs[i])
                    done_group[key] = total
                    side_mortar_stride[s_] = total
                    blocks.append((total, subs_m, subs_p, f, fp, o, code))
                    total += sum((pq + 1) ** 2 for pq in pqs)
        sj = np.empty(total); hm_a = np.empty(total); hp_a = np.empty(total)
        nrm = np.zeros(3 * total); drst_m = np.zeros(9 * total); drst_p = np.zeros(9 * total)
        mismatch = 0.0
        for (S0, subs_m, subs_p, f_m, f_p, o, code) in blocks:
            Ttot = sum((sm[4] + 1) ** 2 for sm in subs_m)
            nsub = len(subs_m)
            off = 0
            offs_m = []
            xm = []
            for (t, oc, sz, f, pq) in subs_m:
                Tn = (pq + 1) ** 2
                sjv, nv, inv, hv, X = self._mortar_geom(mp, t, oc, sz, f, pq)
                sj[S0 + off:S0 + off + Tn] = sjv
                hm_a[S0 + off:S0 + off + Tn] = hv
                for j in range(3):
                    nrm[3 * S0 + j * Ttot + off:3 * S0 + j * Ttot + off + Tn] = nv[:, j]
                for i in range(3):
                    for j in range(3):
                        a0 = 9 * S0 + (i + 3 * j) * Ttot + off
                        drst_m[a0:a0 + Tn] = inv[:, i, j]
                offs_m.append(off)
                xm.append(X)
                off += Tn
            if subs_p is None:          # boundary: the (+) arrays are never read; keep them finite
                hp_a[S0:S0 + Ttot] = hm_a[S0:S0 + Ttot]
                for c in range(9):
                    drst_p[9 * S0 + c * Ttot:9 * S0 + (c + 1) * Ttot] = drst_m[9 * S0 + c * Ttot:9 * S0 + (c + 1) * Ttot]
                continue
            # (+) side in porder: sub-faces in the (+) side's own order, nodes in its own order (d4est_mesh.c:1011-1031)
            off_p = 0
            hp_porder, xp_porder, offs_p = [], [], []
            for (t, oc, sz, f, pq) in subs_p:
                Tn = (pq + 1) ** 2
                _, _, inv, hv, X = self._mortar_geom(mp, t, oc, sz, f, pq)
                for i in range(3):
                    for j in range(3):
                        a0 = 9 * S0 + (i + 3 * j) * Ttot + off_p
                        drst_p[a0:a0 + Tn] = inv[:, i, j]
                hp_porder.append(hv); xp_porder.append(X); offs_p.append(off_p)
                off_p += Tn
            # hp = J/sj of the (+) side re-oriented into (-) order (d4est_mesh.c:1034-1060)
            for i in range(nsub):
                j = i if nsub == 1 else int(lib.d4est_hip_reorient_face_order(f_m, f_p, o, i))
                pq = subs_m[i][4]
                Tn = (pq + 1) ** 2
                hp_a[S0 + offs_m[i]:S0 + offs_m[i] + Tn] = self.reorient(hp_porder[j], code, pq + 1)
                for c in range(3):
                    xr = self.reorient(xp_porder[j][:, c], code, pq + 1)
                    mismatch = max(mismatch, float(np.abs(xr - xm[i][:, c]).max()))
        # boundary sides: Dirichlet values live on the Lobatto face nodes
        bnd = side_nbr == -1
        deg_m = np.repeat(self.deg, 6)
        nbn = np.where(bnd, (deg_m.astype(np.int64) + 1) ** 2, 0)
        side_bndry_stride = np.concatenate([[0], np.cumsum(nbn)[:-1]]).astype(np.int32) if ne else np.zeros(0, np.int32)
        total_bndry = int(nbn.sum())
        bndry_xyz = np.zeros((3, total_bndry))
        for s_ in np.nonzero(bnd)[0]:
            e, f = divmod(int(s_), 6)
            p = int(self.deg[e])
            X = mp.x(int(self.tree[e]), self._cell_xi(self.org[e], self.size[e], self._face_ref(f, table("lobatto_nodes", p))))
            B0 = int(side_bndry_stride[s_])
            bndry_xyz[:, B0:B0 + (p + 1) ** 2] = X.T
        ghost_deg = self.deg_global[ghost_ids].astype(np.int32)
        ghost_deg_quad = self.deg_quad_global[ghost_ids].astype(np.int32)
        gn3 = (ghost_deg.astype(np.int64) + 1) ** 3
        ghost_nodal_stride = np.concatenate([[0], np.cumsum(gn3)[:-1]]).astype(np.int32) if len(ghost_ids) else np.zeros(0, np.int32)
        out = dict(side_nbr=side_nbr, side_nbr_face=side_nbr_face, side_reorder=side_reorder,
                   side_mortar_stride=side_mortar_stride, side_bndry_stride=side_bndry_stride,
                   total_mortar_nodes=total, total_bndry_nodes=total_bndry, bndry_xyz=bndry_xyz,
                   sj=sj, n=nrm, drst_m=drst_m, drst_p=drst_p, hm=hm_a, hp=hp_a,
                   ghost_global_ids=ghost_ids, ghost_deg=ghost_deg, ghost_deg_quad=ghost_deg_quad,
                   ghost_nodal_stride=ghost_nodal_stride, ghost_nodes=int(gn3.sum()), mortar_xyz_mismatch=mismatch,
                   hanging_order_mismatch=order_mismatch)
        if hanging:
            out.update(side_hang=side_hang, side_sub=side_sub, side_nbr4=side_nbr4, side_orientation=side_orientation)
        return out

    def build_sides_c(self):
        """The side arrays of Plan.set_faces from d4est_hip_build_sides (csrc/d4est_hip_sides.cpp: connectivity + quadrant list, the
        host-side replacement of the p4est_iterate face walk) WITHOUT any geometric factor -- for plans whose mortar factors are
        generated on the device (Plan.set_faces(..., analytic=...) / brick=...).  Whole meshes only (no ghost quadrants)."""
        import ctypes
        if self.n_elements != self.global_elements:
            raise ValueError("build_sides_c: whole meshes only (a shard's ghost quadrants come from build_sides)")
        lib = capi.load_library()
        ne = self.n_elements
        tree, q, dq = self.cells()
        I = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        ins = [I(self.conn.tree_to_tree), I(self.conn.tree_to_face), I(tree), I(q), I(dq), I(self.deg), I(self.deg_quad)]
        z = np.zeros(1, dtype=np.int32)
        out = {k: np.zeros(6 * ne, dtype=np.int32) for k in ("side_nbr", "side_nbr_face", "side_reorder", "side_orientation", "side_hang",
                                                             "side_sub", "side_mortar_stride", "side_bndry_stride")}
        out["side_nbr4"] = np.zeros(24 * ne, dtype=np.int32)
        tm, tb = ctypes.c_int(0), ctypes.c_int(0)
        hang = lib.d4est_hip_build_sides(self.conn.num_trees, vp(ins[0]), vp(ins[1]), self.nf, ne, vp(ins[2]), vp(ins[3]), vp(ins[4]), vp(ins[5]),
                                         vp(ins[6]), 0, vp(z), vp(z), vp(z), vp(z), vp(out["side_nbr"]), vp(out["side_nbr_face"]),
                                         vp(out["side_reorder"]), vp(out["side_orientation"]), vp(out["side_hang"]), vp(out["side_sub"]),
                                         vp(out["side_nbr4"]), vp(out["side_mortar_stride"]), vp(out["side_bndry_stride"]),
                                         ctypes.byref(tm), ctypes.byref(tb))
        out.update(total_mortar_nodes=tm.value, total_bndry_nodes=tb.value, ghost_deg=np.zeros(0, np.int32), ghost_deg_quad=np.zeros(0, np.int32),
                   ghost_global_ids=np.zeros(0, np.int64), ghost_nodes=0)
        if not hang:
            for k in ("side_hang", "side_sub", "side_nbr4", "side_orientation"):
                out.pop(k)
        return out

    def cells(self, global_ids=None):
        """(tree, q[n,3], dq) of the local elements (or of the given global ids, e.g. sides["ghost_global_ids"]) in units of the fine
        grid, root length = self.nf: d4est_element_data_t::tree, ::q, ::dq for d4est_hip_plan_set_geometry_analytic"""
        g = self.elements if global_ids is None else np.asarray(global_ids, dtype=np.int64)
        return (self._tree_all[g].astype(np.int32), self._org_all[g].astype(np.int32), self._size_all[g].astype(np.int32))

    def gather_ghost(self, sides, u_global):
        out = np.empty(sides["ghost_nodes"])
        for i, g in enumerate(sides["ghost_global_ids"]):
            n3 = (int(sides["ghost_deg"][i]) + 1) ** 3
            s0 = int(self.global_nodal_stride[g])
            out[sides["ghost_nodal_stride"][i]:sides["ghost_nodal_stride"][i] + n3] = u_global[s0:s0 + n3]
        return out
