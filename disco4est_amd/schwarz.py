"""Host side of the additive Schwarz smoother.

Mirrors the reference's setup for it:
  * ``SchwarzMetadata``  <->  d4est_solver_schwarz_metadata_init (src/Solver/d4est_solver_schwarz_metadata.c:186-405, :447-520):
    one subdomain per local element = the core and every element that shares a face, an edge or a corner with it, sorted by
    (tree, quadid); per subdomain element the ``faces`` that touch the core and the mirrored ``core_faces``.
  * ``subdomain_sides``  <->  d4est_solver_schwarz_geometric_data_init (src/Solver/d4est_solver_schwarz_geometric_data.c): the mortar
    list of every subdomain with its zero_and_skip flags, expressed as the side arrays of a *subdomain plan* (see
    csrc/d4est_hip_schwarz.hip): neighbours inside the subdomain are the local copies, neighbours outside become ghost sides that the
    smoother feeds with a zero trace; geometric factors are not copied, the strides alias the mesh's arrays.
  * ``Schwarz``          <->  d4est_solver_schwarz_t + d4est_solver_schwarz_iterate (src/Solver/d4est_solver_schwarz.c:20-285).

Meshes: ``mesh.BrickMesh`` (uniform or mixed degrees), ``mesh.HangingBrickMesh`` (hanging 1 <-> 4 faces) on 1 .. N ranks, and
``forest.ForestMesh`` (several trees, inter-tree face orientation, hanging faces) on one rank.
"""
import ctypes

import numpy as np

from . import capi

_vp = ctypes.c_void_p


class SchwarzMetadata:
    """Flat d4est_solver_schwarz_metadata_t (src/Solver/d4est_solver_schwarz_metadata.h:19-62)."""

    def __init__(self, mesh, sides, num_nodes_overlap, cores=None, sort_key=None):
        """cores: local indices of the elements that get a subdomain (default: all); on a rank's extended mesh (own elements + ghost
        layer, see SchwarzShard) these are the own elements.  sort_key: per element, the (tree, quadid) order used inside a subdomain
        (default: the local index; on an extended mesh the global id)."""
        ne = mesh.n_elements
        if num_nodes_overlap <= 0:
            raise ValueError("num_nodes_overlap <= 0")                      # d4est_solver_schwarz_metadata.c:166-168
        if num_nodes_overlap == 1:
            # overlap_size = 1 - r[deg] = 0: the hat weights (d4est_solver_schwarz_operators.c:27-40) are 0/0 at the element ends
            raise ValueError("num_nodes_overlap = 1 gives a zero-width weight ramp")
        if num_nodes_overlap > int(mesh.deg.min()) + 1:
            raise ValueError("num_nodes_overlap exceeds the minimum mesh degree + 1")   # metadata.h:69-70
        nbr = np.asarray(sides["side_nbr"]).reshape(ne, 6)
        nbr4 = np.asarray(sides["side_nbr4"]) if "side_nbr4" in sides else np.zeros(0, dtype=np.int32)
        hanging = "side_hang" in sides and np.any(np.asarray(sides["side_hang"]) != 0)
        if (np.any(nbr <= -2) or np.any(nbr4 <= -2)) and cores is None:
            raise NotImplementedError("subdomain elements on other ranks need the extended mesh of SchwarzShard")
        self.num_nodes_overlap = int(num_nodes_overlap)
        if hasattr(mesh, "conn"):        # multi-tree forest (forest.ForestMesh): corners identified through the tree maps
            core_l, elem_l, faces_l = self._corner_neighbours_forest(mesh)
        elif hanging:
            core_l, elem_l, faces_l = self._corner_neighbours(mesh)
        else:
            core_l, elem_l, faces_l = self._walk_neighbours(ne, nbr)
        core = np.concatenate(core_l)
        elem = np.concatenate(elem_l)
        faces = np.concatenate(faces_l)
        if cores is not None:
            cores = np.asarray(cores, dtype=np.int64)
            rank_of = -np.ones(ne, dtype=np.int64)
            rank_of[cores] = np.arange(cores.size)
            keep = rank_of[core] >= 0
            core, elem, faces = rank_of[core[keep]], elem[keep], faces[keep]        # core = subdomain index from here on
        n_sub = ne if cores is None else int(cores.size)
        self.num_subdomains = n_sub
        self.cores = np.arange(ne) if cores is None else cores
        key = elem if sort_key is None else np.asarray(sort_key)[elem]
        order = np.lexsort((key, core))                      # per subdomain sorted by (tree, quadid) = Morton id (:447-455)
        core, elem, faces = core[order], elem[order], faces[order]
        self.sub_first = np.concatenate([[0], np.cumsum(np.bincount(core, minlength=n_sub))]).astype(np.int32)
        self.sub_core = core.astype(np.int32)
        self.sub_elem = elem.astype(np.int32)
        self.sub_faces = np.ascontiguousarray(faces, dtype=np.int32)
        self.sub_core_faces = np.where(self.sub_faces >= 0, self.sub_faces ^ 1, -1).astype(np.int32)   # get_mirrored_face (:379-385)
        self.num_elements = int(elem.size)
        n1 = mesh.deg[elem].astype(np.int64) + 1
        res = np.where(self.sub_faces >= 0, self.num_nodes_overlap, n1[:, None])
        self.elem_restricted_nodal_size = res.prod(axis=1)
        self.elem_nodal_size = n1 ** 3
        self.nodal_size = int(self.elem_nodal_size.sum())
        self.restricted_nodal_size = int(self.elem_restricted_nodal_size.sum())

    @staticmethod
    def _walk_neighbours(ne, nbr):
        """conforming mesh: the 26 offsets reached by walking face neighbours direction by direction"""
        core_l, elem_l, faces_l = [np.arange(ne)], [np.arange(ne)], [np.full((ne, 3), -1, dtype=np.int32)]
        for oz in (-1, 0, 1):
            for oy in (-1, 0, 1):
                for ox in (-1, 0, 1):
                    off = (ox, oy, oz)
                    if off == (0, 0, 0):
                        continue
                    idx = np.arange(ne)
                    valid = np.ones(ne, dtype=bool)
                    for d in range(3):
                        if off[d] == 0:
                            continue
                        nxt = nbr[idx, 2 * d + (1 if off[d] > 0 else 0)]
                        valid &= nxt >= 0
                        idx = np.where(valid, nxt, 0)
                    # faces of the subdomain element that touch the core: the side facing back (ascending like p8est_edge_faces /
                    # p8est_corner_faces, metadata.c:336-372); an element to the right of the core touches it with its "-" face
                    fc = [2 * d + (0 if off[d] > 0 else 1) for d in range(3) if off[d] != 0]
                    fc = fc + [-1] * (3 - len(fc))
                    sel = np.nonzero(valid)[0]
                    core_l.append(sel)
                    elem_l.append(idx[sel])
                    faces_l.append(np.tile(np.array(fc, dtype=np.int32), (sel.size, 1)))
        return core_l, elem_l, faces_l

    @staticmethod
    def _corner_neighbours(mesh):
        """mesh with hanging faces (HangingBrickMesh: origin / size on the fine grid): p8est_iterate calls the corner callback at
        every CONFORMAL corner, i.e. a point that is a corner of every element touching it (p8est_iterate.h, p4est_iter_corner_t),
        and d4est_solver_schwarz_metadata_corner_callback puts every element of that corner into the subdomain of every other one."""
        ne = mesh.n_elements
        org, size = np.asarray(mesh.org), np.asarray(mesh.size)
        nf = 1 << (mesh.level + 1)
        owner = -np.ones((nf, nf, nf), dtype=np.int64)     # -1: a cell of an element this (extended) mesh does not hold
        for e in range(ne):
            o, sz = org[e], int(size[e])
            owner[o[0]:o[0] + sz, o[1]:o[1] + sz, o[2]:o[2] + sz] = e
        pairs = set()
        seen = set()
        for e in range(ne):
            for c in range(8):
                P = org[e] + size[e] * np.array([c & 1, (c >> 1) & 1, (c >> 2) & 1])
                key = (int(P[0]), int(P[1]), int(P[2]))
                if key in seen:
                    continue
                seen.add(key)
                touching = set()
                for q in range(8):
                    cell = P - np.array([q & 1, (q >> 1) & 1, (q >> 2) & 1])
                    if np.all(cell >= 0) and np.all(cell < nf):
                        touching.add(int(owner[cell[0], cell[1], cell[2]]))
                if -1 in touching:        # not all elements around P are known (beyond the ghost layer): never a corner of an own element
                    continue
                conformal = all(np.all((P == org[t]) | (P == org[t] + size[t])) for t in touching)
                if conformal:
                    for a in touching:
                        for b in touching:
                            if a != b:
                                pairs.add((a, b))
        core = np.array([a for a, _ in pairs] + list(range(ne)), dtype=np.int64)
        elem = np.array([b for _, b in pairs] + list(range(ne)), dtype=np.int64)
        faces = np.full((core.size, 3), -1, dtype=np.int32)
        for k in range(len(pairs)):
            c, e = int(core[k]), int(elem[k])
            fc = []
            for d in range(3):
                if org[e][d] == org[c][d] + size[c]:          # e to the right of the core: touches it with its "-" face
                    fc.append(2 * d)
                elif org[e][d] + size[e] == org[c][d]:
                    fc.append(2 * d + 1)
            assert fc, "two elements of one corner must be separated in at least one direction"
            faces[k, :len(fc)] = fc
        return [core], [elem], [faces]

    @staticmethod
    def _corner_neighbours_forest(mesh):
        """forest.ForestMesh (several trees, inter-tree orientation, optional hanging faces): the corner callback of the reference
        (d4est_solver_schwarz_metadata_corner_callback, src/Solver/d4est_solver_schwarz_metadata.c:186-405) restated on points.
        A corner is conformal when it is a corner of every element that touches it; all elements of a conformal corner enter each
        other's subdomains.  The faces recorded for a subdomain element are in ITS OWN frame (p8est_corner_faces /
        p8est_edge_faces of its corner / edge): the shared face if the two share a face, else the two faces at the shared edge, else
        the three faces at the corner."""
        ne = mesh.n_elements
        mp = mesh.mapping
        key = lambda X: tuple(np.round(X * 2 ** 20).astype(np.int64).tolist())
        corners, mids = {}, set()
        ckey = [[None] * 8 for _ in range(ne)]
        mkey = {}
        lattice = np.array([[a, b, c] for c in (0, 1, 2) for b in (0, 1, 2) for a in (0, 1, 2)], dtype=np.float64) - 1.0   # 27 points
        for e in range(ne):
            X = mp.x(int(mesh.tree[e]), mesh._cell_xi(mesh.org[e], mesh.size[e], lattice))
            for n_, (ref, x) in enumerate(zip(lattice, X)):
                k = key(x)
                nz = int(np.sum(ref == 0.0))
                if nz == 0:
                    c = int(ref[0] > 0) + 2 * int(ref[1] > 0) + 4 * int(ref[2] > 0)
                    corners.setdefault(k, []).append((e, c))
                    ckey[e][c] = k
                elif nz <= 2:
                    mids.add(k)                       # edge midpoints and face centres: a hanging node of a smaller neighbour lands here
                    mkey[(e, n_)] = k
        def edge_mid_key(e, c, d):                    # midpoint of the edge of e through corner c along direction d
            ref = np.array([1.0 if (c >> t) & 1 else -1.0 for t in range(3)])
            ref[d] = 0.0
            n_ = int((ref[0] + 1) + 3 * (ref[1] + 1) + 9 * (ref[2] + 1))
            return mkey[(e, n_)]
        def linked(e, fe, c, fc):                     # face fe of e and face fc of c are the two sides of one (possibly hanging) mesh face
            k = mesh.face_neighbours(int(mesh.elements[e]), fe)
            gc = int(mesh.elements[c])
            if k[0] == "full" or k[0] == "small":
                return k[1] == gc and k[2] == fc
            if k[0] == "big":
                return gc in k[1] and k[2] == fc
            return False
        found = {}
        for P, lst in corners.items():
            if P in mids:
                continue                              # a hanging node of somebody: not a conformal corner
            for (c, kc) in lst:
                for (e, ke) in lst:
                    if e == c:
                        continue
                    fe = [2 * d + ((ke >> d) & 1) for d in range(3)]
                    fc = [2 * d + ((kc >> d) & 1) for d in range(3)]
                    shared_face = [a for a in fe for b in fc if linked(e, a, c, b)]
                    if shared_face:
                        faces = (shared_face[0], -1, -1)
                        rank = 2
                    else:
                        faces, rank = None, 0
                        for de in range(3):           # a shared edge: same far end point, or one is half of the other
                            Qe = ckey[e][ke ^ (1 << de)]
                            for dc in range(3):
                                Qc = ckey[c][kc ^ (1 << dc)]
                                if Qe == Qc or Qe == edge_mid_key(c, kc, dc) or Qc == edge_mid_key(e, ke, de):
                                    faces = tuple(sorted(fe[t] for t in range(3) if t != de)) + (-1,)
                                    rank = 1
                        if faces is None:
                            faces = tuple(fe)
                    old = found.get((c, e))
                    if old is None or rank > old[1]:
                        found[(c, e)] = (faces, rank)
        pairs = sorted(found)
        core = np.array([a for a, _ in pairs] + list(range(ne)), dtype=np.int64)
        elem = np.array([b for _, b in pairs] + list(range(ne)), dtype=np.int64)
        faces = np.full((core.size, 3), -1, dtype=np.int32)
        for k, pr in enumerate(pairs):
            faces[k] = found[pr][0]
        return [core], [elem], [faces]

    def subdomain(self, i):
        a, b = int(self.sub_first[i]), int(self.sub_first[i + 1])
        return self.sub_elem[a:b], self.sub_faces[a:b], self.sub_core_faces[a:b]


def subdomain_sides(mesh, sides, md):
    """Side arrays of the subdomain plan (same keys as BrickMesh.build_sides, geometry arrays shared with the mesh)."""
    ne = mesh.n_elements
    nv = md.num_elements
    # position of mesh element e inside subdomain s: binary search in the sorted keys s * ne + e
    key = md.sub_core.astype(np.int64) * ne + md.sub_elem.astype(np.int64)
    perm = np.argsort(key, kind="stable")
    skey = key[perm]

    def locate(ref):                                   # ref: (nv, k) mesh element references; -> (inside mask, copy index)
        want = md.sub_core.astype(np.int64)[:, None] * ne + np.clip(ref, 0, None)
        pos = np.clip(np.searchsorted(skey, want), 0, nv - 1)
        return (ref >= 0) & (skey[pos] == want), perm[pos]

    nbr_mesh = np.asarray(sides["side_nbr"]).reshape(ne, 6)[md.sub_elem]             # (nv, 6) neighbours of the originals
    inside, loc_c = locate(nbr_mesh)
    # outside neighbours: one ghost "element" per (deg, deg_quad) pair that occurs; its trace is never computed, only sized.
    # A neighbour beyond the mesh's own ghost layer (code <= -2 on an extended mesh) is outside every subdomain.
    dq_pairs = {}
    ghost_deg, ghost_deg_quad = [], []
    mesh_ghost_deg, mesh_ghost_degq = np.asarray(sides["ghost_deg"]), np.asarray(sides["ghost_deg_quad"])

    def ghost_codes(ref):
        od = np.where(ref >= 0, mesh.deg[np.clip(ref, 0, None)], mesh_ghost_deg[np.clip(-(ref + 2), 0, None)] if mesh_ghost_deg.size else 0)
        oq = np.where(ref >= 0, mesh.deg_quad[np.clip(ref, 0, None)], mesh_ghost_degq[np.clip(-(ref + 2), 0, None)] if mesh_ghost_degq.size else 0)
        codes = np.empty(od.size, dtype=np.int32)
        for i, (a, b) in enumerate(zip(od.tolist(), oq.tolist())):
            if (a, b) not in dq_pairs:
                dq_pairs[(a, b)] = len(ghost_deg)
                ghost_deg.append(a)
                ghost_deg_quad.append(b)
            codes[i] = -(dq_pairs[(a, b)] + 2)
        return codes

    side_nbr = np.full((nv, 6), -1, dtype=np.int32)
    side_nbr[inside] = loc_c[inside]
    out = (nbr_mesh != -1) & ~inside
    if out.any():
        side_nbr[out] = ghost_codes(nbr_mesh[out])
    rep = lambda k: np.asarray(sides[k]).reshape(ne, 6)[md.sub_elem].reshape(-1).astype(np.int32)
    vs = dict(sides)
    vs.update(side_nbr=side_nbr.reshape(-1), side_nbr_face=rep("side_nbr_face"), side_reorder=rep("side_reorder"),
              side_mortar_stride=rep("side_mortar_stride"), side_bndry_stride=rep("side_bndry_stride"),
              ghost_deg=np.asarray(ghost_deg, dtype=np.int32), ghost_deg_quad=np.asarray(ghost_deg_quad, dtype=np.int32))
    if "side_hang" in sides and np.any(np.asarray(sides["side_hang"]) != 0):
        # hanging faces: the group / neighbour lists are remapped entry by entry the same way (inside: the copy, outside: a zero ghost)
        n4_mesh = np.asarray(sides["side_nbr4"]).reshape(ne, 24)[md.sub_elem]                       # (nv, 24)
        in4, loc4 = locate(n4_mesh)
        n4 = np.full((nv, 24), -1, dtype=np.int32)
        n4[in4] = loc4[in4]
        out4 = (n4_mesh != -1) & ~in4
        if out4.any():
            n4[out4] = ghost_codes(n4_mesh[out4])
        vs.update(side_hang=rep("side_hang"), side_sub=rep("side_sub"), side_orientation=rep("side_orientation"),
                  side_nbr4=n4.reshape(-1), ghost_deg=np.asarray(ghost_deg, dtype=np.int32),
                  ghost_deg_quad=np.asarray(ghost_deg_quad, dtype=np.int32))
    else:
        for k in ("side_hang", "side_sub", "side_nbr4", "side_orientation"):
            vs.pop(k, None)
    return vs


class Schwarz:
    """d4est_solver_schwarz_t on the device: metadata + subdomain plan + the native smoother handle."""

    def __init__(self, mesh, sides, J_quad, rst_xyz_quad, num_nodes_overlap, subdomain_iter, subdomain_atol, subdomain_rtol,
                 penalty_prefactor=10.0, penalty_fcn=0, stream=None, cores=None, sort_key=None):
        self.lib = capi.load_library()
        self.metadata = md = SchwarzMetadata(mesh, sides, num_nodes_overlap, cores=cores, sort_key=sort_key)
        self.subdomain_iter, self.subdomain_atol, self.subdomain_rtol = int(subdomain_iter), float(subdomain_atol), float(subdomain_rtol)
        e = md.sub_elem
        vdeg, vdegq = mesh.deg[e], mesh.deg_quad[e]
        vstride = np.concatenate([[0], np.cumsum(md.elem_nodal_size)[:-1]])
        if md.nodal_size > 0x7fffffff:
            raise ValueError("field over the subdomains exceeds 32-bit strides")
        self.plan = capi.Plan(vdeg, vdegq, vstride.astype(np.int32), mesh.quad_stride[e], quad_type=mesh.quad_type, stream=stream)
        self.plan.set_geometry(J_quad, rst_xyz_quad)                      # the mesh's own arrays: quad_stride aliases them
        self.plan.set_tuning(8, 1)                                        # all outside faces read ONE zero trace block
        self.plan.set_tuning(12, 0)                                       # no stream mode: the copies of an element share its metric, which is therefore re-read
        self._sub_sides = subdomain_sides(mesh, sides, md)
        self.plan.set_faces(self._sub_sides, penalty_prefactor, penalty_fcn)
        self.plan.set_dirichlet_values(None)                              # the correction has homogeneous boundary data
        self._keep = [capi._iarr(a) for a in (md.sub_first, md.sub_elem, md.sub_faces.reshape(-1), md.sub_core_faces.reshape(-1),
                                              mesh.deg, mesh.nodal_stride)]
        k = self._keep
        self.handle = self.lib.d4est_hip_schwarz_create(self.plan.handle, md.num_subdomains, k[0][1], k[1][1], k[2][1], k[3][1],
                                                        md.num_nodes_overlap, mesh.n_elements, k[4][1], k[5][1])
        self.nodal_size = self.lib.d4est_hip_schwarz_nodal_size(self.handle)
        self.restricted_nodal_size = self.lib.d4est_hip_schwarz_restricted_nodal_size(self.handle)
        assert self.nodal_size == md.nodal_size and self.restricted_nodal_size == md.restricted_nodal_size
        self.local_nodes = mesh.local_nodes

    def set_lhs_element_blocks(self, blocks, mesh):
        """the zeroth-order term of a coarse multigrid level (dense element blocks, consecutive in the MESH's element order) as part of
        the subdomain operator: copy k of mesh element e reads e's block"""
        n3 = (mesh.deg.astype(np.int64) + 1) ** 3
        off = np.concatenate([[0], np.cumsum(n3 * n3)[:-1]])
        self.plan.set_lhs_element_blocks(blocks, None if blocks is None else off[self.metadata.sub_elem])

    def condensed_copies(self):
        """element copies whose operator rows are dense blocks (corner copies of conforming one-degree subdomains); 0 = none"""
        return int(self.lib.d4est_hip_schwarz_condensed_copies(self.handle))

    def restrict_field(self, field, out):
        assert field.numel() == self.local_nodes and out.numel() == self.nodal_size
        self.lib.d4est_hip_schwarz_restrict_field(self.handle, capi._ptr(field), capi._ptr(out))

    def apply_over_subdomains(self, x, out):
        assert x.numel() == self.nodal_size and out.numel() == self.nodal_size
        self.lib.d4est_hip_schwarz_apply_over_subdomains(self.handle, capi._ptr(x), capi._ptr(out))

    def add_correction(self, du, u):
        assert du.numel() == self.nodal_size and u.numel() == self.local_nodes
        self.lib.d4est_hip_schwarz_add_correction(self.handle, capi._ptr(du), capi._ptr(u))

    def iterate(self, u, r):
        """u += Schwarz correction of the residual r (d4est_solver_schwarz_iterate); returns the number of batched CG sweeps"""
        assert u.numel() == self.local_nodes and r.numel() == self.local_nodes
        return self.lib.d4est_hip_schwarz_iterate(self.handle, capi._ptr(u), capi._ptr(r), self.subdomain_iter, self.subdomain_atol,
                                                  self.subdomain_rtol)

    def smooth(self, mesh_plan, u, rhs, r, smoother_iterations):
        """d4est_solver_multigrid_smoother_schwarz: iterations x { r = rhs - A u; iterate(u, r) }, then r = rhs - A u"""
        for t in (u, rhs, r):
            assert t.numel() == self.local_nodes
        self.lib.d4est_hip_schwarz_smooth(self.handle, mesh_plan.handle, capi._ptr(u), capi._ptr(rhs), capi._ptr(r), int(smoother_iterations),
                                          self.subdomain_iter, self.subdomain_atol, self.subdomain_rtol)

    def info(self):
        it = np.zeros(self.metadata.num_subdomains, dtype=np.int32)
        res = np.zeros(self.metadata.num_subdomains)
        self.lib.d4est_hip_schwarz_get_info(self.handle, it.ctypes.data_as(_vp), res.ctypes.data_as(_vp))
        return it, res

    def destroy(self):
        if getattr(self, "handle", None):
            self.lib.d4est_hip_schwarz_destroy(self.handle)
            self.handle = None
        if getattr(self, "plan", None) is not None:
            self.plan.destroy()
            self.plan = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def ghost_layer(level, parts, rank):
    """(own global ids, ghost-layer global ids, needed_by) of one rank of a uniform brick: the ghost layer holds every off-rank element
    that shares a face, an edge or a corner with an own element (P4EST_CONNECT_FULL, src/Mesh/d4est_ghost.c:47); needed_by[peer] =
    own elements that lie in ``peer``'s ghost layer.  Every rank derives both from the global Morton numbering alone."""
    from .mesh import morton_order
    from .parallel import owner_of
    ijk = morton_order(level)
    n = 1 << level
    lookup = np.empty((n, n, n), dtype=np.int64)
    lookup[ijk[:, 0], ijk[:, 1], ijk[:, 2]] = np.arange(ijk.shape[0])
    owner = owner_of(parts, ijk.shape[0])
    first, count = parts[rank]
    own = np.arange(first, first + count, dtype=np.int64)
    ghosts, needed_by = set(), {}
    for oz in (-1, 0, 1):
        for oy in (-1, 0, 1):
            for ox in (-1, 0, 1):
                if (ox, oy, oz) == (0, 0, 0):
                    continue
                c = ijk[own] + np.array([ox, oy, oz])
                ok = np.all((c >= 0) & (c < n), axis=1)
                g = lookup[c[ok, 0], c[ok, 1], c[ok, 2]]
                off_rank = owner[g] != rank
                ghosts.update(g[off_rank].tolist())
                for peer, mine in zip(owner[g[off_rank]].tolist(), own[ok][off_rank].tolist()):
                    needed_by.setdefault(peer, set()).add(mine)
    return own, np.array(sorted(ghosts), dtype=np.int64), needed_by


def ghost_layer_hanging(level, refine, parts, rank):
    """ghost_layer() for a HangingBrickMesh: every off-rank element whose closed box touches an own element's (faces, edges, corners,
    hanging or not -- P4EST_CONNECT_FULL); the conformal-corner test of the subdomain builder then sees every element around an own
    element's corners."""
    from .mesh import HangingBrickMesh
    from .parallel import owner_of
    mg = HangingBrickMesh(level, refine, 1)
    org, size = mg._org_all, mg._size_all
    nf = 1 << (level + 1)
    cell = -np.ones((nf, nf, nf), dtype=np.int64)
    for g in range(mg.global_elements):
        o, sz = org[g], int(size[g])
        cell[o[0]:o[0] + sz, o[1]:o[1] + sz, o[2]:o[2] + sz] = g
    owner = owner_of(parts, mg.global_elements)
    first, count = parts[rank]
    own = np.arange(first, first + count, dtype=np.int64)
    ghosts, needed_by = set(), {}
    for g in own:
        lo = np.maximum(org[g] - 1, 0)
        hi = np.minimum(org[g] + size[g] + 1, nf)
        for t in np.unique(cell[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]).tolist():
            if owner[t] != rank:
                ghosts.add(t)
                needed_by.setdefault(int(owner[t]), set()).add(int(g))
    return own, np.array(sorted(ghosts), dtype=np.int64), needed_by


class SchwarzShard:
    """The smoother on one rank of a sharded brick (``refine=None``: conforming BrickMesh; else a HangingBrickMesh).  The rank's
    *extended mesh* is its own elements followed by the ghost layer; subdomains exist for the own elements only and may contain
    ghost-layer elements, whose geometric factors the rank computes itself (as d4est_solver_schwarz_geometric_data_init does for ghost
    elements) and whose residual arrives by a whole-element exchange; corrections computed for ghost-layer elements travel back to
    their owners and are added there (d4est_solver_schwarz_transfer_ghost_data_and_add_corrections).  One forward and one backward
    point-to-point exchange per iterate."""

    def __init__(self, level, deg_global, parts, rank, mapping, num_nodes_overlap, subdomain_iter, subdomain_atol, subdomain_rtol,
                 transport, device, penalty_prefactor=10.0, penalty_fcn=0, deg_quad_inc=0, quad_type=0, transport_back=None, refine=None):
        import torch
        from .mesh import BrickMesh, HangingBrickMesh
        from .parallel import ElementSchedule, TraceExchange
        if refine is not None:
            own, ghosts, needed_by = ghost_layer_hanging(level, refine, parts, rank)
            self.n_own = int(own.size)
            self.mesh = m = HangingBrickMesh(level, refine, deg_global, deg_quad_inc=deg_quad_inc, quad_type=quad_type,
                                             elements=np.concatenate([own, ghosts]))
        else:
            own, ghosts, needed_by = ghost_layer(level, parts, rank)
            self.n_own = int(own.size)
            self.mesh = m = BrickMesh(level, deg_global, deg_quad_inc=deg_quad_inc, quad_type=quad_type, elements=np.concatenate([own, ghosts]))
        J, rst = m.geometry(mapping)
        sides = m.build_sides(mapping)
        self.schwarz = Schwarz(m, sides, J, rst, num_nodes_overlap, subdomain_iter, subdomain_atol, subdomain_rtol, penalty_prefactor,
                               penalty_fcn, cores=np.arange(self.n_own), sort_key=m.elements)
        self.own_nodes = int(((m.deg[:self.n_own].astype(np.int64) + 1) ** 3).sum())
        sched = ElementSchedule(m, self.n_own, parts, needed_by)
        self.forward = TraceExchange(sched, transport, self.schwarz.plan.copy_blocks, device)
        self.backward = TraceExchange(sched.reversed(), transport_back or transport, self.schwarz.plan.copy_blocks, device)
        self.r_ext = torch.zeros(m.local_nodes, dtype=torch.float64, device=device)
        self.u_ext = torch.zeros(m.local_nodes, dtype=torch.float64, device=device)
        self.back = torch.zeros(m.local_nodes, dtype=torch.float64, device=device)

    # the two halves of an exchange are separate calls so that in-process virtual ranks can be interleaved by the tests
    def begin_residual_exchange(self, r_own):
        self.r_ext[:self.own_nodes].copy_(r_own)
        self.forward.begin(self.r_ext)

    def solve_and_begin_correction_exchange(self):
        self.forward.end(self.r_ext)
        self.u_ext.zero_()
        sweeps = self.schwarz.iterate(self.u_ext, self.r_ext)
        self.backward.begin(self.u_ext)                       # the ghost-layer part of the correction goes back to the owners
        return sweeps

    def end_correction_exchange(self, u_own):
        u_own += self.u_ext[:self.own_nodes]                   # own subdomains first, then the peers' in rank order (fixed order of additions)
        bw = self.backward
        bw.transport.finish(bw._pending)
        bw._pending = None
        for p in bw.s.peers:                                  # several peers may correct the same own element: one accumulation pass each
            _, _, _, ro, rp, rl = bw.idx[p]
            if len(rl) == 0:
                continue
            self.back.zero_()
            bw.copy_blocks(len(rl), bw.recv_buf[p], rp, self.back, ro, rl)
            u_own += self.back[:self.own_nodes]

    def iterate(self, u_own, r_own):
        """one d4est_solver_schwarz_iterate on this rank (collective: every rank calls it)"""
        self.begin_residual_exchange(r_own)
        sweeps = self.solve_and_begin_correction_exchange()
        self.end_correction_exchange(u_own)
        return sweeps
