// Flat side list of a 2:1 balanced forest, built on the host in C++ from nothing but the p8est connectivity and the list of
// quadrants -- SURVEY.md section 8f rank 1: the once-per-mesh pass that replaces the reference's serial p4est_iterate face walk
// (src/Mesh/d4est_mortars.c:601-840) as the producer of the arrays d4est_hip_plan_set_faces / _set_hanging take.  A d4est build has
// p4est_iterate and records the same numbers from its callback; this entry is for hosts without p4est
// (and is what the plain-C multi-tree probe tests/c/forest_probe.c uses).
//
// Neighbours across tree faces follow p4est: p4est_expand_face_transform + p4est_quadrant_transform_face
// (p4est-2.8 src/p4est_connectivity.c:2877-2944, src/p4est_bits.c:1838-1925, restated); hanging quadrants are reported in the
// z-order of their own side's face; the (+) elements of a big side come in (-) order (d4est_element_data_reorient_f_p_elements_to_f_m_order,
// src/Mesh/d4est_element_data.c:130-150).  Mortar strides are assigned in side order; the local small sides of a hanging face share
// the block of the group's first local member (src/Mesh/d4est_mesh.c:956-962).
#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "d4est_hip_internal.h"
#include "d4est_hip_topology.h"

namespace {

using Key = std::array<int, 5>;   // tree, x, y, z, size

void expand_face_transform(int iface, int target_face, int o, int ft[9]) {
  const auto& refs = d4est_hip::topo::face_permutation_refs;
  ft[0] = iface < 2 ? 1 : 0;
  ft[1] = iface < 4 ? 2 : 1;
  ft[2] = iface / 2;
  int rev = refs[0][iface] ^ refs[0][target_face] ^ ((o == 0 || o == 3) ? 1 : 0);
  ft[3 + rev] = target_face < 2 ? 1 : 0;
  ft[3 + !rev] = target_face < 4 ? 2 : 1;
  ft[5] = target_face / 2;
  rev = (refs[iface][target_face] == 1);
  ft[6 + rev] = o & 1;
  ft[6 + !rev] = o >> 1;
  ft[8] = 2 * (iface & 1) + (target_face & 1);
}

void transform_quadrant(const int q[3], int h, int root, const int ft[9], int r[3]) {
  const int mh = -h, Rmh = root + mh, tRmh = root + Rmh;
  r[ft[3]] = !ft[6] ? q[ft[0]] : Rmh - q[ft[0]];
  r[ft[4]] = !ft[7] ? q[ft[1]] : Rmh - q[ft[1]];
  const int a = q[ft[2]];
  switch (ft[8]) {
    case 0: r[ft[5]] = mh - a; break;
    case 1: r[ft[5]] = a + root; break;
    case 2: r[ft[5]] = a - root; break;
    default: r[ft[5]] = tRmh - a; break;
  }
}

}  // namespace

extern "C" int d4est_hip_build_sides(int n_trees, const int* tree_to_tree, const int* tree_to_face, int root_len, int n_local,
                                     const int* tree, const int* q, const int* dq, const int* deg, const int* deg_quad, int n_ghost,
                                     const int* ghost_tree, const int* ghost_q, const int* ghost_dq, const int* ghost_deg_quad,
                                     int* side_nbr, int* side_nbr_face, int* side_reorder, int* side_orientation, int* side_hang,
                                     int* side_sub, int* side_nbr4, int* side_mortar_stride, int* side_bndry_stride,
                                     int* total_mortar_nodes, int* total_bndry_nodes) {
  if (n_trees < 1 || !tree_to_tree || !tree_to_face || root_len < 1 || n_local < 0 || n_ghost < 0) D4EST_HIP_ABORT("build_sides: bad arguments");
  std::map<Key, int> cell;   // -> element reference: local id, or ghost code -(g + 2)
  for (int e = 0; e < n_local; ++e) cell[{tree[e], q[3 * e], q[3 * e + 1], q[3 * e + 2], dq[e]}] = e;
  for (int g = 0; g < n_ghost; ++g) cell[{ghost_tree[g], ghost_q[3 * g], ghost_q[3 * g + 1], ghost_q[3 * g + 2], ghost_dq[g]}] = -(g + 2);
  auto degq_of = [&](int ref) { return ref >= 0 ? deg_quad[ref] : ghost_deg_quad[-(ref + 2)]; };
  const int kNone = 0x7fffffff;
  // the cell (corner c, side h) of tree t lying across face f of the tree when outside; returns tree', c', f', orientation
  struct Across { bool boundary; int t, c[3], fp, o; };
  auto across = [&](int t, const int c[3], int h, int f) {
    Across r{};
    const int d = f >> 1;
    if (c[d] >= 0 && c[d] < root_len) { r.boundary = false; r.t = t; r.c[0] = c[0]; r.c[1] = c[1]; r.c[2] = c[2]; r.fp = f ^ 1; r.o = 0; return r; }
    const int tp = tree_to_tree[6 * t + f], code = tree_to_face[6 * t + f];
    if (tp == t && code == f) { r.boundary = true; return r; }
    r.boundary = false; r.t = tp; r.fp = code % 6; r.o = code / 6;
    int ft[9];
    expand_face_transform(f, r.fp, r.o, ft);
    transform_quadrant(c, h, root_len, ft, r.c);
    return r;
  };
  auto find = [&](int t, const int c[3], int h) {
    auto it = cell.find({t, c[0], c[1], c[2], h});
    return it == cell.end() ? kNone : it->second;
  };
  int total = 0, total_b = 0, any_hanging = 0;
  std::map<std::array<int, 5>, int> group_block;   // (4 members of a hanging group, face) -> stride of the shared block
  for (int e = 0; e < n_local; ++e) {
    const int t = tree[e], h = dq[e];
    const int* c0 = &q[3 * e];
    for (int f = 0; f < 6; ++f) {
      const int s = 6 * e + f, d = f >> 1, a0 = d == 0 ? 1 : 0, a1 = d == 2 ? 1 : 2;
      side_nbr[s] = -1; side_nbr_face[s] = f ^ 1; side_reorder[s] = 0; side_orientation[s] = 0; side_hang[s] = 0; side_sub[s] = 0;
      for (int i = 0; i < 4; ++i) side_nbr4[4 * s + i] = -1;
      side_bndry_stride[s] = total_b;
      int out[3] = {c0[0], c0[1], c0[2]};
      out[d] = (f & 1) ? c0[d] + h : c0[d] - h;
      const Across nb = across(t, out, h, f);
      if (nb.boundary) {
        side_mortar_stride[s] = total;
        total += (deg_quad[e] + 1) * (deg_quad[e] + 1);
        total_b += (deg[e] + 1) * (deg[e] + 1);
        continue;
      }
      side_nbr_face[s] = nb.fp;
      side_orientation[s] = nb.o;
      side_reorder[s] = d4est_hip::face_reorder_code(f, nb.fp, nb.o);
      int ref = find(nb.t, nb.c, h);
      if (ref != kNone) {                       // same size
        side_nbr[s] = ref;
        side_nbr4[4 * s] = ref;
        side_mortar_stride[s] = total;
        const int pq = std::max(deg_quad[e], degq_of(ref));
        total += (pq + 1) * (pq + 1);
        continue;
      }
      // the neighbour is bigger: the cell of side 2h that contains nb.c
      const int pc[3] = {nb.c[0] & ~(2 * h - 1), nb.c[1] & ~(2 * h - 1), nb.c[2] & ~(2 * h - 1)};
      ref = find(nb.t, pc, 2 * h);
      if (ref != kNone) {
        any_hanging = 1;
        side_hang[s] = 2;
        side_nbr[s] = ref;
        side_sub[s] = ((c0[a0] / h) & 1) + 2 * ((c0[a1] / h) & 1);
        std::array<int, 5> key{};
        for (int i = 0; i < 4; ++i) {
          int gc[3] = {c0[0], c0[1], c0[2]};
          gc[a0] = (c0[a0] & ~(2 * h - 1)) + (i & 1) * h;
          gc[a1] = (c0[a1] & ~(2 * h - 1)) + (i >> 1) * h;
          const int m = find(t, gc, h);
          if (m == kNone) D4EST_HIP_ABORT("build_sides: element %d face %d: member %d of its hanging group is not in the mesh (2:1 balance / ghost layer)", e, f, i);
          side_nbr4[4 * s + i] = m;
          key[i] = m;
        }
        key[4] = f;
        auto it = group_block.find(key);
        if (it != group_block.end()) { side_mortar_stride[s] = it->second; continue; }
        group_block[key] = total;
        side_mortar_stride[s] = total;
        for (int i = 0; i < 4; ++i) { const int pq = std::max(degq_of(side_nbr4[4 * s + i]), degq_of(ref)); total += (pq + 1) * (pq + 1); }
        continue;
      }
      // the neighbours are smaller: four half-size cells, in the (-) side's face z-order
      if (h < 2) D4EST_HIP_ABORT("build_sides: element %d face %d has no neighbour of a compatible size", e, f);
      any_hanging = 1;
      side_hang[s] = 1;
      side_mortar_stride[s] = total;
      for (int i = 0; i < 4; ++i) {
        int oc[3] = {c0[0], c0[1], c0[2]};
        oc[d] = (f & 1) ? c0[d] + h : c0[d] - h / 2;
        oc[a0] += (i & 1) * (h / 2);
        oc[a1] += (i >> 1) * (h / 2);
        const Across sb = across(t, oc, h / 2, f);
        const int m = sb.boundary ? kNone : find(sb.t, sb.c, h / 2);
        if (m == kNone) D4EST_HIP_ABORT("build_sides: element %d face %d: the mesh is not 2:1 balanced (or the ghost layer is incomplete)", e, f);
        side_nbr4[4 * s + i] = m;
        const int pq = std::max(deg_quad[e], degq_of(m));
        total += (pq + 1) * (pq + 1);
      }
      side_nbr[s] = side_nbr4[4 * s];
    }
  }
  if (total_mortar_nodes) *total_mortar_nodes = total;
  if (total_bndry_nodes) *total_bndry_nodes = total_b;
  return any_hanging;
}

// the integer tables of d4est_hip_topology.h by id (out == NULL: only the entry count): 0 p8est_face_corners [6][4], 1 p8est_face_dual [6],
// 2 p8est_face_permutations [8][4], 3 p8est_face_permutation_sets [3][4], 4 p8est_face_permutation_refs [6][6], 5 p8est_corner_faces
// [8][3], 10 / 11 / 12 d4est_reference_p8est_FToF_code [6][6] / _code_to_perm [3][4] / _perm_to_order [8][4]
extern "C" int d4est_hip_topology_table(int id, int* out) {
  using namespace d4est_hip::topo;
  const int* src = nullptr;
  int n = 0;
  switch (id) {
    case 0: src = &face_corners[0][0]; n = 24; break;
    case 1: src = &face_dual[0]; n = 6; break;
    case 2: src = &face_permutations[0][0]; n = 32; break;
    case 3: src = &face_permutation_sets[0][0]; n = 12; break;
    case 4: src = &face_permutation_refs[0][0]; n = 36; break;
    case 5: src = &corner_faces[0][0]; n = 24; break;
    case 10: src = &d4est_FToF_code[0][0]; n = 36; break;
    case 11: src = &d4est_code_to_perm[0][0]; n = 12; break;
    case 12: src = &d4est_perm_to_order[0][0]; n = 32; break;
    default: return -1;
  }
  if (out) for (int i = 0; i < n; ++i) out[i] = src[i];
  return n;
}
