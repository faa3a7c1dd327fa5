// The integer topology tables of the hot path, in ONE place: p8est's face tables (p4est-2.8 src/p8est_connectivity.c:29-63, :145-152)
// and d4est's face re-orientation tables (src/dGMath/d4est_reference.c:3-12).  Every use in the library (d4est_hip_faces.hip,
// d4est_hip_sides.cpp) reads these arrays; d4est_hip_topology_table() hands them out, and tests/test_topology_tables.py compares them
// entry by entry with the originals extracted as data from the reference's own files (tests/golden/p8est_tables.json).
#pragma once

namespace d4est_hip {
namespace topo {

// p8est_face_corners: corners of face f in the face's z-order
constexpr int face_corners[6][4] = {{0, 2, 4, 6}, {1, 3, 5, 7}, {0, 1, 4, 5}, {2, 3, 6, 7}, {0, 1, 2, 3}, {4, 5, 6, 7}};
// p8est_face_dual
constexpr int face_dual[6] = {1, 0, 3, 2, 5, 4};
// p8est_face_permutations: the 8 permutations of a face's four corners that a face connection can produce
constexpr int face_permutations[8][4] = {{0, 1, 2, 3}, {0, 2, 1, 3}, {1, 0, 3, 2}, {1, 3, 0, 2},
                                         {2, 0, 3, 1}, {2, 3, 0, 1}, {3, 1, 2, 0}, {3, 2, 1, 0}};
// p8est_face_permutation_sets: per reference class, the permutation number of each orientation
constexpr int face_permutation_sets[3][4] = {{1, 2, 5, 6}, {0, 3, 4, 7}, {0, 4, 3, 7}};
// p8est_face_permutation_refs: the reference class of a face pair
constexpr int face_permutation_refs[6][6] = {{0, 1, 1, 0, 0, 1}, {2, 0, 0, 1, 1, 0}, {2, 0, 0, 1, 1, 0},
                                             {0, 2, 2, 0, 0, 1}, {0, 2, 2, 0, 0, 1}, {2, 0, 0, 2, 2, 0}};
// p8est_corner_faces: the three faces a corner lies on
constexpr int corner_faces[8][3] = {{0, 2, 4}, {1, 2, 4}, {0, 3, 4}, {1, 3, 4}, {0, 2, 5}, {1, 2, 5}, {0, 3, 5}, {1, 3, 5}};

// d4est_reference.c:3-12 (the reference keeps its own copies of the three p8est tables above, under its own names)
constexpr int d4est_FToF_code[6][6] = {{0, 1, 1, 0, 0, 1}, {2, 0, 0, 1, 1, 0}, {2, 0, 0, 1, 1, 0},
                                       {0, 2, 2, 0, 0, 1}, {0, 2, 2, 0, 0, 1}, {2, 0, 0, 2, 2, 0}};
constexpr int d4est_code_to_perm[3][4] = {{1, 2, 5, 6}, {0, 3, 4, 7}, {0, 4, 3, 7}};
constexpr int d4est_perm_to_order[8][4] = {{0, 1, 2, 3}, {0, 2, 1, 3}, {1, 0, 3, 2}, {1, 3, 0, 2},
                                           {2, 0, 3, 1}, {2, 3, 0, 1}, {3, 1, 2, 0}, {3, 2, 1, 0}};

}  // namespace topo
}  // namespace d4est_hip
