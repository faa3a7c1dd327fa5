// Second translation unit of the multi-wave whole-operator kernel: the instantiations N = 13 ... 16 (d4est_hip_direct_mw.hip holds the
// kernel and N = 9 ... 12; two units halve the longest compile of the build).
#define D4EST_HIP_MW_PART 1
#include "d4est_hip_direct_mw.hip"
