// sweep_global.hip — kernel instantiations of one mask route (see sweep_launch.inc); built as its own object so the routes
// compile in parallel.
#define FMH_ROUTE_FN launch_sweep_global
#define FMH_ROUTE_MM 1  // fmh::kMask* (sweep_kernels.hpp): 0 bytes in LDS, 1 bytes in global memory, 2 bits in LDS, 3 packed matrix
#define FMH_ROUTE_LPR 16
#include "sweep_launch.inc"
