// sweep_packed16_3p.hip — the multi-allelic kernels of a three-plane packed matrix (alleles 4..7), 16 lanes per row
// (see sweep_launch.inc); built as its own object so the routes compile in parallel.
#define FMH_ROUTE_FN launch_sweep_packed16_3p
#define FMH_ROUTE_MM 3  // fmh::kMaskPacked
#define FMH_ROUTE_LPR 16
#define FMH_ROUTE_NPL 3
#include "sweep_launch.inc"
