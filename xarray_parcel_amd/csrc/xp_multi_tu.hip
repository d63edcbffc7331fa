// One translation unit of libxparcel per (XP_TU_T, XP_MULTI_NP): the fused several-parcels-in-one-pass kernel
// (xp_multi.hpp) for that data type and number of parcels, and its launcher.  The workgroup size and the number of LDS slot
// fields per chain come with the compile flags (xarray_parcel_amd/_lib.py): what has to fit the CU's 160 KB of LDS is
// 58.6 KB of tables + NP x XP_SLOT_FIELDS x XP_CAPE_THREADS x 8 B of slots.
#include <hip/hip_runtime.h>

#include "xp_multi.hpp"

#if !defined(XP_TU_T) || !defined(XP_MULTI_NP)
#error "compile with -DXP_TU_T=<float|double> -DXP_MULTI_NP=<1|2|3> -DXP_CAPE_THREADS=... -DXP_SLOT_FIELDS=..."
#endif

namespace xp {

static_assert((LDS_TAB + FAM_SIZE + XP_MULTI_NP * SLOT_FIELDS * SLOT_STRIDE) * 8 + 64 <= 160 * 1024, "LDS budget of one workgroup per CU");

template <> void launch_cape_multi<XP_TU_T, XP_MULTI_NP>(const MultiArgs &a, hipStream_t s) {
    if (a.base.ncol == 0) return;
    const int b = XP_CAPE_THREADS;
    unsigned nblk = (unsigned)((a.base.ncol + b - 1) / b);
    if (a.base.persist) {                                  // persistent wavefronts: one workgroup per CU
        static const int n_cu = [] { int d = 0; hipDeviceProp_t pr; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
        if (nblk > (unsigned)n_cu) nblk = (unsigned)n_cu;
        hipLaunchKernelGGL((k_cape_cin_multi<XP_TU_T, XP_MULTI_NP, true>), dim3(nblk), dim3(b), 0, s, a);
    } else {
        hipLaunchKernelGGL((k_cape_cin_multi<XP_TU_T, XP_MULTI_NP, false>), dim3(nblk), dim3(b), 0, s, a);
    }
}

}  // namespace xp
