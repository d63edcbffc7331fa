// One translation unit of libxparcel per (XP_TU_T, XP_TU_MODE): the k_cape_cin instantiations of that data type and
// moist mode (4 parcel modes x profile on/off x dewpoint / specific-humidity input, + the default-options specialisation of
// the CAPE/CIN-only dewpoint-input kernels) and the launcher that picks one.
// Compiled six times by the build (xarray_parcel_amd/_lib.py), e.g. -DXP_TU_T=double -DXP_TU_MODE=0.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "xp_kernels.hpp"

#if !defined(XP_TU_T) || !defined(XP_TU_MODE)
#error "compile with -DXP_TU_T=<float|double> -DXP_TU_MODE=<0|1|2>"
#endif

namespace xp {
namespace {

int cape_block() {             // XP_CAPE_BLOCK: workgroup size for experiments (64 / 128 / 256; default 256)
    static int b = [] { const char *e = getenv("XP_CAPE_BLOCK"); int v = e ? atoi(e) : XP_CAPE_THREADS; return (v == 64 || v == 128 || v == 256 || v == XP_CAPE_THREADS) ? v : XP_CAPE_THREADS; }();
    return b;
}

template <typename T, int PM, int MODE, bool PERSIST> void launch_p(const CapeArgs &a, bool profile, hipStream_t s) {
    const int b = cape_block();
    unsigned nblk = (unsigned)((a.ncol + b - 1) / b);
    if (PERSIST) {                                     // persistent wavefronts: one workgroup's worth of waves per CU
        static const int n_cu = [] { int d = 0; hipDeviceProp_t pr; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
        unsigned cap = (unsigned)(n_cu * (b >= 1024 ? 1 : 1024 / b));
        if (nblk > cap) nblk = cap;
    }
    dim3 gr(nblk), bl(b);
    if (a.hum) {
        if (profile) hipLaunchKernelGGL((k_cape_cin<T, PM, true, MODE, true, false, false, PERSIST>), gr, bl, 0, s, a);
        else hipLaunchKernelGGL((k_cape_cin<T, PM, false, MODE, true, false, false, PERSIST>), gr, bl, 0, s, a);
        return;
    }
    if (profile) {
        if constexpr (MODE == 2) {
            // the lifted index and nothing else of the profile, default options, CAPE / CIN only (the product bundle's
            // parcel passes): the LAZY instantiation (PROFILE + DEF + LEAN, see k_cape_cin)
            const bool cc_only = !a.s.lfc_t && !a.s.el_t && !a.s.lfc_idx && !a.s.el_idx && !a.s.status;
            if (a.prof.nlev_out == 0 && a.prof.li && a.vtc && a.pos_neg && a.log_interp && cc_only) {
                hipLaunchKernelGGL((k_cape_cin<T, PM, true, MODE, false, true, true, PERSIST>), gr, bl, 0, s, a);
                return;
            }
        }
        hipLaunchKernelGGL((k_cape_cin<T, PM, true, MODE, false, false, false, PERSIST>), gr, bl, 0, s, a);
        return;
    }
    // Default-options (DEF) and CAPE/CIN-only (LEAN) specialisations: the reference's default option set as compile-time
    // constants, and -- when the caller wants neither LFC / EL temperatures nor interval indices (the bench, a multi-GPU
    // gather, the product bundle) -- no tracking of them.  The RK4 and lookup-table modes take both; the family kernels
    // sit at the 128-VGPR cap of their 1024-thread workgroups and take only the combination DEF + LEAN, which comes out
    // of the register allocator without a spill for every parcel (121-128 VGPRs; round 2's code base spilled here and
    // always took the generic kernel) and runs 2.5-5 % faster than the generic one (same-box A/B); DEF
    // alone still spills for the searching parcels (40-55 VGPRs) and stays with the generic kernel.
    // tests/test_kernel_resources.py watches the numbers this rule rests on.
    const bool lean = !a.s.lfc_t && !a.s.el_t && !a.s.lfc_idx && !a.s.el_idx && !a.s.status;   // no LFC / EL temperatures, indices or status word wanted
    const bool dflt = a.vtc && a.pos_neg && a.log_interp;                           // the reference's defaults (pf.py:1396, 1293)
    if (dflt && lean) { hipLaunchKernelGGL((k_cape_cin<T, PM, false, MODE, false, true, true, PERSIST>), gr, bl, 0, s, a); return; }
    if constexpr (MODE != 2) {
        if (dflt) { hipLaunchKernelGGL((k_cape_cin<T, PM, false, MODE, false, true, false, PERSIST>), gr, bl, 0, s, a); return; }
    }
    hipLaunchKernelGGL((k_cape_cin<T, PM, false, MODE, false, false, false, PERSIST>), gr, bl, 0, s, a);
}
template <typename T, int PM, int MODE> void launch_t(const CapeArgs &a, bool profile, hipStream_t s) {
    if constexpr (MODE == 2) {
        if (a.persist) { launch_p<T, PM, MODE, true>(a, profile, s); return; }
    }
    launch_p<T, PM, MODE, false>(a, profile, s);
}

}  // namespace

template <> void launch_cape_mode<XP_TU_T, XP_TU_MODE>(const CapeArgs &a, int pm, bool profile, hipStream_t s) {
    switch (pm) {
        case PM_SURFACE: launch_t<XP_TU_T, PM_SURFACE, XP_TU_MODE>(a, profile, s); break;
        case PM_MU: launch_t<XP_TU_T, PM_MU, XP_TU_MODE>(a, profile, s); break;
        case PM_ML: launch_t<XP_TU_T, PM_ML, XP_TU_MODE>(a, profile, s); break;
        default: launch_t<XP_TU_T, PM_EXPLICIT, XP_TU_MODE>(a, profile, s); break;
    }
}

}  // namespace xp
