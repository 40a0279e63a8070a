// Definitions behind launch.hpp; included by the policy_*.hip translation units only.
#pragma once
#include <cstdlib>

#include "launch.hpp"

namespace nttx {

template <class A, bool FWD, bool CONTIG, int K1, int K2, bool XB = false, bool POW2 = false>
hipError_t launch_one(const PassParams& p, const typename A::Ctx& c, unsigned grid, hipStream_t s)
{
    (void)hipGetLastError(); // drop any error an earlier, unrelated call left behind on this thread
#ifdef NTTX_PROFILING
    // NTTX_DEBUG_EXTRA_LDS=<bytes>: unused dynamic LDS per workgroup, to lower occupancy in experiments
    static const unsigned extra_lds = [] {
        const char* e = std::getenv("NTTX_DEBUG_EXTRA_LDS");
        return e ? (unsigned)std::atol(e) : 0u;
    }();
#else
    constexpr unsigned extra_lds = 0;
#endif
    hipLaunchKernelGGL((pass_kernel<A, FWD, CONTIG, K1, K2, XB, POW2>), dim3(grid), dim3(THREADS), extra_lds, s, p, c);
    return hipGetLastError();
}

template <class A, bool FWD>
hipError_t launch_shape(const PassDesc& d, const PassParams& p, const typename A::Ctx& c, unsigned grid, hipStream_t s)
{
    const int key = ((p.flags & PASS_CROSS_BATCH) ? 1000 : 0) + (d.contig ? 100 : 0) + d.k1 * 10 + d.k2;
    if constexpr (A::HAS_POW2_TWIDDLES && half_lds_shape<false, 4, 4, false>()) {
        if (key == 44 && (p.flags & PASS_POW2_TWIDDLES)) // the first four stages' twiddles as shifts
            return launch_one<A, FWD, false, 4, 4, false, true>(p, c, grid, s);
    }
    switch (key) {
    case 1144: return launch_one<A, FWD, true, 4, 4, true>(p, c, grid, s);
    // odd stage counts: plans of the corrected constructor with odd log2 n
    case 110: return launch_one<A, FWD, true, 1, 0>(p, c, grid, s);
    case 130: return launch_one<A, FWD, true, 3, 0>(p, c, grid, s);
    case 141: return launch_one<A, FWD, true, 4, 1>(p, c, grid, s);
    case 143: return launch_one<A, FWD, true, 4, 3>(p, c, grid, s);
    case 10: return launch_one<A, FWD, false, 1, 0>(p, c, grid, s);
    case 30: return launch_one<A, FWD, false, 3, 0>(p, c, grid, s);
    case 41: return launch_one<A, FWD, false, 4, 1>(p, c, grid, s);
    case 43: return launch_one<A, FWD, false, 4, 3>(p, c, grid, s);
    case 120: return launch_one<A, FWD, true, 2, 0>(p, c, grid, s);
    case 140: return launch_one<A, FWD, true, 4, 0>(p, c, grid, s);
    case 142: return launch_one<A, FWD, true, 4, 2>(p, c, grid, s);
    case 144: return launch_one<A, FWD, true, 4, 4>(p, c, grid, s);
    case 20: return launch_one<A, FWD, false, 2, 0>(p, c, grid, s);
    case 40: return launch_one<A, FWD, false, 4, 0>(p, c, grid, s);
    case 42: return launch_one<A, FWD, false, 4, 2>(p, c, grid, s);
    case 44: return launch_one<A, FWD, false, 4, 4>(p, c, grid, s);
    default: return hipErrorInvalidValue;
    }
}

template <class A>
hipError_t launch_pass_policy(bool fwd, const PassDesc& d, const PassParams& p, const typename A::Ctx& c, unsigned grid,
                              hipStream_t s)
{
    return fwd ? launch_shape<A, true>(d, p, c, grid, s) : launch_shape<A, false>(d, p, c, grid, s);
}

template <class A>
hipError_t launch_fused_policy(bool fwd, int kf, const PassParams& p, const typename A::Ctx& c, unsigned grid, hipStream_t s)
{
    (void)hipGetLastError();
    if (fwd && kf == 2)
        hipLaunchKernelGGL((fused_small_kernel<A, true, 2>), dim3(grid), dim3(THREADS), 0, s, p, c);
    else if (fwd)
        hipLaunchKernelGGL((fused_small_kernel<A, true, 4>), dim3(grid), dim3(THREADS), 0, s, p, c);
    else if (kf == 2)
        hipLaunchKernelGGL((fused_small_kernel<A, false, 2>), dim3(grid), dim3(THREADS), 0, s, p, c);
    else
        hipLaunchKernelGGL((fused_small_kernel<A, false, 4>), dim3(grid), dim3(THREADS), 0, s, p, c);
    return hipGetLastError();
}

template <class A>
hipError_t launch_product_mid_policy(const PassParams& pf, const PassParams& pb, const typename A::Ctx& c, unsigned grid,
                                     hipStream_t s)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL((product_mid_kernel<A>), dim3(grid), dim3(THREADS), 0, s, pf, pb, c);
    return hipGetLastError();
}

#define NTTX_INSTANTIATE_POLICY(A)                                                                                   \
    template hipError_t launch_pass_policy<A>(bool, const PassDesc&, const PassParams&, const A::Ctx&, unsigned,      \
                                              hipStream_t);                                                          \
    template hipError_t launch_fused_policy<A>(bool, int, const PassParams&, const A::Ctx&, unsigned, hipStream_t);   \
    template hipError_t launch_product_mid_policy<A>(const PassParams&, const PassParams&, const A::Ctx&, unsigned,   \
                                                     hipStream_t);

} // namespace nttx
