// Kernel launchers, one set per arithmetic policy.  Each policy's kernels are compiled in a
// translation unit of their own (policy_*.hip, which instantiate launch_impl.hpp), so the library
// builds in parallel; nttx.hip sees only these declarations.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace nttx {

struct PassDesc {
    int k1, k2;
    bool contig;
    uint32_t s0;
    int fused = 0; // 2 or 4: the stages s0 - fused + 1 .. s0 run in the same launch (fused_small_kernel)
};

// Launches pass `d` (direction fwd) over `grid` tiles; PASS_CROSS_BATCH in p.flags selects the
// cross-batch form of the 8-stage CONTIG pass.
template <class A>
hipError_t launch_pass_policy(bool fwd, const PassDesc& d, const PassParams& p, const typename A::Ctx& c, unsigned grid,
                              hipStream_t s);
// Launches fused_small_kernel: the last 8 + kf stages (kf = 2 or 4) in one launch.
template <class A>
hipError_t launch_fused_policy(bool fwd, int kf, const PassParams& p, const typename A::Ctx& c, unsigned grid,
                               hipStream_t s);

// Launches product_mid_kernel over `grid` cross-batch tiles: last forward pass of both operands, product,
// first backward pass (pf / pb: the forward / backward pass parameters).
template <class A>
hipError_t launch_product_mid_policy(const PassParams& pf, const PassParams& pb, const typename A::Ctx& c, unsigned grid,
                                     hipStream_t s);

#define NTTX_DECLARE_POLICY(A)                                                                                        \
    extern template hipError_t launch_pass_policy<A>(bool, const PassDesc&, const PassParams&, const A::Ctx&, unsigned, \
                                                     hipStream_t);                                                    \
    extern template hipError_t launch_fused_policy<A>(bool, int, const PassParams&, const A::Ctx&, unsigned, hipStream_t); \
    extern template hipError_t launch_product_mid_policy<A>(const PassParams&, const PassParams&, const A::Ctx&, unsigned, \
                                                            hipStream_t);
NTTX_DECLARE_POLICY(Shoup61)
NTTX_DECLARE_POLICY(Shoup61Near)
NTTX_DECLARE_POLICY(Mont<false>)
NTTX_DECLARE_POLICY(Mont<true>)
NTTX_DECLARE_POLICY(Goldilocks)
NTTX_DECLARE_POLICY(GoldilocksCanonical)
#undef NTTX_DECLARE_POLICY

} // namespace nttx
