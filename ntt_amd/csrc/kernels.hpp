// Transform kernels for gfx950.
//
// One launch ("pass") applies K = K1 + K2 consecutive butterfly stages to every element of
// the batch.  Before a pass that starts after s0 stages the data of one transform is split
// into 2^s0 independent blocks of B = n / 2^s0 elements; inside a block the K stages act on
// 2^K elements at stride T = B / 2^K, with twiddles that depend only on the block and on the
// position inside that 2^K-point sub-transform (reference index structure: ntt.rs:129-171 for
// the forward Cooley-Tukey order, ntt.rs:185-229 for the backward Gentleman-Sande order).
//
// A workgroup of 256 threads owns a tile of 4096 elements = 4096 / 2^K sub-transforms, and
// every thread keeps 16 elements in registers.  The K stages run as two register rounds
// (K1 stages, exchange through LDS, K2 stages) so a tile crosses HBM once per pass.
//   STRIDED tile: 2^K rows x (4096 / 2^K) adjacent columns, rows T elements apart; every
//                 global access is a run of >= 128 contiguous bytes.  Used while T > 1.
//   CONTIG  tile: 4096 contiguous elements (T = 1, the pass that holds the stride-1 stage);
//                 moved between HBM and LDS with 16-byte-per-lane accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "arith.hpp"

namespace nttx {

constexpr int TILE_LOG = 12;
constexpr int TILE = 1 << TILE_LOG;      // elements per workgroup
constexpr int THREADS = 256;             // 4 waves
constexpr int REGS = TILE / THREADS;     // 16 elements per thread
constexpr int LDS_ELEMS = TILE + TILE / 16;

enum : uint32_t {
    PASS_CONDITION_INPUT = 1u, // first pass of a transform: map arbitrary u64 input to residues
    PASS_SCALE = 2u,           // pass holding stage 1 of the inverse: fold the n^-1 factor in
    PASS_FINISH = 4u,          // last pass of the forward transform: store canonical residues
    PASS_DEBUG_NO_MATH = 8u    // profiling aid (NTTX_DEBUG_NO_MATH=1): skip the butterflies, keep all traffic
};

struct PassParams {
    uint64_t* data;
    const void* tw;           // device twiddle table of this direction, n entries of Arith::TwElem
    uint64_t nblocks_total;   // batch << s0
    uint32_t logB;            // log2 of the block size n >> s0
    uint32_t s0;              // stages that precede this pass in forward order
    uint32_t flags;
};

// Twiddles of K stages for twiddle base P: entry (1 << (u-1)) + b of `w` is
// tw[(P << (u-1)) + b], the reference's powers_psi_bo[m + i] with m = P << (u-1)
// (ntt.rs:139-147 forward, :196-221 backward).
template <class A, int K>
__device__ __forceinline__ void load_twiddles(typename A::Tw (&w)[1 << K], const typename A::TwElem* __restrict__ tw,
                                              uint64_t P, const typename A::Ctx& c)
{
#pragma unroll
    for (int u = 1; u <= K; u++) {
        const typename A::TwElem* row = tw + (P << (u - 1));
#pragma unroll
        for (int b = 0; b < (1 << (u - 1)); b++)
            w[(1 << (u - 1)) + b] = A::load(row, b, c);
    }
}

// K butterfly stages on 2^K registers.  Register i holds sub-transform element i (stride
// handled by the caller); stage u = 1..K pairs i with i + 2^(K-u) and uses twiddle
// w[(1 << (u-1)) + (i >> (K-u+1))].
template <class A, int K, bool FWD>
__device__ __forceinline__ void radix(uint64_t (&x)[1 << K], const typename A::Tw (&w)[1 << K],
                                      const typename A::Ctx& c, bool scale)
{
    constexpr int M = 1 << K;
    if (FWD) {
#pragma unroll
        for (int u = 1; u <= K; u++) {
            const int half = M >> u;
#pragma unroll
            for (int b = 0; b < (1 << (u - 1)); b++) {
#pragma unroll
                for (int i = 0; i < half; i++) {
                    const int lo = b * 2 * half + i;
                    A::ct(x[lo], x[lo + half], w[(1 << (u - 1)) + b], c);
                }
            }
        }
    } else {
#pragma unroll
        for (int u = K; u >= 1; u--) {
            const int half = M >> u;
            if (u == 1 && scale) {
                // last stage of the inverse with the n^-1 factor folded in (ntt.rs:231-235)
#pragma unroll
                for (int i = 0; i < half; i++)
                    A::gs_last(x[i], x[i + half], w[1], c);
            } else {
#pragma unroll
                for (int b = 0; b < (1 << (u - 1)); b++) {
#pragma unroll
                    for (int i = 0; i < half; i++) {
                        const int lo = b * 2 * half + i;
                        A::gs(x[lo], x[lo + half], w[(1 << (u - 1)) + b], c);
                    }
                }
            }
        }
    }
}

// The reference's first sweep accepts any u64 (ntt.rs:125-127).  Residues are the common case,
// so one max over the high words screens all M values and the per-element reduction runs only
// in lanes that hold a candidate.
template <class A, int M>
__device__ __forceinline__ void condition_all(uint64_t (&x)[M], const typename A::Ctx& c)
{
    if constexpr (A::CONDITIONS_INPUT) {
        uint32_t mx = 0;
#pragma unroll
        for (int i = 0; i < M; i++)
            mx = max(mx, hi32(x[i]));
        if (mx >= hi32(A::modulus(c))) {
#pragma unroll
            for (int i = 0; i < M; i++)
                x[i] = A::condition_input(x[i], c);
        }
    }
}

template <bool CONTIG, int K>
__device__ __forceinline__ uint32_t lds_addr(uint32_t sigma, uint32_t r)
{
    if (CONTIG) {
        const uint32_t f = (sigma << K) + r;
        return f + (f >> 4);
    } else {
        const uint32_t f = (r << (TILE_LOG - K)) + sigma;
        return f + ((f >> 8) << 4);
    }
}

// Geometry of one tile, shared by all rounds of a pass.
template <int K>
struct TileGeom {
    uint32_t logB, logT, logCw, lognblk, logtpb, s0;
    uint64_t tile, nblocks_total;

    __device__ __forceinline__ TileGeom(const PassParams& p, uint64_t tile_)
    {
        constexpr uint32_t logS = TILE_LOG - K;
        logB = p.logB;
        s0 = p.s0;
        logT = logB - K;
        logCw = logT < logS ? logT : logS;
        lognblk = logS - logCw;
        logtpb = logT - logCw;
        tile = tile_;
        nblocks_total = p.nblocks_total;
    }
    __device__ __forceinline__ uint64_t block_of(uint32_t sigma) const
    {
        return ((tile >> logtpb) << lognblk) + (sigma >> logCw);
    }
    __device__ __forceinline__ uint64_t column_of(uint32_t sigma) const
    {
        return ((tile & ((1ull << logtpb) - 1)) << logCw) + (sigma & ((1u << logCw) - 1));
    }
    // index of element r of sub-transform sigma in the batch array
    __device__ __forceinline__ uint64_t gidx(uint32_t sigma, uint32_t r) const
    {
        return (block_of(sigma) << logB) + ((uint64_t)r << logT) + column_of(sigma);
    }
    __device__ __forceinline__ bool valid(uint32_t sigma) const { return block_of(sigma) < nblocks_total; }
    // twiddle base of the block: 2^s0 + (block index inside its transform)
    __device__ __forceinline__ uint64_t twiddle_base(uint32_t sigma) const
    {
        return (1ull << s0) + (block_of(sigma) & ((1ull << s0) - 1));
    }
};

// One register round of a pass for one group of 2^KR registers.  ROUND_A covers the K1 stages
// with the larger strides (elements rlow + (i << K2)), round B the K2 stages with the smaller
// ones (elements (rhigh << K2) + i).
template <class A, bool CONTIG, int K1, int K2, bool ROUND_A>
struct Round {
    static constexpr int K = K1 + K2;
    static constexpr int KR = ROUND_A ? K1 : K2; // stages in this round
    static constexpr int KO = ROUND_A ? K2 : K1; // stages in the other round
    static constexpr int M = 1 << KR;
    static constexpr int GROUPS = REGS / M;
    static constexpr uint32_t logS = TILE_LOG - K;
    static constexpr uint32_t rstep_log = ROUND_A ? K2 : 0;

    uint32_t sigma, r0;
    bool ok;
    uint64_t P; // twiddle base of this group's KR stages

    __device__ __forceinline__ Round(const TileGeom<K>& g, int gamma)
    {
        const uint32_t slot = gamma * THREADS + threadIdx.x;
        uint32_t rpart;
        if (CONTIG) {
            rpart = slot & ((1u << KO) - 1);
            sigma = slot >> KO;
        } else {
            sigma = slot & ((1u << logS) - 1);
            rpart = slot >> logS;
        }
        r0 = ROUND_A ? rpart : (rpart << K2);
        ok = g.valid(sigma);
        const uint64_t base = g.twiddle_base(sigma);
        P = ok ? (ROUND_A ? base : ((base << K1) + rpart)) : 1;
    }

    __device__ __forceinline__ void load_global(uint64_t (&x)[M], const PassParams& p, const TileGeom<K>& g,
                                                const typename A::Ctx& c) const
    {
        if (ok) {
            const uint64_t* src = p.data + g.gidx(sigma, r0);
            const uint64_t step = 1ull << (g.logT + rstep_log);
#pragma unroll
            for (int i = 0; i < M; i++)
                x[i] = src[i * step];
        } else {
#pragma unroll
            for (int i = 0; i < M; i++)
                x[i] = 0;
        }
        if (p.flags & PASS_CONDITION_INPUT)
            condition_all<A, M>(x, c);
    }
    __device__ __forceinline__ void store_global(const uint64_t (&x)[M], const PassParams& p,
                                                 const TileGeom<K>& g) const
    {
        if (ok) {
            uint64_t* dst = p.data + g.gidx(sigma, r0);
            const uint64_t step = 1ull << (g.logT + rstep_log);
#pragma unroll
            for (int i = 0; i < M; i++)
                dst[i * step] = x[i];
        }
    }
    __device__ __forceinline__ void load_lds(uint64_t (&x)[M], const uint64_t* lds) const
    {
#pragma unroll
        for (int i = 0; i < M; i++)
            x[i] = lds[lds_addr<CONTIG, K>(sigma, r0 + (i << rstep_log))];
    }
    __device__ __forceinline__ void store_lds(const uint64_t (&x)[M], uint64_t* lds) const
    {
#pragma unroll
        for (int i = 0; i < M; i++)
            lds[lds_addr<CONTIG, K>(sigma, r0 + (i << rstep_log))] = x[i];
    }
};

template <class A, int M>
__device__ __forceinline__ void finish_all(uint64_t (&x)[M], const typename A::Ctx& c)
{
#pragma unroll
    for (int i = 0; i < M; i++)
        x[i] = A::finish(x[i], c);
}

// Generic round: every group loads its twiddles, transforms and stores.  SRC_G / DST_G: read
// from / write to HBM directly instead of LDS.
template <class A, bool FWD, bool CONTIG, int K1, int K2, bool ROUND_A, bool SRC_G, bool DST_G>
__device__ __forceinline__ void do_round(const PassParams& p, const TileGeom<K1 + K2>& g, const typename A::Ctx& c,
                                         uint64_t* lds)
{
    using R = Round<A, CONTIG, K1, K2, ROUND_A>;
    constexpr bool LAST_OF_PASS = ROUND_A ? (K2 == 0) : true; // this round runs the last forward stage of the pass
    const typename A::TwElem* tw = static_cast<const typename A::TwElem*>(p.tw);
#pragma unroll
    for (int gamma = 0; gamma < R::GROUPS; gamma++) {
        const R r(g, gamma);
        typename A::Tw w[R::M];
        uint64_t x[R::M];
        load_twiddles<A, R::KR>(w, tw, r.P, c);
        if (SRC_G)
            r.load_global(x, p, g, c);
        else
            r.load_lds(x, lds);
        if (!(p.flags & PASS_DEBUG_NO_MATH))
            radix<A, R::KR, FWD>(x, w, c, ROUND_A && (p.flags & PASS_SCALE));
        if (FWD && LAST_OF_PASS && (p.flags & PASS_FINISH))
            finish_all<A, R::M>(x, c);
        if (DST_G)
            r.store_global(x, p, g);
        else
            r.store_lds(x, lds);
    }
}

// Copies a CONTIG tile between HBM and LDS, 16 bytes per lane.
template <class A, int K, bool TO_LDS>
__device__ __forceinline__ void copy_contig(const PassParams& p, const TileGeom<K>& g, const typename A::Ctx& c,
                                            uint64_t* lds)
{
    const uint64_t total = p.nblocks_total << p.logB;
    const uint64_t base = g.tile << TILE_LOG;
    if (TO_LDS) {
        uint64_t x[REGS];
#pragma unroll
        for (int i = 0; i < REGS / 2; i++) {
            const uint32_t f = 2 * (i * THREADS + threadIdx.x);
            ulonglong2 v = make_ulonglong2(0, 0);
            if (base + f < total) // total is a multiple of 4 (n >= 4), f even
                v = *reinterpret_cast<const ulonglong2*>(p.data + base + f);
            x[2 * i] = v.x;
            x[2 * i + 1] = v.y;
        }
        if (p.flags & PASS_CONDITION_INPUT)
            condition_all<A, REGS>(x, c);
#pragma unroll
        for (int i = 0; i < REGS / 2; i++) {
            const uint32_t f = 2 * (i * THREADS + threadIdx.x);
            const uint32_t a0 = f + (f >> 4); // f is even, so f and f+1 share a padding group
            lds[a0] = x[2 * i];
            lds[a0 + 1] = x[2 * i + 1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < REGS / 2; i++) {
            const uint32_t f = 2 * (i * THREADS + threadIdx.x);
            const uint32_t a0 = f + (f >> 4);
            if (base + f < total) {
                ulonglong2 v;
                v.x = lds[a0];
                v.y = lds[a0 + 1];
                *reinterpret_cast<ulonglong2*>(p.data + base + f) = v;
            }
        }
    }
}

// The 8-stage pass (two radix-16 rounds, one group per thread), written out so that the
// twiddles of the second round are requested before the workgroup barrier and their latency
// overlaps the LDS exchange.  In a CONTIG tile the round with 128-byte runs (round A) talks to
// HBM directly; the other side goes through LDS for 16-byte-per-lane coalescing.
template <class A, bool FWD, bool CONTIG>
__device__ __forceinline__ void pass_8(const PassParams& p, const TileGeom<8>& g, const typename A::Ctx& c,
                                       uint64_t* lds)
{
    using RA = Round<A, CONTIG, 4, 4, true>;
    using RB = Round<A, CONTIG, 4, 4, false>;
    const typename A::TwElem* tw = static_cast<const typename A::TwElem*>(p.tw);
    const RA ra(g, 0);
    const RB rb(g, 0);
    typename A::Tw w[16];
    uint64_t x[16];
    const bool scale = p.flags & PASS_SCALE;
    const bool math = !(p.flags & PASS_DEBUG_NO_MATH);

    if (FWD) {
        load_twiddles<A, 4>(w, tw, ra.P, c);
        ra.load_global(x, p, g, c);
        if (math)
            radix<A, 4, true>(x, w, c, false);
        ra.store_lds(x, lds);
        load_twiddles<A, 4>(w, tw, rb.P, c);
        __syncthreads();
        rb.load_lds(x, lds);
        if (math)
            radix<A, 4, true>(x, w, c, false);
        if (p.flags & PASS_FINISH)
            finish_all<A, 16>(x, c);
        if (CONTIG) {
            rb.store_lds(x, lds);
            __syncthreads();
            copy_contig<A, 8, false>(p, g, c, lds);
        } else {
            rb.store_global(x, p, g);
        }
    } else {
        load_twiddles<A, 4>(w, tw, rb.P, c);
        if (CONTIG) {
            copy_contig<A, 8, true>(p, g, c, lds);
            __syncthreads();
            rb.load_lds(x, lds);
        } else {
            rb.load_global(x, p, g, c);
        }
        if (math)
            radix<A, 4, false>(x, w, c, false);
        rb.store_lds(x, lds);
        load_twiddles<A, 4>(w, tw, ra.P, c);
        __syncthreads();
        ra.load_lds(x, lds);
        if (math)
            radix<A, 4, false>(x, w, c, scale);
        ra.store_global(x, p, g);
    }
}

template <class A, bool FWD, bool CONTIG, int K1, int K2>
__global__ __launch_bounds__(THREADS, 4) void pass_kernel(PassParams p, typename A::Ctx c)
{
    constexpr int K = K1 + K2;
    __shared__ __attribute__((aligned(16))) uint64_t lds[LDS_ELEMS];
    const TileGeom<K> g(p, blockIdx.x);

    if constexpr (K1 == 4 && K2 == 4) {
        pass_8<A, FWD, CONTIG>(p, g, c, lds);
    } else if constexpr (CONTIG) {
        copy_contig<A, K, true>(p, g, c, lds);
        __syncthreads();
        if constexpr (FWD) {
            do_round<A, FWD, CONTIG, K1, K2, true, false, false>(p, g, c, lds);
            if constexpr (K2 > 0) {
                __syncthreads();
                do_round<A, FWD, CONTIG, K1, K2, false, false, false>(p, g, c, lds);
            }
        } else {
            if constexpr (K2 > 0) {
                do_round<A, FWD, CONTIG, K1, K2, false, false, false>(p, g, c, lds);
                __syncthreads();
            }
            do_round<A, FWD, CONTIG, K1, K2, true, false, false>(p, g, c, lds);
        }
        __syncthreads();
        copy_contig<A, K, false>(p, g, c, lds);
    } else {
        if constexpr (K2 == 0) {
            do_round<A, FWD, CONTIG, K1, K2, true, true, true>(p, g, c, lds);
        } else if constexpr (FWD) {
            do_round<A, FWD, CONTIG, K1, K2, true, true, false>(p, g, c, lds);
            __syncthreads();
            do_round<A, FWD, CONTIG, K1, K2, false, false, true>(p, g, c, lds);
        } else {
            do_round<A, FWD, CONTIG, K1, K2, false, true, false>(p, g, c, lds);
            __syncthreads();
            do_round<A, FWD, CONTIG, K1, K2, true, false, true>(p, g, c, lds);
        }
    }
}

} // namespace nttx
