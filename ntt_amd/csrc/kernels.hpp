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
//                 moved between HBM and LDS with 16-byte-per-lane accesses.  For batches of 16
//                 or more transforms the 8-stage CONTIG pass takes cross-batch tiles instead:
//                 the same 256-element block of 16 consecutive transforms (see pass_8).
// Global memory is addressed through buffer resources (scalar base + 32-bit lane offset) so
// that element addresses cost no vector instructions: the passes are bound by VALU issue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "arith.hpp"

namespace nttx {

constexpr int TILE_LOG = 12;
constexpr int TILE = 1 << TILE_LOG;      // elements per workgroup
constexpr int THREADS = 256;             // 4 waves
constexpr int REGS = TILE / THREADS;     // 16 elements per thread
constexpr int LDS_ELEMS = TILE + TILE / 16;

// The PASS_DEBUG_* switches skip work or redirect traffic and so produce wrong results on purpose.
// They are compiled in only with -DNTTX_PROFILING (libnttx_prof.so, `make prof`); in the shipped
// library NTTX_PROF() folds every test of them to false and the code behind them disappears.
#ifdef NTTX_PROFILING
#define NTTX_PROF(cond) (cond)
#else
#define NTTX_PROF(cond) false
#endif

enum : uint32_t {
    PASS_CONDITION_INPUT = 1u, // first pass of a transform: map arbitrary u64 input to residues
    PASS_SCALE = 2u,           // pass holding stage 1 of the inverse: fold the n^-1 factor in
    PASS_FINISH = 4u,          // last pass of the forward transform: store canonical residues
    // the four PASS_DEBUG_* flags act in profiling builds only (NTTX_PROF above)
    PASS_DEBUG_NO_MATH = 8u,   // profiling aid (NTTX_DEBUG_NO_MATH=1): skip the butterflies, keep all traffic
    PASS_DEBUG_NO_TW = 16u,    // profiling aid (NTTX_DEBUG_NO_TW=1, with NO_MATH): skip the twiddle loads too
    PASS_DEBUG_NO_MATH_A = 32u, // profiling aids (NTTX_DEBUG_NO_MATH=A / B): skip one register round of the 8-stage pass
    PASS_DEBUG_NO_MATH_B = 64u,
    PASS_CROSS_BATCH = 128u,   // CONTIG 8-stage pass: a tile is the same 256-element block of 16 consecutive transforms
    PASS_POINTWISE = 256u,     // first pass of a backward transform: multiply by the vector at data2 first
    PASS_OFFSET32 = 512u,      // every element of a tile lies within 2^31 bytes of the tile's first: buffer addressing
    PASS_LAZY = 1024u,         // outputs of the transform in the policy's lazy range ([0, 2q)) instead of canonical
    PASS_POW2_TWIDDLES = 4096u, // host side only: STRIDED 8-stage pass with s0 = 0 on a table whose entries 1..15 are the
                               // standard powers of two -- selects the kernel whose first radix-16 round multiplies by shifts
    PASS_DEBUG_NO_MEM = 8192u,  // profiling aid (NTTX_DEBUG_NO_MEM=1): no global loads or stores of data -- registers are
                               // filled with made-up residues and results dropped: the pass time with memory out of the picture
    PASS_DEBUG_HOT_TILES = 2048u // profiling aid (NTTX_DEBUG_HOT_TILES=1): every workgroup works on one of the first 64
                               // tiles, so all traffic is served by the L2s -- the pass time with HBM out of the picture
};
// radix()'s `scale` argument
enum : int { SCALE_NONE = 0, SCALE_CANONICAL = 1, SCALE_LAZY = 2 };
__device__ __forceinline__ int scale_mode(uint32_t flags)
{
    return (flags & PASS_SCALE) ? ((flags & PASS_LAZY) ? SCALE_LAZY : SCALE_CANONICAL) : SCALE_NONE;
}

struct PassParams {
    uint64_t* data;
    const uint64_t* data2;    // PASS_POINTWISE: second operand, same layout as data
    const void* tw;           // device twiddle table of this direction, n entries of Arith::TwElem
    const void* tw_perm;      // same entries with the last four stages regrouped (see permuted_index)
    uint64_t nblocks_total;   // blocks in the buffer: batch << s0, or one shard's share of a split transform
    uint64_t block_base;      // index, inside its transform, of the buffer's first block (0 unless the
                              // buffer is one shard of a transform split over several devices)
    uint32_t logB;            // log2 of the block size n >> s0
    uint32_t s0;              // stages that precede this pass in forward order
    uint32_t flags;
    uint32_t xcd_group;       // 0: tile = workgroup id.  G > 0: runs of G consecutive tiles per XCD (see pass_kernel)
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

// Twiddle sources for radix(): entry `i` (1 <= i < 2^K) is the twiddle of stage
// u = floor(log2 i) + 1, block b = i - 2^(u-1).
template <class A, int K>
struct RegTw { // preloaded registers
    const typename A::Tw (&w)[1 << K];
    __device__ __forceinline__ typename A::Tw get(int i) const { return w[i]; }
};
template <class A>
struct LdsTw { // one 16-entry set staged in LDS, read where it is used
    const typename A::TwElem* set;
    const typename A::Ctx& c;
    __device__ __forceinline__ typename A::Tw get(int i) const { return A::load(set, i, c); }
};

// K butterfly stages on 2^K registers.  Register i holds sub-transform element i (stride
// handled by the caller); stage u = 1..K pairs i with i + 2^(K-u) and uses twiddle entry
// (1 << (u-1)) + (i >> (K-u+1)) of the source.
//
// A policy with DEFERRED_CHECK computes a stage with its fast butterflies, which report whether
// any of them may deviate from the strict rules; the stage is then redone strictly from its
// inputs.  One branch per stage keeps the 2^(K-1) independent butterflies interleavable.
//
// GROUPED: the last inverse stage in its whole-group form (gs_last_stage: one screen for all outputs);
// the generic rounds, which also keep their twiddles in registers, take the one-at-a-time form.
template <class A, int K, bool FWD, class Src, bool GROUPED = true>
__device__ __forceinline__ void radix(uint64_t (&x)[1 << K], const Src& w, const typename A::Ctx& c, int scale)
{
    constexpr int M = 1 << K;
    if constexpr (A::DEFERRED_CHECK) {
        uint64_t in[M];
#pragma unroll
        for (int step = 1; step <= K; step++) {
            const int u = FWD ? step : K + 1 - step;
            const int half = M >> u;
            const bool last = !FWD && u == 1 && scale;
#pragma unroll
            for (int i = 0; i < M; i++)
                in[i] = x[i];
            // maximum of the words the fast butterflies ask to be watched, folded two at a time
            uint32_t suspect = 0, held = 0;
            bool holding = false;
            auto watch = [&](const uint32_t* tops, int count) {
#pragma unroll
                for (int k = 0; k < count; k++) {
                    if (holding) {
                        suspect = A::max3(suspect, held, tops[k]);
                        holding = false;
                    } else {
                        held = tops[k];
                        holding = true;
                    }
                }
            };
#pragma unroll
            for (int b = 0; b < (1 << (u - 1)); b++) {
                typename A::Tw s = w.get((1 << (u - 1)) + b);
                if (last)
                    s = A::scaled(s, c);
#pragma unroll
                for (int i = 0; i < half; i++) {
                    const int lo = b * 2 * half + i;
                    uint32_t tops[3];
                    if (FWD) {
                        A::ct_fast(x[lo], x[lo + half], s, c, tops);
                        watch(tops, A::TOPS_CT);
                    } else if (last) {
                        A::gs_last_fast(x[lo], x[lo + half], s, c, tops);
                        watch(tops, A::TOPS_GS_LAST);
                    } else {
                        A::gs_fast(x[lo], x[lo + half], s, c, tops);
                        watch(tops, A::TOPS_GS);
                    }
                }
            }
            if (holding)
                suspect = A::max3(suspect, held, 0);
            if (__builtin_expect(suspect == A::SUSPECT, 0)) {
#pragma unroll
                for (int b = 0; b < (1 << (u - 1)); b++) {
                    typename A::Tw s = w.get((1 << (u - 1)) + b);
                    if (last)
                        s = A::scaled(s, c);
#pragma unroll
                    for (int i = 0; i < half; i++) {
                        const int lo = b * 2 * half + i;
                        x[lo] = in[lo];
                        x[lo + half] = in[lo + half];
                        if (FWD)
                            A::ct_strict(x[lo], x[lo + half], s, c);
                        else if (last)
                            A::gs_last_strict(x[lo], x[lo + half], s, c);
                        else
                            A::gs_strict(x[lo], x[lo + half], s, c);
                    }
                }
            }
        }
    } else if (FWD) {
#pragma unroll
        for (int u = 1; u <= K; u++) {
            const int half = M >> u;
#pragma unroll
            for (int b = 0; b < (1 << (u - 1)); b++) {
                const typename A::Tw s = w.get((1 << (u - 1)) + b);
#pragma unroll
                for (int i = 0; i < half; i++) {
                    const int lo = b * 2 * half + i;
                    A::ct(x[lo], x[lo + half], s, c);
                }
            }
        }
    } else {
        // stages K .. 2, then stage 1 on its own: it is the one that may carry the n^-1 factor, and its
        // code inside the loop would count K times against the unroller's size limit
#pragma unroll
        for (int u = K; u >= 2; u--) {
            const int half = M >> u;
#pragma unroll
            for (int b = 0; b < (1 << (u - 1)); b++) {
                const typename A::Tw s = w.get((1 << (u - 1)) + b);
#pragma unroll
                for (int i = 0; i < half; i++) {
                    const int lo = b * 2 * half + i;
                    A::gs(x[lo], x[lo + half], s, c);
                }
            }
        }
        constexpr int half = M >> 1;
        const typename A::Tw s = w.get(1);
        if (scale) {
            // last stage of the inverse with the n^-1 factor folded in (ntt.rs:231-235)
            if (A::HAS_LAZY && scale == SCALE_LAZY) {
#pragma unroll
                for (int i = 0; i < half; i++)
                    A::gs_last_lazy(x[i], x[i + half], s, c);
            } else if (GROUPED) {
                A::template gs_last_stage<M>(x, s, c);
            } else {
#pragma unroll
                for (int i = 0; i < half; i++)
                    A::gs_last(x[i], x[i + half], s, c);
            }
        } else {
#pragma unroll
            for (int i = 0; i < half; i++)
                A::gs(x[i], x[i + half], s, c);
        }
    }
}

// Compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// radix<A, 4, FWD> for a policy with DEFERRED_CHECK whose 15 twiddles are the table's entries 1..15 and
// known powers of two (A::pow2_exponent): the stages 1..4 of a forward transform, or 4..1 of an inverse
// one, on a table with the standard psi^(n/32).  The products become shifts (A::ct_pow2_fast /
// gs_pow2_fast); the strict redo of a suspect stage and the inverse stage that carries the n^-1 factor
// take the twiddle values from `w` as radix() does.
template <class A, bool FWD, class Src>
__device__ __forceinline__ void radix16_pow2(uint64_t (&x)[16], const Src& w, const typename A::Ctx& c, int scale)
{
    constexpr int K = 4, M = 16;
    uint64_t in[M];
    static_for<0, K>([&](auto step_c) {
        constexpr int step = decltype(step_c)::value + 1;
        constexpr int u = FWD ? step : K + 1 - step;
        constexpr int half = M >> u;
        const bool last = !FWD && u == 1 && scale;
#pragma unroll
        for (int i = 0; i < M; i++)
            in[i] = x[i];
        uint32_t suspect = 0, held = 0;
        bool holding = false;
        auto watch = [&](const uint32_t* tops, int count) {
#pragma unroll
            for (int k = 0; k < count; k++) {
                if (holding) {
                    suspect = A::max3(suspect, held, tops[k]);
                    holding = false;
                } else {
                    held = tops[k];
                    holding = true;
                }
            }
        };
        if (last) {
            // n^-1 = 2^-log2(n) and the scaled stage-1 twiddle are powers of two as well; their exponents
            // depend on n, so the sizes large batches come in have a compiled-in form each
            auto shifted = [&](auto log_c) {
                constexpr int L = decltype(log_c)::value;
#pragma unroll
                for (int i = 0; i < half; i++) {
                    uint32_t tops[3];
                    A::template gs_last_pow2_fast<L>(x[i], x[i + half], tops);
                    watch(tops, A::TOPS_GS_LAST);
                }
            };
            switch (c.logn) {
            case 16: shifted(std::integral_constant<int, 16>{}); break;
            case 20: shifted(std::integral_constant<int, 20>{}); break;
            case 24: shifted(std::integral_constant<int, 24>{}); break;
            case 28: shifted(std::integral_constant<int, 28>{}); break;
            default: {
                const typename A::Tw s = A::scaled(w.get(1), c);
#pragma unroll
                for (int i = 0; i < half; i++) {
                    uint32_t tops[3];
                    A::gs_last_fast(x[i], x[i + half], s, c, tops);
                    watch(tops, A::TOPS_GS_LAST);
                }
            }
            }
        } else {
            static_for<0, (1 << (u - 1))>([&](auto b_c) {
                constexpr int b = decltype(b_c)::value;
                constexpr int E = A::pow2_exponent((1 << (u - 1)) + b, !FWD);
#pragma unroll
                for (int i = 0; i < half; i++) {
                    const int lo = b * 2 * half + i;
                    uint32_t tops[4];
                    if (FWD) {
                        A::template ct_pow2_fast<E>(x[lo], x[lo + half], tops);
                        watch(tops, E < 96 ? 3 : A::TOPS_CT_POW2);
                    } else {
                        A::template gs_pow2_fast<E>(x[lo], x[lo + half], tops);
                        watch(tops, A::TOPS_GS);
                    }
                }
            });
        }
        if (holding)
            suspect = A::max3(suspect, held, 0);
        if (__builtin_expect(suspect == A::SUSPECT, 0)) {
#pragma unroll
            for (int b = 0; b < (1 << (u - 1)); b++) {
                typename A::Tw s = w.get((1 << (u - 1)) + b);
                if (last)
                    s = A::scaled(s, c);
#pragma unroll
                for (int i = 0; i < half; i++) {
                    const int lo = b * 2 * half + i;
                    x[lo] = in[lo];
                    x[lo + half] = in[lo + half];
                    if (FWD)
                        A::ct_strict(x[lo], x[lo + half], s, c);
                    else if (last)
                        A::gs_last_strict(x[lo], x[lo + half], s, c);
                    else
                        A::gs_strict(x[lo], x[lo + half], s, c);
                }
            }
        }
    });
}

// The reference's first sweep accepts any u64 (ntt.rs:125-127).  Values the butterflies can take
// as they are (below the policy's input_bound: residues, and for the lazy policy anything below
// 4q) are the common case: the largest high word of the thread's values is compared with the
// bound's high word once, and only when that screen fires is each value tested and reduced.
template <class A, int M>
__device__ __forceinline__ void condition_all(uint64_t (&x)[M], const typename A::Ctx& c)
{
    if constexpr (A::CONDITIONS_INPUT) {
        uint32_t qh = hi32(A::input_bound(c));
        asm volatile("" : "+s"(qh)); // keep the screen a 32-bit compare (hipcc widens it otherwise)
        uint32_t top = 0;
#pragma unroll
        for (int i = 0; i < M; i++)
            top = hi32(x[i]) > top ? hi32(x[i]) : top;
        if (__builtin_expect(top >= qh, 0)) {
#pragma unroll
            for (int i = 0; i < M; i++) {
                if (hi32(x[i]) >= qh)
                    x[i] = A::condition_input(x[i], c);
            }
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
template <int K, bool XB = false>
struct TileGeom {
    uint32_t logB, logT, logCw, lognblk, logtpb, s0;
    uint64_t tile, nblocks_total, block_base;

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
        block_base = p.block_base;
    }
    // Block (of 2^logB elements, counted over the whole buffer) that sub-transform sigma belongs to.
    // Ordinarily a tile takes consecutive blocks.  A cross-batch tile (CONTIG, K = 8: blocks are
    // its 16 sub-transforms) takes block `tile mod 2^s0` of 16 consecutive transforms instead, so
    // all its sub-transforms use the same twiddles.
    __device__ __forceinline__ uint64_t block_of(uint32_t sigma) const
    {
        if (XB)
            return ((((tile >> s0) << 4) + sigma) << s0) + (tile & ((1ull << s0) - 1));
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
    // gidx(sigma, r) = uniform_index() + lane_index(sigma, r): a wave-uniform part (scalar
    // registers) and a per-lane part that fits 32 bits when the host sets PASS_OFFSET32.
    __device__ __forceinline__ uint64_t uniform_index() const
    {
        if (XB)
            return ((((tile >> s0) << 4) << s0) + (tile & ((1ull << s0) - 1))) << logB;
        return ((((tile >> logtpb) << lognblk)) << logB) + ((tile & ((1ull << logtpb) - 1)) << logCw);
    }
    __device__ __forceinline__ uint64_t lane_index(uint32_t sigma, uint32_t r) const
    {
        if (XB)
            return ((uint64_t)sigma << (s0 + logB)) + ((uint64_t)r << logT);
        return ((uint64_t)(sigma >> logCw) << logB) + ((uint64_t)r << logT) + (sigma & ((1u << logCw) - 1));
    }
    // twiddle base of the block: 2^s0 + (block index inside its transform)
    __device__ __forceinline__ uint64_t twiddle_base(uint32_t sigma) const
    {
        return (1ull << s0) + ((block_base + block_of(sigma)) & ((1ull << s0) - 1));
    }
};

// Buffer addressing (scalar resource + 32-bit lane offset + scalar offset): the address of an
// element costs no vector arithmetic.  The resource starts at a tile's first element and spans
// 4 GiB, so the range check never fires; callers keep every offset inside the allocation.
// Cache policy of the data loads / stores (the `aux` operand of the buffer instructions on gfx950:
// 1 = sc0, 2 = nt, 16 = sc1, 0 = default policy).  A pass reads every element once and writes it
// once, so both are non-temporal.  A/B on the headline workload: default / nt loads / nt stores / both
// = 1.183 / 1.195 / 1.184 / 1.218 M pairs/s on one box (sc1 loads: 1.182), no difference on another;
// Goldilocks 2^16 and 2^20 gain 2 % there.  Twiddles are read through ordinary loads and stay cached.
#ifndef NTTX_LOAD_AUX
#define NTTX_LOAD_AUX 2
#endif
#ifndef NTTX_STORE_AUX
#define NTTX_STORE_AUX 2
#endif
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_resource(const uint64_t* first)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(first), 0, 0xffffffff, 0x00020000);
}
template <int AUX = NTTX_LOAD_AUX>
__device__ __forceinline__ uint64_t buffer_load64(__amdgpu_buffer_rsrc_t r, uint32_t lane_bytes, uint32_t uniform_bytes)
{
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, lane_bytes, uniform_bytes, AUX);
    return pack64(v.x, v.y);
}
template <int AUX = NTTX_LOAD_AUX>
__device__ __forceinline__ ulonglong2 buffer_load128(__amdgpu_buffer_rsrc_t r, uint32_t lane_bytes, uint32_t uniform_bytes)
{
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, uniform_bytes, AUX);
    return make_ulonglong2(pack64(v.x, v.y), pack64(v.z, v.w));
}
template <int AUX = NTTX_STORE_AUX>
__device__ __forceinline__ void buffer_store128(__amdgpu_buffer_rsrc_t r, uint32_t lane_bytes, uint32_t uniform_bytes,
                                                ulonglong2 x)
{
    const u32x4_t v = {lo32(x.x), hi32(x.x), lo32(x.y), hi32(x.y)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, lane_bytes, uniform_bytes, AUX);
}
template <int AUX = NTTX_STORE_AUX>
__device__ __forceinline__ void buffer_store64(__amdgpu_buffer_rsrc_t r, uint32_t lane_bytes, uint32_t uniform_bytes,
                                               uint64_t x)
{
    const u32x2_t v = {lo32(x), hi32(x)};
    __builtin_amdgcn_raw_buffer_store_b64(v, r, lane_bytes, uniform_bytes, AUX);
}

// Made-up residue (below 2^60) for element i of this thread: stands in for a load under PASS_DEBUG_NO_MEM.
__device__ __forceinline__ uint64_t made_up_value(int i)
{
    const uint32_t h = (threadIdx.x * 2654435761u) ^ ((uint32_t)i * 40503u);
    return pack64(h * 2246822519u, h & 0x0fffffffu);
}

// Tests a pass flag through an opaque scalar copy: the compiler then cannot tell that two tests
// of the same flag agree, and does not clone everything between them for each outcome.
__device__ __forceinline__ bool flag_set(uint32_t flags, uint32_t bit)
{
    asm volatile("" : "+s"(flags));
    return flags & bit;
}

// One register round of a pass for one group of 2^KR registers.  ROUND_A covers the K1 stages
// with the larger strides (elements rlow + (i << K2)), round B the K2 stages with the smaller
// ones (elements (rhigh << K2) + i).
template <class A, bool CONTIG, int K1, int K2, bool ROUND_A, bool XB = false>
struct Round {
    using Geom = TileGeom<K1 + K2, XB>;
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

    __device__ __forceinline__ Round(const Geom& g, int gamma)
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

    // A sub-transform past the end of the buffer (ragged last tile) reads the tile's first
    // block or transform instead -- always inside the buffer -- and stores nothing.
    __device__ __forceinline__ uint32_t safe_sigma(const Geom& g) const
    {
        if (ok)
            return sigma;
        return XB ? 0u : (sigma & ((1u << g.logCw) - 1));
    }
    __device__ __forceinline__ void load_global(uint64_t (&x)[M], const PassParams& p, const Geom& g,
                                                const typename A::Ctx& c) const
    {
        const uint32_t ss = safe_sigma(g);
        if (NTTX_PROF(flag_set(p.flags, PASS_DEBUG_NO_MEM))) {
#pragma unroll
            for (int i = 0; i < M; i++)
                x[i] = made_up_value(i);
        } else if (flag_set(p.flags, PASS_OFFSET32)) {
            const __amdgpu_buffer_rsrc_t res = tile_resource(p.data + g.uniform_index());
            const uint32_t lane = (uint32_t)g.lane_index(ss, r0) * 8u;
            const uint32_t stepb = 8u << (g.logT + rstep_log);
#pragma unroll
            for (int i = 0; i < M; i++)
                x[i] = buffer_load64(res, lane, i * stepb);
        } else {
            const uint64_t* src = p.data + g.uniform_index() + g.lane_index(ss, r0);
            const uint64_t step = 1ull << (g.logT + rstep_log);
#pragma unroll
            for (int i = 0; i < M; i++)
                x[i] = src[i * step];
        }
        if (p.flags & PASS_CONDITION_INPUT)
            condition_all<A, M>(x, c);
    }
    __device__ __forceinline__ void store_global(const uint64_t (&x)[M], const PassParams& p,
                                                 const Geom& g) const
    {
        if (!ok || NTTX_PROF(flag_set(p.flags, PASS_DEBUG_NO_MEM)))
            return;
        if (flag_set(p.flags, PASS_OFFSET32)) {
            const __amdgpu_buffer_rsrc_t res = tile_resource(p.data + g.uniform_index());
            const uint32_t lane = (uint32_t)g.lane_index(sigma, r0) * 8u;
            const uint32_t stepb = 8u << (g.logT + rstep_log);
#pragma unroll
            for (int i = 0; i < M; i++)
                buffer_store64(res, lane, i * stepb, x[i]);
        } else {
            uint64_t* dst = p.data + g.gidx(sigma, r0);
            const uint64_t step = 1ull << (g.logT + rstep_log);
#pragma unroll
            for (int i = 0; i < M; i++)
                dst[i * step] = x[i];
        }
    }
    // position of register i in the padded exchange tile
    __device__ __forceinline__ uint32_t lds_index(int i) const
    {
        return lds_addr<CONTIG, K>(sigma, r0 + ((uint32_t)i << rstep_log));
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

template <class A, int M, bool GROUPED = true>
__device__ __forceinline__ void finish_all(uint64_t (&x)[M], const typename A::Ctx& c, uint32_t flags)
{
    if (A::HAS_LAZY && flag_set(flags, PASS_LAZY)) {
#pragma unroll
        for (int i = 0; i < M; i++)
            x[i] = A::finish_lazy(x[i], c);
    } else if (GROUPED) {
        A::template finish_group<M>(x, c);
    } else {
#pragma unroll
        for (int i = 0; i < M; i++)
            x[i] = A::finish(x[i], c);
    }
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
        if (!NTTX_PROF(p.flags & PASS_DEBUG_NO_MATH))
            radix<A, R::KR, FWD, RegTw<A, R::KR>, false>(x, RegTw<A, R::KR>{w}, c, ROUND_A ? scale_mode(p.flags) : SCALE_NONE);
        if (FWD && LAST_OF_PASS && (p.flags & PASS_FINISH))
            finish_all<A, R::M, false>(x, c, p.flags);
        if (DST_G)
            r.store_global(x, p, g);
        else
            r.store_lds(x, lds);
    }
}

// Orders this wave's LDS writes before its later LDS reads from other lanes' addresses (LDS
// executes a wave's instructions in order; this only stops the compiler from moving them).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Copies a CONTIG tile between HBM and LDS, 16 bytes per lane.  XB: the tile is 16 runs of 256
// elements (cross-batch tile) instead of one run of 4096.
// PER_WAVE: every wave moves its own quarter of the tile (elements [1024 w, 1024 w + 1024)).
//
// Step i of a thread handles tile elements f, f + 1 with f = fu(i) + fl: fl depends on the lane
// only and fu(i) is wave-uniform, so run numbers, bounds checks and the uniform part of every
// address stay in scalar registers and the lane parts are computed once.
template <class A, int K, bool TO_LDS, bool XB = false, bool PER_WAVE = false>
__device__ __forceinline__ void copy_contig(const PassParams& p, const TileGeom<K, XB>& g, const typename A::Ctx& c,
                                            uint64_t* lds)
{
    static_assert(!XB || K == 8, "cross-batch tiles have 256-element runs");
    uint32_t fl, group = 0;
    if (PER_WAVE) {
        group = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        fl = 2 * (threadIdx.x & 63);
    } else if (K == 8) {
        group = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
        fl = 2 * (threadIdx.x & 127);
    } else {
        fl = 2 * threadIdx.x;
    }
    auto fu = [&](int i) -> uint32_t {
        if (PER_WAVE)
            return group * (TILE / 4) + i * 128;
        if (K == 8)
            return (2 * i + group) * 256;
        return 2 * i * THREADS;
    };
    // first element of the tile in the buffer, and what is left of the buffer from there
    const uint64_t tau16 = (g.tile >> g.s0) << 4;
    const uint64_t first = XB ? (((tau16 << g.s0) + (g.tile & ((1ull << g.s0) - 1))) << K) : (g.tile << TILE_LOG);
    const uint64_t run_stride = 1ull << (g.s0 + K); // XB: run sigma starts n elements after run sigma - 1
    uint32_t left;                                   // XB: runs (transforms) present; else: elements present
    if (XB) {
        const uint64_t runs = (p.nblocks_total >> g.s0) - tau16;
        left = runs < 16 ? (uint32_t)runs : 16u;
    } else {
        const uint64_t rest = (p.nblocks_total << p.logB) - first;
        left = rest < TILE ? (uint32_t)rest : (uint32_t)TILE;
    }
    // uniform element offset of step i from `first`, and whether lane fl takes part (n >= 4 makes
    // the element count a multiple of 4, and f is even: pairs are present or absent together)
    auto uniform_elems = [&](int i) -> uint64_t {
        const uint32_t u = fu(i);
        return XB ? (u >> K) * run_stride + (u & ((1u << K) - 1)) : (uint64_t)u;
    };
    auto present = [&](int i) -> bool { return XB ? (fu(i) >> K) < left : fu(i) + fl < left; };
    const bool buffers = flag_set(p.flags, PASS_OFFSET32);
    const bool no_mem = NTTX_PROF(flag_set(p.flags, PASS_DEBUG_NO_MEM));
    auto load_pair = [&](const uint64_t* from, int i) -> ulonglong2 {
        ulonglong2 v = make_ulonglong2(0, 0);
        if (no_mem) {
            v = make_ulonglong2(made_up_value(2 * i), made_up_value(2 * i + 1));
        } else if (present(i)) {
            if (buffers)
                v = buffer_load128(tile_resource(from + first), fl * 8, (uint32_t)uniform_elems(i) * 8);
            else
                v = *reinterpret_cast<const ulonglong2*>(from + first + uniform_elems(i) + fl);
        }
        return v;
    };
    // LDS position of element f: f + (f >> 4) = [fu + (fu >> 4)] + [fl + (fl >> 4)] (fu is a multiple of 16)
    uint64_t* const lane_lds = lds + (fl + (fl >> 4));
    auto lds_of = [&](int i) -> uint64_t* { return lane_lds + (fu(i) + (fu(i) >> 4)); };

    if (TO_LDS) {
        uint64_t x[REGS];
#pragma unroll
        for (int i = 0; i < REGS / 2; i++) {
            const ulonglong2 v = load_pair(p.data, i);
            x[2 * i] = v.x;
            x[2 * i + 1] = v.y;
        }
        if (p.flags & PASS_CONDITION_INPUT)
            condition_all<A, REGS>(x, c);
        if (p.flags & PASS_POINTWISE) {
            // product of two forward transforms, element by element, ahead of the backward stages
#pragma unroll
            for (int i = 0; i < REGS / 2; i++) {
                const ulonglong2 v = load_pair(p.data2, i);
                x[2 * i] = A::pointwise(x[2 * i], v.x, c);
                x[2 * i + 1] = A::pointwise(x[2 * i + 1], v.y, c);
            }
        }
#pragma unroll
        for (int i = 0; i < REGS / 2; i++) {
            uint64_t* const at = lds_of(i); // f is even, so f and f + 1 share a padding group
            at[0] = x[2 * i];
            at[1] = x[2 * i + 1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < REGS / 2; i++) {
            if (present(i) && !no_mem) {
                const uint64_t* const at = lds_of(i);
                ulonglong2 v;
                v.x = at[0];
                v.y = at[1];
                if (buffers)
                    buffer_store128(tile_resource(p.data + first), fl * 8, (uint32_t)uniform_elems(i) * 8, v);
                else
                    *reinterpret_cast<ulonglong2*>(p.data + first + uniform_elems(i) + fl) = v;
            }
        }
    }
}

// Where entry idx = (P16 * 16 + rhigh) * 2^(v-1) + i of the last four stages' twiddles (v = 1..4,
// the stride-8..1 stages of a transform; P16 counts groups of 16 consecutive rhigh) lives in the
// regrouped table: inside each group the 16 rhigh values become the fastest index, so the 16
// lanes that share a sub-transform read 16 consecutive entries.  Stage v occupies
// [n >> (5 - v), n >> (4 - v)) in both layouts.
__host__ __device__ inline uint64_t permuted_index(uint64_t idx, uint32_t logn)
{
    if (logn < 8 || idx < (1ull << (logn - 4)))
        return idx;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t top = 63 - __clzll((long long)idx);
#else
    const uint32_t top = 63 - (uint32_t)__builtin_clzll(idx);
#endif
    const uint32_t v = top - (logn - 5);            // 1..4
    const uint64_t base = 1ull << top;
    const uint64_t off = idx - base;                 // (beta * 16 + rhigh) << (v-1) | i
    const uint64_t i = off & ((1ull << (v - 1)) - 1);
    const uint64_t pr = off >> (v - 1);
    const uint64_t rhigh = pr & 15, beta = pr >> 4;
    return base + (beta << (v + 3)) + (i << 4) + rhigh;
}

// The 8-stage pass (two radix-16 rounds, one group per thread).
//
// Twiddles.  Three tile kinds:
//   STRIDED           round A: one set of 15 for the whole tile; round B: 16 sets, one per rhigh.
//   CONTIG cross-batch (XB: the same 256-element block of 16 transforms) -- the same as STRIDED.
//   CONTIG plain      round A: 16 sets, one per sub-transform; round B: one set per thread (the
//                     last four stages of one transform have n/16 distinct sets).
// Shared sets are fetched once per workgroup -- one 16-byte load per thread -- into an LDS table
// (set s at entry 17 s: the odd stride keeps the 16 sets a wave may read at once on different
// banks) and read at the point of use; per-thread sets come from the regrouped table with
// lane-contiguous loads, requested before the workgroup barrier so their latency overlaps the LDS
// exchange.  The plain CONTIG tile spends 60 registers on them and reads as many twiddle bytes as
// data bytes, which is why batches of 16 or more transforms take the cross-batch tile.
//
// Data: in a CONTIG tile the round with 128-byte runs (round A) talks to HBM directly; the
// other side goes through LDS for 16-byte-per-lane coalescing.
constexpr int TW_SET_STRIDE = 17;
constexpr int LDSTW_ELEMS = 16 * TW_SET_STRIDE + 16;

// FUSED (plain CONTIG tiles only): the round that talks to HBM directly (round A) reads from /
// writes to the tile held in LDS instead, at the element's position in the tile (the CONTIG
// exchange layout is the tile itself); fused_small_kernel runs the remaining stages around it.
template <class A, bool FWD, bool CONTIG, bool XB, bool FUSED = false>
__device__ __forceinline__ void pass_8(const PassParams& p, const TileGeom<8, XB>& g, const typename A::Ctx& c,
                                       uint64_t* lds, typename A::TwElem* ldstw)
{
    static_assert(CONTIG || !XB, "cross-batch tiles are CONTIG tiles");
    static_assert(!FUSED || (CONTIG && !XB), "only plain CONTIG tiles are fused");
    constexpr bool SHARED_B = !CONTIG || XB; // round B reads shared sets, round A the single set
    using RA = Round<A, CONTIG, 4, 4, true, XB>;
    using RB = Round<A, CONTIG, 4, 4, false, XB>;
    using TwElem = typename A::TwElem;
    const TwElem* tw = static_cast<const TwElem*>(p.tw);
    const TwElem* twp = static_cast<const TwElem*>(p.tw_perm);
    const RA ra(g, 0);
    const RB rb(g, 0);
    uint64_t x[16];
    const int scale = scale_mode(p.flags);
    const bool math = !NTTX_PROF(p.flags & PASS_DEBUG_NO_MATH);
    const bool math_a = math && !NTTX_PROF(p.flags & PASS_DEBUG_NO_MATH_A);
    const bool math_b = math && !NTTX_PROF(p.flags & PASS_DEBUG_NO_MATH_B);
    const bool twl = !NTTX_PROF(p.flags & PASS_DEBUG_NO_TW);

    // ---- stage the shared twiddle sets: thread t fetches entry j = t & 15 of set t >> 4 ----
    const uint32_t set = threadIdx.x >> 4, j = threadIdx.x & 15;
    const uint32_t u = 32 - __clz(j | 1);           // stage of entry j (j = 0 is unused)
    const uint32_t b = j ? j - (1u << (u - 1)) : 0;
    // SHARED_B: set = rhigh of round B, base (P << 4) + set; plain CONTIG: set = sigma of round A
    const uint64_t Pset = SHARED_B ? ((g.twiddle_base(0) << 4) + set) : (g.valid(set) ? g.twiddle_base(set) : 1);
    // moved as raw dwords: a struct temporary held across the data loads ends up on the stack
    using Raw = __attribute__((ext_vector_type(sizeof(TwElem) / 4))) uint32_t;
    const Raw mine = *reinterpret_cast<const Raw*>(tw + ((Pset << (u - 1)) + b));
    const Raw uni = *reinterpret_cast<const Raw*>(tw + ((g.twiddle_base(0) << (u - 1)) + b)); // threads 0..15, SHARED_B
    // the single set of round A sits after the 16 sets
    TwElem* const shared_sets = ldstw;
    TwElem* const uniform_set = ldstw + 16 * TW_SET_STRIDE;
    auto stage_twiddles = [&]() {
        reinterpret_cast<Raw*>(shared_sets)[set * TW_SET_STRIDE + j] = mine;
        if (SHARED_B && threadIdx.x < 16)
            reinterpret_cast<Raw*>(uniform_set)[threadIdx.x] = uni;
    };
    // the set each round reads
    const TwElem* const set_a = SHARED_B ? uniform_set : shared_sets + ra.sigma * TW_SET_STRIDE;
    const TwElem* const set_b = shared_sets + (CONTIG ? (threadIdx.x & 15) : (threadIdx.x >> 4)) * TW_SET_STRIDE;

    // per-thread twiddles of plain CONTIG round B: entry (1 << (v-1)) + i <- regrouped table
    auto load_last = [&](typename A::Tw (&w)[16]) {
        const uint64_t beta16 = (rb.P >> 4) << 4;   // twiddle base of rhigh = 0 in this sub-transform
        const uint32_t rhigh = (uint32_t)(rb.P & 15);
#pragma unroll
        for (int v = 1; v <= 4; v++) {
            // original index (beta16 + rhigh) << (v-1) | i  ->  (beta16 << (v-1)) + (i << 4) + rhigh
            const TwElem* row = twp + (beta16 << (v - 1)) + rhigh;
#pragma unroll
            for (int i = 0; i < (1 << (v - 1)); i++)
                w[(1 << (v - 1)) + i] = A::load(row, i << 4, c);
        }
    };

    if (FWD) {
        if (FUSED)
            ra.load_lds(x, lds);
        else
            ra.load_global(x, p, g, c);
        stage_twiddles();
        __syncthreads();
        if (math_a)
            radix<A, 4, true>(x, LdsTw<A>{set_a, c}, c, SCALE_NONE);
        ra.store_lds(x, lds);
        if (CONTIG) {
            typename A::Tw w[16];
            if (!SHARED_B && twl)
                load_last(w);
            __syncthreads();
            rb.load_lds(x, lds);
            if (math_b) {
                if constexpr (SHARED_B)
                    radix<A, 4, true>(x, LdsTw<A>{set_b, c}, c, SCALE_NONE);
                else
                    radix<A, 4, true>(x, RegTw<A, 4>{w}, c, SCALE_NONE);
            }
            if (p.flags & PASS_FINISH)
                finish_all<A, 16>(x, c, p.flags);
            // Round B left each wave with four whole 256-element sub-transforms, so the way back
            // to HBM (16 bytes per lane) only moves data between lanes of one wave: no workgroup
            // barrier, the waves leave as they finish.
            rb.store_lds(x, lds); // the addresses this thread just read: no hazard with other threads
            wave_sync();
            copy_contig<A, 8, false, XB, true>(p, g, c, lds);
        } else {
            __syncthreads();
            rb.load_lds(x, lds);
            if (math_b)
                radix<A, 4, true>(x, LdsTw<A>{set_b, c}, c, SCALE_NONE);
            if (p.flags & PASS_FINISH)
                finish_all<A, 16>(x, c, p.flags);
            rb.store_global(x, p, g);
        }
    } else {
        if (CONTIG) {
            typename A::Tw w[16];
            copy_contig<A, 8, true, XB>(p, g, c, lds);
            if (!SHARED_B && twl)
                load_last(w); // after the tile copy: 64 more live registers during it would spill
            stage_twiddles();
            __syncthreads();
            rb.load_lds(x, lds);
            if (math_b) {
                if constexpr (SHARED_B)
                    radix<A, 4, false>(x, LdsTw<A>{set_b, c}, c, SCALE_NONE);
                else
                    radix<A, 4, false>(x, RegTw<A, 4>{w}, c, SCALE_NONE);
            }
            rb.store_lds(x, lds);
        } else {
            rb.load_global(x, p, g, c);
            stage_twiddles();
            __syncthreads();
            if (math_b)
                radix<A, 4, false>(x, LdsTw<A>{set_b, c}, c, SCALE_NONE);
            rb.store_lds(x, lds);
        }
        __syncthreads();
        ra.load_lds(x, lds);
        if (math_a)
            radix<A, 4, false>(x, LdsTw<A>{set_a, c}, c, scale);
        if (FUSED)
            ra.store_lds(x, lds); // the positions this thread just read
        else
            ra.store_global(x, p, g);
    }
}


// ---- the 8-stage pass on half the LDS ------------------------------------------------------------
//
// pass_8 hands a tile from one register round to the other through a 34 KiB LDS tile; with the
// twiddle table that is 38.5 KiB per workgroup and four workgroups (16 waves) per CU, whatever the
// kernels' register counts would allow.  pass_8h does the same work in 21.5 KiB for the two tile
// kinds the large batches run on (STRIDED, and the cross-batch CONTIG tile: twiddle sets shared by
// the workgroup in both rounds):
//   * the exchange between the rounds moves the low 32-bit halves of all 4096 elements through a
//     17 KiB tile of words, then the high halves through the same tile (exchange_planes): the same
//     LDS bytes as before in twice as many, half as wide instructions, and two more barriers;
//   * the cross-batch tile's way to and from HBM with 16 bytes per lane (move_runs) is private to a
//     wave -- round B leaves each wave with four whole 256-element runs -- and goes through a
//     4.25 KiB staging area per wave, two runs at a time, with no workgroup barrier.
constexpr int WAVE_STAGE_ELEMS = 512 + 512 / 16; // two runs of 256 elements, padded like the CONTIG tile
static_assert(4 * WAVE_STAGE_ELEMS * sizeof(uint64_t) <= LDS_ELEMS * sizeof(uint32_t),
              "the four waves' staging areas overlay the word tile");

// Register layout `from` -> register layout `to` of the same tile, one 32-bit plane at a time.
// Called by all threads of the workgroup; nobody may still be using `plane` for something else.
template <class RFrom, class RTo>
__device__ __forceinline__ void exchange_planes(uint64_t (&x)[16], const RFrom& from, const RTo& to, uint32_t* plane)
{
    uint32_t low[16];
#pragma unroll
    for (int i = 0; i < 16; i++)
        plane[from.lds_index(i)] = lo32(x[i]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++)
        low[i] = plane[to.lds_index(i)];
    __syncthreads(); // every low word has been taken out before a high word lands on it
#pragma unroll
    for (int i = 0; i < 16; i++)
        plane[from.lds_index(i)] = hi32(x[i]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++)
        x[i] = pack64(low[i], plane[to.lds_index(i)]);
}

// Cross-batch tile (16 runs of 256 contiguous elements, run sigma in transform 16 (tile >> s0) + sigma)
// <-> the registers of round B, for one wave: thread (sigma = 4w + (lane >> 4), rhigh = lane & 15)
// holds elements 16 rhigh .. 16 rhigh + 15 of run sigma, and HBM is addressed with lane-contiguous
// accesses (stores: 16 bytes per lane, step j = 0..7: elements 2 lane, 2 lane + 1 of half j & 1 of run
// 4w + (j >> 1); loads: 8 bytes per lane, see below).  The two meet in the wave's staging area, two runs
// (lanes 32h .. 32h+31 of round B) at a time.
// TO_REGS: HBM -> registers, with the input screen and the pointwise product of a first backward
// pass; otherwise registers -> HBM.
template <class A, bool TO_REGS>
__device__ __forceinline__ void move_runs(uint64_t (&x)[16], const PassParams& p, const TileGeom<8, true>& g,
                                          const typename A::Ctx& c, uint64_t* stage_all)
{
    const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    uint64_t* const stage = stage_all + w * WAVE_STAGE_ELEMS;
    // first element of the tile in the buffer, distance between runs, runs present (as copy_contig)
    const uint64_t tau16 = (g.tile >> g.s0) << 4;
    const uint64_t first = ((tau16 << g.s0) + (g.tile & ((1ull << g.s0) - 1))) << 8;
    const uint64_t run_stride = 1ull << (g.s0 + 8);
    const uint64_t runs = (p.nblocks_total >> g.s0) - tau16;
    const uint32_t left = runs < 16 ? (uint32_t)runs : 16u;
    const bool buffers = flag_set(p.flags, PASS_OFFSET32);
    auto run_of = [&](int j) -> uint32_t { return 4 * w + (uint32_t)(j >> 1); };
    auto uniform_elems = [&](int j) -> uint64_t { return run_of(j) * run_stride + (uint32_t)(j & 1) * 128u; };
    auto present = [&](int j) -> bool { return run_of(j) < left; };
    // staging position of element e of the two runs in flight: e + (e >> 4), computed in two parts
    uint64_t* const lane_at = stage + (2 * lane + (lane >> 3));         // e = 128 jj + 2 lane
    auto step_at = [&](int jj) -> uint64_t* { return lane_at + jj * 136; };
    const uint32_t own = ((lane >> 4) & 1) * 256 + (lane & 15) * 16;    // first of this thread's 16 elements
    uint64_t* const own_at = stage + (own + (own >> 4));
    const uint32_t own_half = lane >> 5;

    if (TO_REGS) {
        // Loads take 8 bytes per lane (step j = 0..15: element 64 (j & 3) + lane of run 4w + (j >> 2), 512
        // contiguous bytes per wave instruction): measured on this tile shape, 16-byte-per-lane loads reach
        // 4.8 TB/s where these reach 5.9 (tools/ubench_tiles.hip); stores show no such difference.  A run
        // past the end of the buffer reads the tile's first run instead (always present) and is never stored.
        auto load_all = [&](const uint64_t* from, uint64_t (&t)[16]) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t run = 4 * w + (uint32_t)(j >> 2);
                const uint64_t u = (run < left ? run : 0u) * run_stride + (uint32_t)(j & 3) * 64u;
                if (NTTX_PROF(flag_set(p.flags, PASS_DEBUG_NO_MEM)))
                    t[j] = made_up_value(j);
                else if (buffers)
                    t[j] = buffer_load64(tile_resource(from + first), lane * 8, (uint32_t)u * 8);
                else
                    t[j] = from[first + u + lane];
            }
        };
        uint64_t t[16];
        load_all(p.data, t);
        if (p.flags & PASS_CONDITION_INPUT)
            condition_all<A, 16>(t, c);
        if (p.flags & PASS_POINTWISE) {
            uint64_t t2[16];
            load_all(p.data2, t2);
#pragma unroll
            for (int i = 0; i < 16; i++)
                t[i] = A::pointwise(t[i], t2[i], c);
        }
        uint64_t* const lane8_at = stage + (lane + (lane >> 4)); // element 64 jj + lane of the two runs in flight
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int jj = 0; jj < 8; jj++)
                lane8_at[jj * 68] = t[8 * h + jj]; // 64 jj + ((64 jj) >> 4)
            wave_sync();
            if (own_half == (uint32_t)h) {
#pragma unroll
                for (int i = 0; i < 16; i++)
                    x[i] = own_at[i];
            }
            wave_sync(); // the area is free for the next two runs
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (own_half == (uint32_t)h) {
#pragma unroll
                for (int i = 0; i < 16; i++)
                    own_at[i] = x[i];
            }
            wave_sync();
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int j = 4 * h + jj;
                if (present(j) && !NTTX_PROF(flag_set(p.flags, PASS_DEBUG_NO_MEM))) {
                    const uint64_t* const at = step_at(jj);
                    ulonglong2 v;
                    v.x = at[0];
                    v.y = at[1];
                    if (buffers)
                        buffer_store128(tile_resource(p.data + first), lane * 16, (uint32_t)uniform_elems(j) * 8, v);
                    else
                        *reinterpret_cast<ulonglong2*>(p.data + first + uniform_elems(j) + 2 * lane) = v;
                }
            }
            wave_sync();
        }
    }
}

// POW2: the round with the larger strides multiplies by shifts (radix16_pow2); the host launches that
// instantiation for passes with s0 = 0 of plans whose table entries 1..15 are the standard powers of two.
template <class A, bool FWD, bool CONTIG, bool POW2 = false>
__device__ __forceinline__ void pass_8h(const PassParams& p, const TileGeom<8, CONTIG>& g, const typename A::Ctx& c,
                                        uint32_t* plane, typename A::TwElem* ldstw)
{
    constexpr bool XB = CONTIG; // the only CONTIG tile handled here is the cross-batch one
    using RA = Round<A, CONTIG, 4, 4, true, XB>;
    using RB = Round<A, CONTIG, 4, 4, false, XB>;
    using TwElem = typename A::TwElem;
    const TwElem* tw = static_cast<const TwElem*>(p.tw);
    const RA ra(g, 0);
    const RB rb(g, 0);
    uint64_t x[16];
    const int scale = scale_mode(p.flags);
    const bool math = !NTTX_PROF(p.flags & PASS_DEBUG_NO_MATH);
    const bool math_a = math && !NTTX_PROF(p.flags & PASS_DEBUG_NO_MATH_A);
    const bool math_b = math && !NTTX_PROF(p.flags & PASS_DEBUG_NO_MATH_B);

    // twiddles as in pass_8: thread t fetches entry j = t & 15 of round B's set t >> 4, threads
    // 0..15 also entry t of round A's single set
    const uint32_t set = threadIdx.x >> 4, j = threadIdx.x & 15;
    const uint32_t u = 32 - __clz(j | 1);
    const uint32_t b = j ? j - (1u << (u - 1)) : 0;
    const uint64_t P0 = g.twiddle_base(0);
    using Raw = __attribute__((ext_vector_type(sizeof(TwElem) / 4))) uint32_t;
    const Raw mine = *reinterpret_cast<const Raw*>(tw + ((((P0 << 4) + set) << (u - 1)) + b));
    const Raw uni = *reinterpret_cast<const Raw*>(tw + ((P0 << (u - 1)) + b));
    TwElem* const shared_sets = ldstw;
    TwElem* const uniform_set = ldstw + 16 * TW_SET_STRIDE;
    auto stage_twiddles = [&]() {
        reinterpret_cast<Raw*>(shared_sets)[set * TW_SET_STRIDE + j] = mine;
        if (threadIdx.x < 16)
            reinterpret_cast<Raw*>(uniform_set)[threadIdx.x] = uni;
    };
    const LdsTw<A> set_a{uniform_set, c};
    const LdsTw<A> set_b{shared_sets + (CONTIG ? (threadIdx.x & 15) : (threadIdx.x >> 4)) * TW_SET_STRIDE, c};
    uint64_t* const stage = reinterpret_cast<uint64_t*>(plane);

    if (FWD) {
        ra.load_global(x, p, g, c);
        stage_twiddles();
        __syncthreads();
        if (math_a) {
            if constexpr (POW2)
                radix16_pow2<A, true>(x, set_a, c, SCALE_NONE);
            else
                radix<A, 4, true>(x, set_a, c, SCALE_NONE);
        }
        exchange_planes(x, ra, rb, plane);
        if (math_b)
            radix<A, 4, true>(x, set_b, c, SCALE_NONE);
        if (p.flags & PASS_FINISH)
            finish_all<A, 16>(x, c, p.flags);
        if constexpr (CONTIG) {
            __syncthreads(); // every wave has its words out of the planes, which the staging areas overlay
            move_runs<A, false>(x, p, g, c, stage);
        } else {
            rb.store_global(x, p, g);
        }
    } else {
        if constexpr (CONTIG)
            move_runs<A, true>(x, p, g, c, stage);
        else
            rb.load_global(x, p, g, c);
        stage_twiddles();
        __syncthreads(); // twiddles staged; every wave is done with its staging area
        if (math_b)
            radix<A, 4, false>(x, set_b, c, SCALE_NONE);
        exchange_planes(x, rb, ra, plane);
        if (math_a) {
            if constexpr (POW2)
                radix16_pow2<A, false>(x, set_a, c, scale);
            else
                radix<A, 4, false>(x, set_a, c, scale);
        }
        ra.store_global(x, p, g);
    }
}

// Which pass_kernel shapes run pass_8h, and with how many waves per SIMD the compiler may count.
#ifndef NTTX_HALF_LDS
#define NTTX_HALF_LDS 1
#endif
#ifndef NTTX_HALF_WAVES
#define NTTX_HALF_WAVES 4
#endif
template <bool CONTIG, int K1, int K2, bool XB>
constexpr bool half_lds_shape()
{
    return NTTX_HALF_LDS && K1 == 4 && K2 == 4 && (!CONTIG || XB);
}
template <bool CONTIG, int K1, int K2, bool XB>
constexpr int pass_waves()
{
    return half_lds_shape<CONTIG, K1, K2, XB>() ? NTTX_HALF_WAVES : 4;
}

template <class A, bool FWD, bool CONTIG, int K1, int K2, bool XB = false, bool POW2 = false>
__global__ __launch_bounds__(THREADS, (pass_waves<CONTIG, K1, K2, XB>())) void pass_kernel(PassParams p, typename A::Ctx c)
{
    static_assert(!POW2 || (A::HAS_POW2_TWIDDLES && half_lds_shape<CONTIG, K1, K2, XB>() && !CONTIG),
                  "shift twiddles: the STRIDED 8-stage pass of a policy that has them");
    constexpr int K = K1 + K2;
    // Workgroups are handed to the 8 XCDs round-robin (id % 8).  With xcd_group = G the tiles are
    // dealt so that each XCD works on runs of G consecutive tiles: workgroup id -> run
    // (id / 8G) * 8 + id % 8, position (id / 8) % G inside it.  The host only sets G when the
    // grid is a multiple of 8G.
    uint64_t tile = blockIdx.x;
    if (p.xcd_group) {
        const uint32_t G = p.xcd_group;
        tile = ((uint64_t)(blockIdx.x / (8 * G)) * 8 + blockIdx.x % 8) * G + (blockIdx.x / 8) % G;
    }
    if (NTTX_PROF(p.flags & PASS_DEBUG_HOT_TILES))
        tile &= 63;
    const TileGeom<K, XB> g(p, tile);

    if constexpr (half_lds_shape<CONTIG, K1, K2, XB>()) {
        __shared__ __attribute__((aligned(16))) uint32_t plane[LDS_ELEMS];
        __shared__ __attribute__((aligned(16))) typename A::TwElem ldstw[LDSTW_ELEMS];
        pass_8h<A, FWD, CONTIG, POW2>(p, g, c, plane, ldstw);
    } else if constexpr (K1 == 4 && K2 == 4) {
        __shared__ __attribute__((aligned(16))) uint64_t lds[LDS_ELEMS];
        __shared__ __attribute__((aligned(16))) typename A::TwElem ldstw[LDSTW_ELEMS];
        pass_8<A, FWD, CONTIG, XB>(p, g, c, lds, ldstw);
    } else if constexpr (CONTIG) {
        __shared__ __attribute__((aligned(16))) uint64_t lds[LDS_ELEMS];
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
        __shared__ __attribute__((aligned(16))) uint64_t lds[LDS_ELEMS];
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

// The last 8 + KF stages (KF = 2 or 4) in ONE launch: a tile of 4096 contiguous elements holds
// whole blocks of 2^(8 + KF) elements, so the KF stages with strides 128..1024 and the eight-stage
// CONTIG pass meet in LDS instead of HBM (one round trip fewer).  Transforms of 2^10 and 2^12
// elements are done completely this way; longer ones with log2 n = 2 or 4 mod 8 end with it.  `p`
// describes the eight-stage pass (s0 = log2 n - 8, blocks of 256) with the flags of both parts.
//   forward:  HBM -> KF stages in registers -> LDS (tile order) -> pass_8 -> HBM
//   backward: HBM -> pass_8 -> LDS (tile order) -> KF stages (with the n^-1 factor) -> HBM
template <class A, bool FWD, int KF>
__global__ __launch_bounds__(THREADS, 4) void fused_small_kernel(PassParams p, typename A::Ctx c)
{
    static_assert(KF == 2 || KF == 4, "n = 2^10 or 2^12");
    __shared__ __attribute__((aligned(16))) uint64_t lds[LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) typename A::TwElem ldstw[LDSTW_ELEMS];
    using R = Round<A, false, KF, 0, true>;
    const uint64_t tile = blockIdx.x;
    const uint32_t outer_only = PASS_CONDITION_INPUT | PASS_POINTWISE | PASS_FINISH; // belong to the eight-stage side ...
    PassParams pf = p; // ... except input conditioning, which is the first round's job going forward
    pf.logB = p.logB + KF;
    pf.s0 = p.s0 - KF;
    pf.nblocks_total = p.nblocks_total >> KF;
    pf.block_base = p.block_base >> KF;
    pf.flags = (p.flags & ~outer_only) | (FWD ? (p.flags & PASS_CONDITION_INPUT) : 0u);
    PassParams p8 = p;
    p8.flags = p.flags & ~(PASS_SCALE | (FWD ? (uint32_t)PASS_CONDITION_INPUT : 0u));
    const TileGeom<KF> gf(pf, tile);
    const TileGeom<8> g8(p8, tile);
    const typename A::TwElem* tw = static_cast<const typename A::TwElem*>(p.tw);
    // position of element r of sub-transform sigma in the tile, padded as the CONTIG layout pads
    auto at = [&](const R& r, int i) -> uint32_t {
        const uint32_t f = (uint32_t)gf.lane_index(r.sigma, r.r0 + i);
        return f + (f >> 4);
    };
    const bool math = !NTTX_PROF(p.flags & PASS_DEBUG_NO_MATH);

    if (FWD) {
#pragma unroll
        for (int gamma = 0; gamma < R::GROUPS; gamma++) {
            const R r(gf, gamma);
            typename A::Tw w[R::M];
            uint64_t x[R::M];
            load_twiddles<A, KF>(w, tw, r.P, c);
            r.load_global(x, pf, gf, c);
            if (math)
                radix<A, KF, true>(x, RegTw<A, KF>{w}, c, SCALE_NONE);
#pragma unroll
            for (int i = 0; i < R::M; i++)
                lds[at(r, i)] = x[i];
        }
        __syncthreads();
        pass_8<A, true, true, false, true>(p8, g8, c, lds, ldstw);
    } else {
        pass_8<A, false, true, false, true>(p8, g8, c, lds, ldstw);
        __syncthreads();
#pragma unroll 1
        for (int gamma = 0; gamma < R::GROUPS; gamma++) {
            const R r(gf, gamma);
            typename A::Tw w[R::M];
            uint64_t x[R::M];
            load_twiddles<A, KF>(w, tw, r.P, c);
#pragma unroll
            for (int i = 0; i < R::M; i++)
                x[i] = lds[at(r, i)];
            if (math)
                radix<A, KF, false>(x, RegTw<A, KF>{w}, c, scale_mode(p.flags));
            r.store_global(x, pf, gf);
        }
    }
}

// ---- the middle of a negacyclic product in ONE launch ----------------------------------------------
//
// a <- a * b in Z_q[X]/(X^n + 1) is forward(a), forward(b), the coefficient-wise product and one
// backward transform.  The last forward pass and the first backward pass are both the 8-stage CONTIG
// pass on the same tiles, and a thread ends the former holding exactly the 16 elements it needs to
// begin the latter (the layout of round B).  So for cross-batch tiles this kernel runs, per tile:
// the last forward pass of a, the last forward pass of b, the product in registers, and the first
// backward pass, storing to a.  HBM sees two reads and one write of the tile instead of the five
// reads and three writes of three separate launches; b is only read (it is left holding the output
// of its earlier forward passes).  pf: the forward pass (data = a, data2 = b, forward twiddles,
// PASS_FINISH); pb: the backward pass (data = a, backward twiddles).  data2 == data squares.
//
// Both operands' registers are live at once, so three workgroups fit a CU instead of four.
template <class A>
__global__ __launch_bounds__(THREADS, 3) void product_mid_kernel(PassParams pf, PassParams pb, typename A::Ctx c)
{
    __shared__ __attribute__((aligned(16))) uint64_t lds[LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) typename A::TwElem ldstw_f[LDSTW_ELEMS];
    __shared__ __attribute__((aligned(16))) typename A::TwElem ldstw_b[LDSTW_ELEMS];
    using RA = Round<A, true, 4, 4, true, true>;
    using RB = Round<A, true, 4, 4, false, true>;
    using TwElem = typename A::TwElem;
    using Raw = __attribute__((ext_vector_type(sizeof(TwElem) / 4))) uint32_t;
    uint64_t tile = blockIdx.x;
    if (NTTX_PROF(pf.flags & PASS_DEBUG_HOT_TILES))
        tile &= 63;
    const TileGeom<8, true> g(pf, tile);
    const RA ra(g, 0);
    const RB rb(g, 0);
    const bool square = pf.data2 == pf.data;
    const bool math = !NTTX_PROF(pf.flags & PASS_DEBUG_NO_MATH);

    // twiddles of both directions, staged as pass_8 stages those of one: thread t fetches entry
    // j = t & 15 of round B's set t >> 4; threads 0..15 also fetch round A's single set
    const uint32_t set = threadIdx.x >> 4, j = threadIdx.x & 15;
    const uint32_t u = 32 - __clz(j | 1);
    const uint32_t b = j ? j - (1u << (u - 1)) : 0;
    const uint64_t at_b = (((g.twiddle_base(0) << 4) + set) << (u - 1)) + b;
    const uint64_t at_a = (g.twiddle_base(0) << (u - 1)) + b;
    const TwElem* twf = static_cast<const TwElem*>(pf.tw);
    const TwElem* twb = static_cast<const TwElem*>(pb.tw);
    const Raw mine_f = *reinterpret_cast<const Raw*>(twf + at_b), uni_f = *reinterpret_cast<const Raw*>(twf + at_a);
    const Raw mine_b = *reinterpret_cast<const Raw*>(twb + at_b), uni_b = *reinterpret_cast<const Raw*>(twb + at_a);

    // both operands' tiles are requested up front: the barrier below drains outstanding loads anyway
    uint64_t xa[16], xb[16];
    ra.load_global(xa, pf, g, c);
    if (!square) {
        PassParams p2 = pf;
        p2.data = const_cast<uint64_t*>(pf.data2);
        ra.load_global(xb, p2, g, c);
    }
    reinterpret_cast<Raw*>(ldstw_f)[set * TW_SET_STRIDE + j] = mine_f;
    reinterpret_cast<Raw*>(ldstw_b)[set * TW_SET_STRIDE + j] = mine_b;
    if (threadIdx.x < 16) {
        reinterpret_cast<Raw*>(ldstw_f + 16 * TW_SET_STRIDE)[threadIdx.x] = uni_f;
        reinterpret_cast<Raw*>(ldstw_b + 16 * TW_SET_STRIDE)[threadIdx.x] = uni_b;
    }
    const uint32_t own = (threadIdx.x & 15) * TW_SET_STRIDE;
    const LdsTw<A> fa{ldstw_f + 16 * TW_SET_STRIDE, c}, fb{ldstw_f + own, c};
    const LdsTw<A> ba{ldstw_b + 16 * TW_SET_STRIDE, c}, bb{ldstw_b + own, c};
    __syncthreads();

    if (math)
        radix<A, 4, true>(xa, fa, c, SCALE_NONE);
    ra.store_lds(xa, lds);
    __syncthreads();
    rb.load_lds(xa, lds);
    if (math)
        radix<A, 4, true>(xa, fb, c, SCALE_NONE);
    finish_all<A, 16>(xa, c, pf.flags);
    if (!square) {
        if (math)
            radix<A, 4, true>(xb, fa, c, SCALE_NONE);
        __syncthreads(); // every thread has taken a's elements out of the tile
        ra.store_lds(xb, lds);
        __syncthreads();
        rb.load_lds(xb, lds);
        if (math)
            radix<A, 4, true>(xb, fb, c, SCALE_NONE);
        finish_all<A, 16>(xb, c, pf.flags);
#pragma unroll
        for (int i = 0; i < 16; i++)
            xa[i] = A::pointwise(xa[i], xb[i], c);
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++)
            xa[i] = A::pointwise(xa[i], xa[i], c);
    }
    if (math)
        radix<A, 4, false>(xa, bb, c, SCALE_NONE);
    rb.store_lds(xa, lds); // the positions this thread read last: no other thread touches them
    __syncthreads();
    ra.load_lds(xa, lds);
    if (math)
        radix<A, 4, false>(xa, ba, c, scale_mode(pb.flags));
    ra.store_global(xa, pb, g);
}

} // namespace nttx
