// libnttx: plans, pass scheduling and the C ABI declared in include/nttx.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nttx.h"
#include "../host/tables.hpp"
#include "launch.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? NTTX_ENOMEM : NTTX_EHIP,                       \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
    } while (0)

// No exception may cross the C boundary: entry points that allocate host memory run under this.
template <class F>
int guarded(F&& body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try {
            return fail(NTTX_ENOMEM, "out of host memory");
        } catch (...) {
            return NTTX_ENOMEM;
        }
    } catch (...) {
        try {
            return fail(NTTX_EINVAL, "unexpected exception");
        } catch (...) {
            return NTTX_EINVAL;
        }
    }
}

// Host threads that move a caller's pageable slice to and from pinned staging memory.  One
// memcpy stream per core is far below what a PCIe 5 x16 link carries; several in parallel keep
// up with it (measured on the MI355X host: 1 thread 27 GB/s, 4 threads 48 GB/s).
class CopyPool {
    struct Piece {
        char* dst;
        const char* src;
        size_t bytes;
    };
    static constexpr size_t PIECE_BYTES = 2u << 20;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable wake, all_done;
    std::vector<Piece> pieces; // guarded by m, as are the counters below
    size_t next = 0, finished = 0;
    uint64_t generation = 0;
    bool stop = false;

    void drain(std::unique_lock<std::mutex>& lk)
    {
        while (next < pieces.size()) {
            const Piece p = pieces[next++];
            lk.unlock();
            std::memcpy(p.dst, p.src, p.bytes);
            lk.lock();
            if (++finished == pieces.size())
                all_done.notify_all();
        }
    }
    void work()
    {
        std::unique_lock<std::mutex> lk(m);
        uint64_t seen = 0;
        for (;;) {
            wake.wait(lk, [&] { return stop || generation != seen; });
            if (stop)
                return;
            seen = generation;
            drain(lk);
        }
    }
    void add(void* dst, const void* src, size_t bytes)
    {
        for (size_t off = 0; off < bytes; off += PIECE_BYTES)
            pieces.push_back(Piece{static_cast<char*>(dst) + off, static_cast<const char*>(src) + off,
                                   std::min(PIECE_BYTES, bytes - off)});
    }

public:
    explicit CopyPool(int threads)
    {
        for (int i = 1; i < threads; i++) // the calling thread copies too
            workers.emplace_back([this] { work(); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        wake.notify_all();
        for (auto& w : workers)
            w.join();
    }
    CopyPool(const CopyPool&) = delete;
    CopyPool& operator=(const CopyPool&) = delete;

    struct Copy {
        void* dst;
        const void* src;
        size_t bytes;
    };
    // Performs the copies (any may be empty), split over all threads; returns when they are done.
    void run(std::initializer_list<Copy> copies)
    {
        std::unique_lock<std::mutex> lk(m);
        pieces.clear();
        next = finished = 0;
        for (const Copy& c : copies)
            add(c.dst, c.src, c.bytes);
        if (pieces.empty())
            return;
        generation++;
        wake.notify_all();
        drain(lk);
        all_done.wait(lk, [&] { return finished == pieces.size(); });
    }
};

} // namespace

struct nttx_plan {
    int kind = 0;
    uint64_t q = 0, k = 0, n = 0, n_inv = 0;
    uint32_t logn = 0;
    // arithmetic policy of the kernels (arith.hpp)
    enum Policy { SHOUP61, SHOUP61_NEAR, MONT, MONT_WIDE, GOLDILOCKS, GOLDILOCKS_CANONICAL } policy = MONT;
    nttx::MontCtx mctx{};
    nttx::Shoup61Ctx sctx{};
    nttx::GoldilocksCtx gctx{};
    // same with n^-1 replaced by n^-1 * 2^64: the backward transform of a pointwise product,
    // which the Montgomery-kind kernels form as a * b * 2^-64 (nttx_negacyclic_mul_*)
    nttx::MontCtx mctx_mul{};
    nttx::Shoup61Ctx sctx_mul{};
    // Tables in the reference's format (powers_psi_bo / powers_psi_inv_bo: Montgomery form for
    // NTTX_MONT64, plain for NTTX_GOLDILOCKS), as the caller passed them.  Empty when the plan
    // generated its tables on the devices (from_root with n >= DEVICE_TABLES_MIN_N): nttx_plan_tables
    // then rebuilds them on the host from (q, psi, n).
    std::vector<uint64_t> h_fwd, h_inv;
    bool pow2_twiddles = false; // Goldilocks kernels: entries 1..15 of both tables are the standard powers of two
    uint64_t psi = 0;     // root the plan was created from (generated tables only)
    bool exact_k = false; // created by nttx_plan_create_from_root_exact_k
    std::vector<int> devices;
    std::vector<void*> d_fwd, d_inv;           // per device: tables in the reference's index order
    std::vector<void*> d_fwd_perm, d_inv_perm; // per device: last four stages regrouped (permuted_index)
    std::vector<nttx::PassDesc> passes;        // launches of a batched transform, forward order
    std::vector<nttx::PassDesc> split_passes;  // launches of a split transform: never fused, first pass = log2 n mod 8 (or 8) stages
    // Host-slice calls (nttx_*_host) keep one stream and one staging buffer per device between
    // calls, so a single-vector call pays two copies and the kernels, not an allocation.
    struct HostSlot {
        std::mutex lock; // host-slice calls on one device run one at a time
        hipStream_t stream = nullptr;
        void* buf = nullptr;
        size_t cap = 0;
        // Large batches are streamed through pinned memory in chunks (run_staged): STAGES chunks
        // in flight, uploads and downloads on separate streams so both directions of the link work.
        static constexpr int STAGES = 4;
        bool staged_ready = false;
        hipStream_t down = nullptr;
        char* pin_in[STAGES] = {};
        char* pin_out[STAGES] = {};
        void* dev[STAGES] = {};
        hipEvent_t computed[STAGES] = {};
        hipEvent_t downloaded[STAGES] = {};
        std::unique_ptr<CopyPool> pool;
    };
    mutable std::vector<std::unique_ptr<HostSlot>> host_slots;
};

namespace {

using namespace nttx;

// Splits log2(n) stages into passes: an optional first pass with log2(n) mod 8 stages,
// then passes of 8 stages; the last one holds the stride-1 stage (CONTIG).
std::vector<PassDesc> make_passes(uint32_t logn)
{
    auto split = [](int k, int& k1, int& k2) {
        k1 = k >= 4 ? 4 : k;
        k2 = k - k1;
    };
    std::vector<PassDesc> out;
    if (logn <= 8) {
        PassDesc d{0, 0, true, 0};
        split((int)logn, d.k1, d.k2);
        out.push_back(d);
        return out;
    }
    uint32_t s0 = 0;
    const int r = (int)(logn % 8);
    if (r) {
        PassDesc d{0, 0, false, 0};
        split(r, d.k1, d.k2);
        out.push_back(d);
        s0 += (uint32_t)r;
    }
    while (s0 < logn) {
        PassDesc d{4, 4, s0 + 8 == logn, s0};
        out.push_back(d);
        s0 += 8;
    }
    return out;
}

// Launches of a batched transform.  When log2 n = 2 or 4 mod 8 the short first pass of
// make_passes is dropped: 8-stage STRIDED passes from stage 1 on, and the last 8 + (log2 n mod 8)
// stages in one fused launch -- one HBM round trip fewer (NTTX_NO_FUSED_SMALL keeps make_passes).
//
// Odd log2 n > 8 (plans of the corrected constructor only): a first STRIDED pass with the largest odd
// stage count r <= 7 that leaves a remainder the even schedule handles (a multiple of 8, or 2 or 4
// more), then that schedule on the remaining stages: 2^17 = 7 + (8 + 2 fused), 2^21 = 5 + 8 + 8.
std::vector<PassDesc> make_schedule(uint32_t logn)
{
    static const bool no_fused = std::getenv("NTTX_NO_FUSED_SMALL") != nullptr;
    uint32_t first = 0; // stages of the leading odd pass
    if (logn > 8 && logn % 2 != 0) {
        first = 1;
        for (uint32_t r = 7; r > 1; r -= 2) {
            const uint32_t m = logn >= r + 8 ? (logn - r) % 8 : 1;
            if (m == 0 || m == 2 || m == 4) {
                first = r;
                break;
            }
        }
    }
    const uint32_t rest = logn - first, kf = rest % 8;
    const bool fused_tail = (kf == 2 || kf == 4) && rest >= 10 && !no_fused;
    if (first == 0 && !fused_tail)
        return make_passes(logn);
    std::vector<PassDesc> out;
    if (first)
        out.push_back(PassDesc{(int)std::min(first, 4u), first > 4 ? (int)first - 4 : 0, false, 0});
    if (fused_tail) {
        for (uint32_t s0 = first; s0 + 8 + kf < logn; s0 += 8)
            out.push_back(PassDesc{4, 4, false, s0});
        PassDesc tail{4, 4, true, logn - 8};
        tail.fused = (int)kf;
        out.push_back(tail);
    } else {
        for (PassDesc d : make_passes(rest)) { // the schedule of a transform of 2^rest elements, `first` stages later
            d.s0 += first;
            out.push_back(d);
        }
    }
    return out;
}

int device_slot(const nttx_plan* plan, int device)
{
    for (size_t i = 0; i < plan->devices.size(); i++)
        if (plan->devices[i] == device)
            return (int)i;
    return -1;
}

// Environment switches, read once.  NTTX_NO_BUFFER_ADDRESSING selects the plain-pointer addressing
// path of the kernels (an alternative code path the tests cover).  Two tuning knobs that leave the
// results unchanged are looked at when NTTX_DEBUG is set:
//   NTTX_XCD_GROUP=<G>         runs of G tiles per XCD on every pass (0 = plain order)
//   NTTX_CHUNK_MIB=<m>         all passes over m MiB of the batch before the next m MiB
// The profiling build (-DNTTX_PROFILING: libnttx_prof.so, whose nttx_version() says so) also reads,
// again only when NTTX_DEBUG is set, the switches that skip work and so produce wrong results:
//   NTTX_DEBUG_NO_MATH=1|A|B   skip the butterflies (of one register round), keep all traffic
//   NTTX_DEBUG_NO_TW=1         with NO_MATH: skip the twiddle loads too
//   NTTX_DEBUG_NO_FINISH=1     skip the canonicalising end of the forward transform
//   NTTX_DEBUG_HOT_TILES=1     every workgroup works on one of the first 64 tiles (all traffic from L2):
//                              the pass time with HBM out of the picture (but 1000 workgroups store to the same lines)
//   NTTX_DEBUG_NO_MEM=1        no global loads or stores of data at all (made-up inputs, results dropped):
//                              the pass time with the memory system out of the picture
//   NTTX_DEBUG_EXTRA_LDS=<b>   unused dynamic LDS per workgroup, to lower occupancy (launch_impl.hpp)
// The shipped library contains none of the latter: no environment makes it return wrong results.
struct Switches {
    bool no_buffer = false;
    long xcd_group = -1;
    size_t chunk_bytes = 0;
#ifdef NTTX_PROFILING
    bool no_finish = false;
    uint32_t debug_flags = 0;
#endif
};
const Switches& switches()
{
    static const Switches v = [] {
        Switches s;
        s.no_buffer = std::getenv("NTTX_NO_BUFFER_ADDRESSING") != nullptr;
        if (std::getenv("NTTX_DEBUG") == nullptr)
            return s;
        if (const char* e = std::getenv("NTTX_XCD_GROUP"))
            s.xcd_group = std::atol(e);
        if (const char* e = std::getenv("NTTX_CHUNK_MIB")) {
            const long mib = std::atol(e);
            s.chunk_bytes = mib <= 0 ? (size_t)0 : (size_t)mib << 20;
        }
#ifdef NTTX_PROFILING
        if (const char* e = std::getenv("NTTX_DEBUG_NO_MATH"))
            s.debug_flags |= e[0] == 'A' ? (uint32_t)PASS_DEBUG_NO_MATH_A : e[0] == 'B' ? (uint32_t)PASS_DEBUG_NO_MATH_B
                                                                                        : (uint32_t)PASS_DEBUG_NO_MATH;
        if (std::getenv("NTTX_DEBUG_NO_TW"))
            s.debug_flags |= PASS_DEBUG_NO_TW;
        if (std::getenv("NTTX_DEBUG_HOT_TILES"))
            s.debug_flags |= PASS_DEBUG_HOT_TILES;
        if (std::getenv("NTTX_DEBUG_NO_MEM"))
            s.debug_flags |= PASS_DEBUG_NO_MEM;
        s.no_finish = std::getenv("NTTX_DEBUG_NO_FINISH") != nullptr;
#endif
        return s;
    }();
    return v;
}
// The profiling switches that act through pass flags; shared by every launch path.
void apply_debug_flags(PassParams& p)
{
#ifdef NTTX_PROFILING
    const Switches& sw = switches();
    p.flags |= sw.debug_flags;
    if (sw.no_finish)
        p.flags &= ~(uint32_t)PASS_FINISH;
#else
    (void)p;
#endif
}

// Launches pass `d` with parameters `p` (data, geometry and flags already set) over `tiles` tiles.
// `product`: the backward transform of a pointwise product (contexts with n^-1 * 2^64).
int launch_pass(const nttx_plan* plan, int slot, const PassDesc& d, PassParams p, uint64_t tiles, bool fwd,
                hipStream_t stream, bool product = false)
{
    if (tiles == 0 || tiles > 0x7fffffffull)
        return fail(NTTX_EINVAL, "element count is out of range for one launch");
    p.tw = fwd ? plan->d_fwd[slot] : plan->d_inv[slot];
    p.tw_perm = fwd ? plan->d_fwd_perm[slot] : plan->d_inv_perm[slot];
    const Switches& sw = switches();
    // Every element a workgroup touches lies within 2^31 bytes of its tile's first element when a
    // block (or, for cross-batch tiles, 16 transforms) spans at most 2^28 elements.
    if (!sw.no_buffer && ((p.flags & PASS_CROSS_BATCH) ? plan->logn + 4 <= 28 : p.logB <= 28))
        p.flags |= PASS_OFFSET32;
    // STRIDED passes: all tiles of a block (they share its rows, 128 bytes each) go to one XCD.
    // Measured on the headline workload: 0.80 ms instead of 0.83 ms for the memory-bound forward
    // pass; no gain for CONTIG tiles.
    uint64_t group = 0;
    if (sw.xcd_group >= 0)
        group = (uint64_t)sw.xcd_group;
    else if (!d.contig && p.logB >= (uint32_t)TILE_LOG)
        group = 1ull << (p.logB - TILE_LOG);
    if (group > 0 && group <= (1u << 20) && tiles % (8 * group) == 0)
        p.xcd_group = (uint32_t)group;
    else if (sw.xcd_group < 0 && group > 256 && tiles % (8 * 256) == 0)
        p.xcd_group = 256; // a block with more tiles than the launch can group whole (one 2^28 transform): runs of
                           // 256 tiles per XCD still take the last backward pass from 1.04 to 0.91 ms
    if (plan->pow2_twiddles && p.s0 == 0 && !d.contig && d.k1 == 4 && d.k2 == 4)
        p.flags |= PASS_POW2_TWIDDLES; // this pass's first four stages read table entries 1..15
    apply_debug_flags(p);
    hipError_t e;
    switch (plan->policy) {
    case nttx_plan::GOLDILOCKS: e = launch_pass_policy<Goldilocks>(fwd, d, p, plan->gctx, (unsigned)tiles, stream); break;
    case nttx_plan::GOLDILOCKS_CANONICAL:
        e = launch_pass_policy<GoldilocksCanonical>(fwd, d, p, plan->gctx, (unsigned)tiles, stream);
        break;
    case nttx_plan::MONT_WIDE: e = launch_pass_policy<Mont<true>>(fwd, d, p, product ? plan->mctx_mul : plan->mctx, (unsigned)tiles, stream); break;
    case nttx_plan::MONT: e = launch_pass_policy<Mont<false>>(fwd, d, p, product ? plan->mctx_mul : plan->mctx, (unsigned)tiles, stream); break;
    case nttx_plan::SHOUP61_NEAR:
        e = launch_pass_policy<Shoup61Near>(fwd, d, p, product ? plan->sctx_mul : plan->sctx, (unsigned)tiles, stream);
        break;
    default: e = launch_pass_policy<Shoup61>(fwd, d, p, product ? plan->sctx_mul : plan->sctx, (unsigned)tiles, stream); break;
    }
    if (e != hipSuccess)
        return fail(NTTX_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return NTTX_OK;
}

// The last 8 + kf stages of `batch` vectors (kf = log2 n mod 8 = 2 or 4) in one launch
// (fused_small_kernel); for n = 2^10 and 2^12 that is the whole transform.
// `mul_with`: multiply by that batch element by element first (backward transform of a product).
int launch_fused(const nttx_plan* plan, int slot, uint64_t* dptr, size_t batch, bool fwd, hipStream_t stream,
                 const uint64_t* mul_with, bool lazy)
{
    const int kf = plan->passes.back().fused;
    const bool whole = plan->logn == (uint32_t)(8 + kf); // no stage precedes (forward) / follows (backward) this launch
    const uint64_t tiles = (((uint64_t)batch << plan->logn) + TILE - 1) / TILE;
    if (tiles == 0 || tiles > 0x7fffffffull)
        return fail(NTTX_EINVAL, "element count is out of range for one launch");
    PassParams p{};
    p.data = dptr;
    p.data2 = mul_with;
    p.tw = fwd ? plan->d_fwd[slot] : plan->d_inv[slot];
    p.tw_perm = fwd ? plan->d_fwd_perm[slot] : plan->d_inv_perm[slot];
    p.nblocks_total = (uint64_t)batch << (plan->logn - 8);
    p.block_base = 0;
    p.logB = 8;
    p.s0 = plan->logn - 8;
    // a tile of the fused launch is 4096 contiguous elements: lane offsets always fit 32 bits
    p.flags = (switches().no_buffer ? 0u : (uint32_t)PASS_OFFSET32) | (mul_with ? (uint32_t)PASS_POINTWISE : 0u) |
              (lazy ? (uint32_t)PASS_LAZY : 0u);
    if (fwd)
        p.flags |= PASS_FINISH | (whole ? (uint32_t)PASS_CONDITION_INPUT : 0u);
    else
        p.flags |= PASS_CONDITION_INPUT | (whole ? (uint32_t)PASS_SCALE : 0u);
    apply_debug_flags(p);
    const bool product = mul_with != nullptr;
    hipError_t e;
    switch (plan->policy) {
    case nttx_plan::GOLDILOCKS: e = launch_fused_policy<Goldilocks>(fwd, kf, p, plan->gctx, (unsigned)tiles, stream); break;
    case nttx_plan::GOLDILOCKS_CANONICAL:
        e = launch_fused_policy<GoldilocksCanonical>(fwd, kf, p, plan->gctx, (unsigned)tiles, stream);
        break;
    case nttx_plan::MONT_WIDE:
        e = launch_fused_policy<Mont<true>>(fwd, kf, p, product ? plan->mctx_mul : plan->mctx, (unsigned)tiles, stream);
        break;
    case nttx_plan::MONT:
        e = launch_fused_policy<Mont<false>>(fwd, kf, p, product ? plan->mctx_mul : plan->mctx, (unsigned)tiles, stream);
        break;
    case nttx_plan::SHOUP61_NEAR:
        e = launch_fused_policy<Shoup61Near>(fwd, kf, p, product ? plan->sctx_mul : plan->sctx, (unsigned)tiles, stream);
        break;
    default:
        e = launch_fused_policy<Shoup61>(fwd, kf, p, product ? plan->sctx_mul : plan->sctx, (unsigned)tiles, stream);
        break;
    }
    if (e != hipSuccess)
        return fail(NTTX_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return NTTX_OK;
}

// Flags of the pass at forward-order position `fi` of `np` when running direction `fwd`.
uint32_t pass_flags(const PassDesc& d, size_t fi, size_t np, bool fwd)
{
    uint32_t flags = 0;
    if (fwd ? fi == 0 : fi + 1 == np)
        flags |= PASS_CONDITION_INPUT; // first pass to touch the caller's data
    if (!fwd && d.s0 == 0)
        flags |= PASS_SCALE;
    if (fwd && fi + 1 == np)
        flags |= PASS_FINISH;
    return flags;
}

// Enqueues the passes of one transform direction for `batch` vectors at dptr.
//
// The pass that holds the last eight stages runs on cross-batch tiles (the same 256-element block
// of 16 consecutive transforms, so a workgroup shares its twiddles) for the largest multiple of 16
// transforms, and on plain contiguous tiles for the rest of the batch.
int enqueue_passes(const nttx_plan* plan, int slot, uint64_t* dptr, size_t batch, bool fwd, hipStream_t stream,
                   int only_pass, const uint64_t* mul_with, bool lazy)
{
    static const bool no_xb = std::getenv("NTTX_NO_CROSS_BATCH") != nullptr;
    const size_t np = plan->passes.size();
    for (size_t i = 0; i < np; i++) {
        if (only_pass >= 0 && (size_t)only_pass != i)
            continue;
        const size_t fi = fwd ? i : np - 1 - i;
        const PassDesc& d = plan->passes[fi];
        if (d.fused) { // sets its own flags: it is the last forward / first backward launch
            int rc = launch_fused(plan, slot, dptr, batch, fwd, stream, fwd ? nullptr : mul_with, lazy);
            if (rc != NTTX_OK)
                return rc;
            continue;
        }
        const bool xb_shape = d.contig && d.k1 == 4 && d.k2 == 4 && !no_xb;
        const size_t xb_batch = xb_shape ? batch & ~(size_t)15 : 0;
        for (int part = 0; part < 2; part++) {
            const size_t first = part == 0 ? 0 : xb_batch;
            const size_t count = part == 0 ? xb_batch : batch - xb_batch;
            if (count == 0)
                continue;
            const uint64_t total = (uint64_t)count << plan->logn;
            PassParams p{};
            p.data = dptr + ((uint64_t)first << plan->logn);
            p.nblocks_total = (uint64_t)count << d.s0;
            p.block_base = 0;
            p.logB = plan->logn - d.s0;
            p.s0 = d.s0;
            p.flags = pass_flags(d, fi, np, fwd) | (part == 0 ? (uint32_t)PASS_CROSS_BATCH : 0u) |
                      (lazy ? (uint32_t)PASS_LAZY : 0u);
            if (mul_with && i == 0) { // the first backward pass multiplies by the other transform first
                p.data2 = mul_with + ((uint64_t)first << plan->logn);
                p.flags |= PASS_POINTWISE;
            }
            int rc = launch_pass(plan, slot, d, p, (total + TILE - 1) / TILE, fwd, stream, mul_with != nullptr);
            if (rc != NTTX_OK)
                return rc;
        }
    }
    return NTTX_OK;
}

// ---- one transform split over several shards ---------------------------------------------------
//
// The first forward pass (f stages, strides n/2 .. n/2^f) only couples elements whose indices
// agree modulo n/2^f; after it the transform consists of 2^f independent contiguous blocks of
// n/2^f elements (the reference's index structure, ntt.rs:129-171: stage u works inside blocks
// of n/2^(u-1) and its twiddle index is 2^(u-1) + block).  With g = 2^k <= 2^f shards:
//   column layout: shard d holds, of every one of the 2^f rows of n/2^f elements, the columns
//                  [d*W, (d+1)*W), W = n/(2^f g), as a [2^f][W] array.  The first pass on it is
//                  the first pass of a transform of length n/g with the same twiddles.
//   block layout:  shard c holds elements [c*n/g, (c+1)*n/g): 2^f/g whole blocks.  All later
//                  passes run on it with the twiddle block index offset by the shard.
// Between the two every shard sends every other shard one [R][W] piece, R = 2^f/g rows.
// Forward: column layout -> first pass -> exchange -> block layout -> remaining passes.
// Backward mirrors it (ntt.rs:185-229): block layout -> all but the last pass -> exchange ->
// column layout -> the pass with the first f stages (and the n^-1 factor).
struct ShardGeom {
    uint32_t k = 0, f = 0;   // log2(shards), stages before the exchange
    uint64_t local = 0;      // elements per shard
    uint64_t rows = 0;       // 2^f
    uint64_t W = 0;          // columns per shard in the column layout
    uint64_t R = 0;          // rows per destination shard
};

int shard_geometry(const nttx_plan* plan, int n_shards, ShardGeom& g)
{
    if (!plan)
        return fail(NTTX_EINVAL, "plan is NULL");
    if (n_shards < 1 || n_shards > 64 || (n_shards & (n_shards - 1)) != 0)
        return fail(NTTX_EINVAL, "number of shards must be a power of two in [1, 64]");
    if (plan->split_passes.size() < 2)
        return fail(NTTX_EINVAL, "n is too small to split (needs n >= 2^10)");
    if (plan->logn % 2 != 0)
        return fail(NTTX_EINVAL, "transforms with odd log2(n) are not split over shards");
    g.k = ntt::host::ilog2((uint64_t)n_shards);
    g.f = (uint32_t)(plan->split_passes[0].k1 + plan->split_passes[0].k2);
    if (g.f < g.k)
        return fail(NTTX_EINVAL, "too many shards for this n: the first pass has fewer stages than log2(shards)");
    if (plan->logn < g.f + g.k + 4 || plan->logn < g.k + TILE_LOG)
        return fail(NTTX_EINVAL, "n is too small for this number of shards");
    g.local = plan->n >> g.k;
    g.rows = 1ull << g.f;
    g.W = plan->n >> (g.f + g.k);
    g.R = g.rows >> g.k;
    return NTTX_OK;
}

// Enqueues on one shard's buffer the passes of `phase` (0: the first forward pass, column layout;
// 1: all other passes, block layout) for direction `fwd`.
int enqueue_shard_phase(const nttx_plan* plan, int slot, uint64_t* dptr, bool fwd, int phase, int shard,
                        const ShardGeom& g, hipStream_t stream)
{
    const size_t np = plan->split_passes.size();
    const uint64_t tiles = g.local / TILE;
    for (size_t i = 0; i < np; i++) {
        const size_t fi = fwd ? i : np - 1 - i;
        const PassDesc& d = plan->split_passes[fi];
        if ((phase == 0) != (fi == 0))
            continue;
        PassParams p{};
        p.data = dptr;
        p.s0 = d.s0;
        if (fi == 0) {
            p.nblocks_total = 1;
            p.block_base = 0;
            p.logB = plan->logn - g.k;
        } else {
            p.nblocks_total = (1ull << d.s0) >> g.k;
            p.block_base = (uint64_t)shard * p.nblocks_total;
            p.logB = plan->logn - d.s0;
        }
        p.flags = pass_flags(d, fi, np, fwd);
        int rc = launch_pass(plan, slot, d, p, tiles, fwd, stream);
        if (rc != NTTX_OK)
            return rc;
    }
    return NTTX_OK;
}

// Whole split transform driven from one process: shard i lives on plan device i % n_devices
// (with one device all shards share it, which exercises the same code on a single GPU).
int run_distributed(const nttx_plan* plan, uint64_t* host, bool fwd, int n_shards)
{
    if (!plan)
        return fail(NTTX_EINVAL, "plan is NULL");
    if (!host)
        return fail(NTTX_EINVAL, "data is NULL");
    const int ndev = (int)plan->devices.size();
    const int gs = n_shards == 0 ? ndev : n_shards; // negative values fail the geometry check
    ShardGeom g;
    int rc = shard_geometry(plan, gs, g);
    if (rc != NTTX_OK)
        return rc;
    struct Shard {
        int dev = 0, slot = 0;
        hipStream_t s = nullptr;
        hipEvent_t ready = nullptr;              // this shard's passes before the exchange have run
        uint64_t *col = nullptr, *blk = nullptr; // column-layout and block-layout buffers
    };
    std::vector<Shard> sh((size_t)gs);
    int prev = 0;
    (void)hipGetDevice(&prev);
    const size_t bytes = g.local * sizeof(uint64_t);
    const size_t wbytes = g.W * sizeof(uint64_t);
    const size_t row_pitch = (plan->n >> g.f) * sizeof(uint64_t); // host pitch between rows

    auto sync_all = [&]() -> int {
        for (auto& x : sh) {
            HIP_TRY(hipSetDevice(x.dev));
            HIP_TRY(hipStreamSynchronize(x.s));
        }
        return NTTX_OK;
    };
    // piece (rho, d) of shard c's block layout <-> rows c*R + rho of shard d's column layout.
    // Copies run on the receiver's stream, each behind the sender's `ready` event only: a shard's
    // pieces start to move as soon as that shard is done, and the receiver's next passes follow its
    // copies in stream order -- no host synchronisation between the phases.
    auto exchange = [&](bool to_blocks) -> int {
        for (int c = 0; c < gs; c++)
            for (int d = 0; d < gs; d++) {
                Shard& owner = to_blocks ? sh[(size_t)c] : sh[(size_t)d];
                Shard& sender = to_blocks ? sh[(size_t)d] : sh[(size_t)c];
                HIP_TRY(hipSetDevice(owner.dev));
                if (&owner != &sender)
                    HIP_TRY(hipStreamWaitEvent(owner.s, sender.ready, 0));
                for (uint64_t rho = 0; rho < g.R; rho++) {
                    uint64_t* in_blk = sh[(size_t)c].blk + (rho * (uint64_t)gs + (uint64_t)d) * g.W;
                    uint64_t* in_col = sh[(size_t)d].col + ((uint64_t)c * g.R + rho) * g.W;
                    if (to_blocks)
                        HIP_TRY(hipMemcpyPeerAsync(in_blk, sh[(size_t)c].dev, in_col, sh[(size_t)d].dev, wbytes, owner.s));
                    else
                        HIP_TRY(hipMemcpyPeerAsync(in_col, sh[(size_t)d].dev, in_blk, sh[(size_t)c].dev, wbytes, owner.s));
                }
            }
        return NTTX_OK;
    };
    auto body = [&]() -> int {
        for (int i = 0; i < gs; i++) {
            Shard& x = sh[(size_t)i];
            x.slot = i % ndev;
            x.dev = plan->devices[(size_t)x.slot];
            HIP_TRY(hipSetDevice(x.dev));
            HIP_TRY(hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&x.ready, hipEventDisableTiming));
            HIP_TRY(hipMalloc(&x.col, bytes));
            HIP_TRY(hipMalloc(&x.blk, bytes));
        }
        int r;
        if (fwd) {
            for (int d = 0; d < gs; d++) {
                Shard& x = sh[(size_t)d];
                HIP_TRY(hipSetDevice(x.dev));
                HIP_TRY(hipMemcpy2DAsync(x.col, wbytes, host + (uint64_t)d * g.W, row_pitch, wbytes, g.rows,
                                         hipMemcpyHostToDevice, x.s));
                if ((r = enqueue_shard_phase(plan, x.slot, x.col, true, 0, d, g, x.s)) != NTTX_OK)
                    return r;
                HIP_TRY(hipEventRecord(x.ready, x.s));
            }
            if ((r = exchange(true)) != NTTX_OK)
                return r;
            for (int c = 0; c < gs; c++) {
                Shard& x = sh[(size_t)c];
                HIP_TRY(hipSetDevice(x.dev));
                if ((r = enqueue_shard_phase(plan, x.slot, x.blk, true, 1, c, g, x.s)) != NTTX_OK)
                    return r;
                HIP_TRY(hipMemcpyAsync(host + (uint64_t)c * g.local, x.blk, bytes, hipMemcpyDeviceToHost, x.s));
            }
        } else {
            for (int c = 0; c < gs; c++) {
                Shard& x = sh[(size_t)c];
                HIP_TRY(hipSetDevice(x.dev));
                HIP_TRY(hipMemcpyAsync(x.blk, host + (uint64_t)c * g.local, bytes, hipMemcpyHostToDevice, x.s));
                if ((r = enqueue_shard_phase(plan, x.slot, x.blk, false, 1, c, g, x.s)) != NTTX_OK)
                    return r;
                HIP_TRY(hipEventRecord(x.ready, x.s));
            }
            if ((r = exchange(false)) != NTTX_OK)
                return r;
            for (int d = 0; d < gs; d++) {
                Shard& x = sh[(size_t)d];
                HIP_TRY(hipSetDevice(x.dev));
                if ((r = enqueue_shard_phase(plan, x.slot, x.col, false, 0, d, g, x.s)) != NTTX_OK)
                    return r;
                HIP_TRY(hipMemcpy2DAsync(host + (uint64_t)d * g.W, row_pitch, x.col, wbytes, wbytes, g.rows,
                                         hipMemcpyDeviceToHost, x.s));
            }
        }
        return sync_all();
    };
    rc = body();
    for (auto& x : sh) {
        if (!x.s && !x.col && !x.blk && !x.ready)
            continue;
        (void)hipSetDevice(x.dev);
        if (x.s)
            (void)hipStreamSynchronize(x.s);
    }
    for (auto& x : sh) { // only after every stream has drained: streams wait on other shards' events
        if (!x.s && !x.col && !x.blk && !x.ready)
            continue;
        (void)hipSetDevice(x.dev);
        if (x.s)
            (void)hipStreamDestroy(x.s);
        if (x.ready)
            (void)hipEventDestroy(x.ready);
        if (x.col)
            (void)hipFree(x.col);
        if (x.blk)
            (void)hipFree(x.blk);
    }
    (void)hipSetDevice(prev);
    return rc;
}

// Optional chunking of a batch (NTTX_DEBUG=1 NTTX_CHUNK_MIB=<m>): all passes run over m MiB of the
// batch before the next m MiB, so that a pass re-reads what the previous one wrote while it may
// still sit in the 256 MiB Infinity Cache.  Measured on MI355X this gains nothing at 256 MiB and
// loses at smaller chunks (more launch boundaries and partial waves of workgroups), so the default
// is the whole batch per launch.
size_t chunk_bytes() { return switches().chunk_bytes; }

// Enqueues one transform of `batch` vectors at dptr on the current device.
// `lazy`: outputs in the arithmetic policy's lazy range instead of canonical residues.
int enqueue(const nttx_plan* plan, int slot, uint64_t* dptr, size_t batch, bool fwd, hipStream_t stream,
            int only_pass = -1, const uint64_t* mul_with = nullptr, bool lazy = false)
{
    if (batch == 0)
        return NTTX_OK;
    const uint64_t total = (uint64_t)batch << plan->logn;
    if ((total >> plan->logn) != batch || total > (1ull << 40))
        return fail(NTTX_EINVAL, "batch * n is too large");
    size_t per = batch;
    const size_t cb = chunk_bytes();
    if (only_pass < 0 && plan->passes.size() > 1 && cb != 0) {
        per = cb >> (plan->logn + 3);
        if (per == 0)
            per = 1;
    }
    for (size_t first = 0; first < batch; first += per) {
        const size_t count = std::min(per, batch - first);
        int rc = enqueue_passes(plan, slot, dptr + ((uint64_t)first << plan->logn), count, fwd, stream, only_pass,
                                mul_with ? mul_with + ((uint64_t)first << plan->logn) : nullptr, lazy);
        if (rc != NTTX_OK)
            return rc;
    }
    return NTTX_OK;
}

// Whether the product of `batch` pairs takes product_mid_kernel for its cross-batch part: the last
// forward pass is the 8-stage CONTIG pass of a longer transform (log2 n = 16, 24, ...).
// NTTX_NO_PRODUCT_FUSION=1 keeps the three separate launches (A/B measurements).
bool product_mid_applies(const nttx_plan* plan, size_t batch)
{
    static const bool off = std::getenv("NTTX_NO_PRODUCT_FUSION") != nullptr || std::getenv("NTTX_NO_CROSS_BATCH") != nullptr;
    const size_t np = plan->passes.size();
    if (off || np < 2 || batch < 16)
        return false;
    const PassDesc& d = plan->passes[np - 1];
    return d.contig && d.k1 == 4 && d.k2 == 4 && !d.fused;
}

// product_mid_kernel on `count` (a multiple of 16) pairs at a, b.
int launch_product_mid(const nttx_plan* plan, int slot, uint64_t* a, const uint64_t* b, size_t count, hipStream_t stream)
{
    const size_t np = plan->passes.size();
    const PassDesc& d = plan->passes[np - 1];
    const uint64_t tiles = ((uint64_t)count << plan->logn) / TILE;
    if (tiles == 0 || tiles > 0x7fffffffull)
        return fail(NTTX_EINVAL, "element count is out of range for one launch");
    PassParams pf{};
    pf.data = a;
    pf.data2 = b;
    pf.tw = plan->d_fwd[slot];
    pf.tw_perm = plan->d_fwd_perm[slot];
    pf.nblocks_total = (uint64_t)count << d.s0;
    pf.block_base = 0;
    pf.logB = plan->logn - d.s0;
    pf.s0 = d.s0;
    // The transforms only need to reach the range the product accepts: for q < 2^61 operands below 2q
    // give hi(a * b) < q, so REDC still returns the canonical residue and the last conditional
    // subtraction of both transforms is skipped (policies without a lazy range ignore the flag).
    pf.flags = PASS_FINISH | PASS_LAZY | PASS_CROSS_BATCH;
    if (!switches().no_buffer && plan->logn + 4 <= 28)
        pf.flags |= PASS_OFFSET32;
    apply_debug_flags(pf);
    PassParams pb = pf;
    pb.data2 = nullptr;
    pb.tw = plan->d_inv[slot];
    pb.tw_perm = plan->d_inv_perm[slot];
    pb.flags &= ~(uint32_t)(PASS_FINISH | PASS_LAZY); // the product is canonical: no input conditioning either
    hipError_t e;
    switch (plan->policy) {
    case nttx_plan::GOLDILOCKS: e = launch_product_mid_policy<Goldilocks>(pf, pb, plan->gctx, (unsigned)tiles, stream); break;
    case nttx_plan::GOLDILOCKS_CANONICAL:
        e = launch_product_mid_policy<GoldilocksCanonical>(pf, pb, plan->gctx, (unsigned)tiles, stream);
        break;
    case nttx_plan::MONT_WIDE: e = launch_product_mid_policy<Mont<true>>(pf, pb, plan->mctx_mul, (unsigned)tiles, stream); break;
    case nttx_plan::MONT: e = launch_product_mid_policy<Mont<false>>(pf, pb, plan->mctx_mul, (unsigned)tiles, stream); break;
    case nttx_plan::SHOUP61_NEAR: e = launch_product_mid_policy<Shoup61Near>(pf, pb, plan->sctx_mul, (unsigned)tiles, stream); break;
    default: e = launch_product_mid_policy<Shoup61>(pf, pb, plan->sctx_mul, (unsigned)tiles, stream); break;
    }
    if (e != hipSuccess)
        return fail(NTTX_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return NTTX_OK;
}

// a <- a * b in Z_q[X]/(X^n + 1) for `batch` pairs: forward(a), forward(b), then the backward
// transform of a whose first pass multiplies by b element by element.  b == a squares (each tile
// of the first backward pass is read completely before it is written).  b is scratch: it ends
// up holding its forward transform, or -- where product_mid_kernel ran -- the output of the
// forward passes before the last.
int enqueue_product(const nttx_plan* plan, int slot, uint64_t* a, uint64_t* b, size_t batch, hipStream_t stream)
{
    if (batch == 0)
        return NTTX_OK;
    const uint64_t total = (uint64_t)batch << plan->logn;
    if ((total >> plan->logn) != batch || total > (1ull << 40))
        return fail(NTTX_EINVAL, "batch * n is too large");
    if (!product_mid_applies(plan, batch)) {
        int rc = enqueue(plan, slot, a, batch, true, stream);
        if (rc == NTTX_OK && b != a)
            rc = enqueue(plan, slot, b, batch, true, stream);
        if (rc == NTTX_OK)
            rc = enqueue(plan, slot, a, batch, false, stream, -1, b);
        return rc;
    }
    const int np = (int)plan->passes.size();
    const size_t mid = batch & ~(size_t)15, rest = batch - mid; // cross-batch tiles hold 16 transforms
    uint64_t* const a_rest = a + ((uint64_t)mid << plan->logn);
    uint64_t* const b_rest = b + ((uint64_t)mid << plan->logn);
    int rc = NTTX_OK;
    for (int i = 0; i + 1 < np && rc == NTTX_OK; i++) {
        rc = enqueue_passes(plan, slot, a, batch, true, stream, i, nullptr, false);
        if (rc == NTTX_OK && b != a)
            rc = enqueue_passes(plan, slot, b, batch, true, stream, i, nullptr, false);
    }
    if (rc == NTTX_OK)
        rc = launch_product_mid(plan, slot, a, b, mid, stream);
    if (rc == NTTX_OK && rest) { // fewer than 16 pairs left over: the separate launches
        rc = enqueue_passes(plan, slot, a_rest, rest, true, stream, np - 1, nullptr, false);
        if (rc == NTTX_OK && b != a)
            rc = enqueue_passes(plan, slot, b_rest, rest, true, stream, np - 1, nullptr, false);
        if (rc == NTTX_OK)
            rc = enqueue_passes(plan, slot, a_rest, rest, false, stream, 0, b_rest, false);
    }
    // `b` here only selects the product's contexts (n^-1 * 2^64): pass 0 alone multiplies by it
    for (int i = 1; i < np && rc == NTTX_OK; i++)
        rc = enqueue_passes(plan, slot, a, batch, false, stream, i, b, false);
    return rc;
}

// Staging buffers up to this size stay allocated between host-slice calls.
constexpr size_t HOST_SLOT_KEEP_BYTES = 256ull << 20;
// Calls that move at least STAGE_MIN_BYTES per device stream the batch through pinned memory in
// chunks of whole transforms of at most STAGE_CHUNK_BYTES (transforms larger than that take the
// plain path), HostSlot::STAGES chunks in flight.  The runtime's own copies from pageable memory
// run one direction after the other at 28 GB/s up and 17 GB/s down on the MI355X host.
constexpr size_t STAGE_CHUNK_BYTES = 16ull << 20;
constexpr size_t STAGE_MIN_BYTES = 16ull << 20;

int copy_threads()
{
    static const int v = [] {
        const char* e = std::getenv("NTTX_HOST_COPY_THREADS");
        long t = e ? std::atol(e) : 8;
        return (int)std::min(64L, std::max(1L, t));
    }();
    return v;
}

// Pinned chunks, device chunks, events and copy threads of a slot; kept until the plan goes.
// The slot's device is current and its lock is held.
int ensure_staging(nttx_plan::HostSlot& s)
{
    if (s.staged_ready)
        return NTTX_OK;
    if (!s.down)
        HIP_TRY(hipStreamCreateWithFlags(&s.down, hipStreamNonBlocking));
    for (int i = 0; i < nttx_plan::HostSlot::STAGES; i++) {
        if (!s.pin_in[i])
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.pin_in[i]), STAGE_CHUNK_BYTES, hipHostMallocDefault));
        if (!s.pin_out[i])
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.pin_out[i]), STAGE_CHUNK_BYTES, hipHostMallocDefault));
        if (!s.dev[i])
            HIP_TRY(hipMalloc(&s.dev[i], STAGE_CHUNK_BYTES));
        if (!s.computed[i])
            HIP_TRY(hipEventCreateWithFlags(&s.computed[i], hipEventDisableTiming));
        if (!s.downloaded[i])
            HIP_TRY(hipEventCreateWithFlags(&s.downloaded[i], hipEventDisableTiming));
    }
    if (!s.pool)
        s.pool.reset(new CopyPool(copy_threads()));
    s.staged_ready = true;
    return NTTX_OK;
}

// Transforms `count` vectors at `data` (host) chunk by chunk, D = STAGES chunks in flight.  In
// step t the host threads fill the pinned input of chunk t and empty the pinned output of chunk
// t - D while the device uploads, transforms and downloads the chunks in between (upload +
// kernels on the slot's stream, downloads on a second stream).  Buffers t mod D are reused every
// D-th chunk; the one host wait per step (download of chunk t - D complete) frees all three.
int run_staged(const nttx_plan* plan, int slot_index, nttx_plan::HostSlot& s, uint64_t* data, size_t count, bool fwd,
               const uint64_t* other, bool lazy)
{
    const size_t vec_bytes = sizeof(uint64_t) << plan->logn;
    const size_t room = other ? STAGE_CHUNK_BYTES / 2 : STAGE_CHUNK_BYTES; // a product keeps both operands in a chunk
    // about eight chunks per call, between 2 MiB and the size of the pinned buffers
    const size_t chunk_bytes = std::min(room, std::max<size_t>(2u << 20, count * vec_bytes / 8));
    const size_t per = std::max<size_t>(1, chunk_bytes / vec_bytes);
    const size_t chunks = (count + per - 1) / per;
    auto first_of = [&](size_t c) { return c * per; };
    auto count_of = [&](size_t c) { return std::min(per, count - c * per); };
    auto body = [&]() -> int {
        constexpr size_t D = nttx_plan::HostSlot::STAGES;
        for (size_t t = 0; t < chunks + D; t++) {
            const int b = (int)(t % D);
            const bool in = t < chunks, out = t >= D && t - D < chunks;
            if (out)
                HIP_TRY(hipEventSynchronize(s.downloaded[b]));
            const size_t in_bytes = in ? count_of(t) * vec_bytes : 0;
            const size_t out_bytes = out ? count_of(t - D) * vec_bytes : 0;
            s.pool->run({{s.pin_in[b], in ? data + (first_of(t) << plan->logn) : nullptr, in_bytes},
                         {s.pin_in[b] + in_bytes, (in && other) ? other + (first_of(t) << plan->logn) : nullptr,
                          other ? in_bytes : 0},
                         {out ? data + (first_of(t - D) << plan->logn) : nullptr, s.pin_out[b], out_bytes}});
            if (!in)
                continue;
            uint64_t* const d = static_cast<uint64_t*>(s.dev[b]);
            HIP_TRY(hipMemcpyAsync(d, s.pin_in[b], other ? 2 * in_bytes : in_bytes, hipMemcpyHostToDevice, s.stream));
            int rc = other ? enqueue_product(plan, slot_index, d, d + (count_of(t) << plan->logn), count_of(t), s.stream)
                           : enqueue(plan, slot_index, d, count_of(t), fwd, s.stream, -1, nullptr, lazy);
            if (rc != NTTX_OK)
                return rc;
            HIP_TRY(hipEventRecord(s.computed[b], s.stream));
            HIP_TRY(hipStreamWaitEvent(s.down, s.computed[b], 0));
            HIP_TRY(hipMemcpyAsync(s.pin_out[b], d, in_bytes, hipMemcpyDeviceToHost, s.down));
            HIP_TRY(hipEventRecord(s.downloaded[b], s.down));
        }
        return NTTX_OK;
    };
    const int rc = body();
    // nothing of this call may still be in flight when the slot is handed to the next one
    const hipError_t e1 = hipStreamSynchronize(s.stream), e2 = hipStreamSynchronize(s.down);
    if (rc == NTTX_OK && (e1 != hipSuccess || e2 != hipSuccess))
        return fail(NTTX_EHIP, std::string("stream synchronisation: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return rc;
}

// One device's share of a host-slice call: `count` vectors at `data` (and at `other` for a
// negacyclic product data <- data * other; fwd is ignored then).
int run_host_on(const nttx_plan* plan, size_t i, uint64_t* data, size_t count, bool fwd, const uint64_t* other,
                bool lazy)
{
    const size_t vec_bytes = (count << plan->logn) * sizeof(uint64_t);
    const size_t bytes = other ? 2 * vec_bytes : vec_bytes;
    HIP_TRY(hipSetDevice(plan->devices[i]));
    nttx_plan::HostSlot& s = *plan->host_slots[i];
    std::lock_guard<std::mutex> guard(s.lock);
    if (!s.stream)
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    static const bool no_staging = std::getenv("NTTX_NO_HOST_STAGING") != nullptr;
    if (!no_staging && vec_bytes >= STAGE_MIN_BYTES &&
        (sizeof(uint64_t) << plan->logn) <= (other ? STAGE_CHUNK_BYTES / 2 : STAGE_CHUNK_BYTES)) {
        if (ensure_staging(s) == NTTX_OK)
            return run_staged(plan, (int)i, s, data, count, fwd, other, lazy);
        (void)hipGetLastError(); // no pinned memory to be had: the plain path below still works
    }
    uint64_t* d = nullptr;
    bool own = false; // d was allocated for this call (too large to keep)
    if (bytes <= s.cap) {
        d = static_cast<uint64_t*>(s.buf);
    } else if (bytes <= HOST_SLOT_KEEP_BYTES) {
        if (s.buf)
            (void)hipFree(s.buf);
        s.buf = nullptr;
        s.cap = 0;
        HIP_TRY(hipMalloc(&s.buf, bytes));
        s.cap = bytes;
        d = static_cast<uint64_t*>(s.buf);
    } else {
        HIP_TRY(hipMalloc(&d, bytes));
        own = true;
    }
    auto body = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(d, data, vec_bytes, hipMemcpyHostToDevice, s.stream));
        int r;
        if (other) {
            uint64_t* d2 = d + (count << plan->logn);
            HIP_TRY(hipMemcpyAsync(d2, other, vec_bytes, hipMemcpyHostToDevice, s.stream));
            r = enqueue_product(plan, (int)i, d, d2, count, s.stream);
        } else {
            r = enqueue(plan, (int)i, d, count, fwd, s.stream, -1, nullptr, lazy);
        }
        if (r != NTTX_OK)
            return r;
        HIP_TRY(hipMemcpyAsync(data, d, vec_bytes, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipStreamSynchronize(s.stream));
        return NTTX_OK;
    };
    const int rc = body();
    if (rc != NTTX_OK)
        (void)hipStreamSynchronize(s.stream);
    if (own)
        (void)hipFree(d);
    return rc;
}

// `other` != NULL: negacyclic product data <- data * other instead of a transform (fwd ignored).
// The batch is split evenly over the plan's devices; with several devices each share runs on a
// host thread of its own.
int run_host(const nttx_plan* plan, uint64_t* data, size_t batch, bool fwd, const uint64_t* other = nullptr,
             bool lazy = false)
{
    if (!plan)
        return fail(NTTX_EINVAL, "plan is NULL");
    if (batch == 0)
        return NTTX_OK;
    if (!data)
        return fail(NTTX_EINVAL, "data is NULL");
    const uint64_t total = (uint64_t)batch << plan->logn;
    if ((total >> plan->logn) != batch || total > (1ull << 40))
        return fail(NTTX_EINVAL, "batch * n is too large");
    const size_t ndev = plan->devices.size();
    const size_t per = (batch + ndev - 1) / ndev;
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = NTTX_OK;
    if (ndev == 1 || batch <= per) {
        rc = run_host_on(plan, 0, data, batch, fwd, other, lazy);
    } else {
        std::vector<int> rcs(ndev, NTTX_OK);
        std::vector<std::string> msgs(ndev);
        std::vector<std::thread> threads;
        bool spawned = true;
        try {
            for (size_t i = 0; i < ndev && i * per < batch; i++) {
                const size_t first = i * per, count = std::min(per, batch - first);
                if (i + 1 == ndev || (i + 1) * per >= batch) { // the last share runs on this thread
                    rcs[i] = run_host_on(plan, i, data + (first << plan->logn), count, fwd,
                                         other ? other + (first << plan->logn) : nullptr, lazy);
                    if (rcs[i] != NTTX_OK)
                        msgs[i] = g_err;
                    break;
                }
                threads.emplace_back([&, i, first, count] {
                    rcs[i] = guarded([&] {
                        return run_host_on(plan, i, data + (first << plan->logn), count, fwd,
                                           other ? other + (first << plan->logn) : nullptr, lazy);
                    });
                    if (rcs[i] != NTTX_OK)
                        msgs[i] = g_err; // thread-local: carried back to the caller's thread below
                });
            }
        } catch (...) {
            spawned = false;
        }
        for (auto& t : threads)
            t.join();
        if (!spawned)
            rc = fail(NTTX_ENOMEM, "could not start a host thread");
        for (size_t i = 0; i < ndev; i++)
            if (rcs[i] != NTTX_OK && rc == NTTX_OK)
                rc = fail(rcs[i], msgs[i]);
    }
    (void)hipSetDevice(prev);
    return rc;
}

int run_device(const nttx_plan* plan, int device, void* dptr, size_t batch, void* stream, bool fwd,
               int only_pass = -1, bool lazy = false)
{
    if (!plan)
        return fail(NTTX_EINVAL, "plan is NULL");
    const int slot = device_slot(plan, device);
    if (slot < 0)
        return fail(NTTX_ENODEV, "device is not one of the plan's devices");
    if (batch == 0)
        return NTTX_OK;
    if (!dptr || (reinterpret_cast<uintptr_t>(dptr) & 15))
        return fail(NTTX_EINVAL, "device pointer must be non-NULL and 16-byte aligned");
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != device)
        HIP_TRY(hipSetDevice(device));
    int rc = enqueue(plan, slot, static_cast<uint64_t*>(dptr), batch, fwd, static_cast<hipStream_t>(stream), only_pass,
                     nullptr, lazy);
    if (cur != device && cur >= 0)
        (void)hipSetDevice(cur);
    return rc;
}

// Fills the regrouped copy of a twiddle table on the device that holds the table itself.
template <class E>
__global__ void regroup_kernel(const E* __restrict__ src, E* __restrict__ dst, uint64_t n, uint32_t logn)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[nttx::permuted_index(i, logn)] = src[i];
}

// One level of the doubling recurrence of host/tables.hpp (fill_bit_reversed_powers):
// tab[half + j] = tab[j] * s for j < half, s = g^(n / 2 half).  The products are the canonical
// residues the reference's chain stores (ntt.rs:251-258: REDC products; ntt_goldilock.rs:205-211:
// mul_reduce), so the table equals the host-built one bit for bit.
__global__ void table_level_kernel(uint64_t* tab, uint64_t half, uint64_t s, int goldilocks, nttx::MontCtx c)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < half)
        tab[half + j] = goldilocks ? nttx::Goldilocks::mul_strict(tab[j], s) : nttx::Mont<true>::mul(tab[j], s, c);
}

// Montgomery-form table -> the lazy policies' (plain twiddle, quotient) pairs:
// w = REDC(raw) (utils.rs:165-178), v = floor(w * 2^64 / q) by restoring division (q < 2^61, so
// the running remainder doubled stays below 2^62).
__global__ void shoup_pairs_kernel(const uint64_t* __restrict__ raw, nttx::TwPair* __restrict__ out, uint64_t n,
                                   nttx::MontCtx c)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint64_t w = nttx::Mont<true>::mul(raw[i], 1, c);
    uint64_t r = w, v = 0;
    for (int bit = 0; bit < 64; bit++) {
        r <<= 1;
        const bool ge = r >= c.q;
        r -= ge ? c.q : 0;
        v = (v << 1) | (ge ? 1u : 0u);
    }
    out[i] = nttx::TwPair{w, v};
}

// Montgomery-form table -> plain residues, in place (GOLDILOCKS_CANONICAL)
__global__ void plain_residues_kernel(uint64_t* tab, uint64_t n, nttx::MontCtx c)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        tab[i] = nttx::Mont<true>::mul(tab[i], 1, c);
}

// Where a plan's tables come from: host arrays in the reference's format, or the seeds of the
// doubling recurrence (the first `seed` entries, built on the host, and one factor per further level).
struct TableSource {
    const uint64_t* h_fwd = nullptr;
    const uint64_t* h_inv = nullptr;
    uint64_t seed = 0;                       // generated: entries [0, seed) come from seed_fwd / seed_inv
    std::vector<uint64_t> seed_fwd, seed_inv;
    std::vector<uint64_t> step_fwd, step_inv; // step[k] = g^(n / 2^(k+1)), k = log2(seed) .. log2(n) - 1
};

int finish_plan(nttx_plan* plan, const TableSource& src, const int* devices, int n_devices)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(NTTX_ENODEV, "no HIP device available");
    std::vector<int> devs;
    if (devices == nullptr || n_devices <= 0)
        devs = {0};
    else
        devs.assign(devices, devices + n_devices);
    for (size_t i = 0; i < devs.size(); i++) {
        if (devs[i] < 0 || devs[i] >= count)
            return fail(NTTX_ENODEV, "device index out of range");
        for (size_t j = 0; j < i; j++)
            if (devs[j] == devs[i])
                return fail(NTTX_EINVAL, "device listed twice");
    }
    plan->devices = devs; // only validated indices reach the plan (its destructor visits them)
    int prev = 0;
    (void)hipGetDevice(&prev);
    const bool pairs = plan->policy == nttx_plan::SHOUP61 || plan->policy == nttx_plan::SHOUP61_NEAR;
    const bool plain = plan->policy == nttx_plan::GOLDILOCKS_CANONICAL;
    const bool goldilocks = plan->kind == NTTX_GOLDILOCKS;
    const size_t raw_bytes = plan->n * sizeof(uint64_t);
    const size_t bytes = plan->n * (pairs ? sizeof(nttx::TwPair) : sizeof(uint64_t));
    const size_t nd = plan->devices.size();
    const unsigned grid = (unsigned)((plan->n + 255) / 256);
    plan->host_slots.clear();
    for (size_t i = 0; i < nd; i++)
        plan->host_slots.emplace_back(new nttx_plan::HostSlot);
    plan->d_fwd.assign(nd, nullptr);
    plan->d_inv.assign(nd, nullptr);
    plan->d_fwd_perm.assign(nd, nullptr);
    plan->d_inv_perm.assign(nd, nullptr);
    for (size_t i = 0; i < nd; i++) {
        HIP_TRY(hipSetDevice(plan->devices[i]));
        HIP_TRY(hipMalloc(&plan->d_fwd_perm[i], bytes));
        HIP_TRY(hipMalloc(&plan->d_inv_perm[i], bytes));
        for (int dir = 0; dir < 2; dir++) {
            // the table in the reference's format: copied from the caller, or generated here
            uint64_t* raw = nullptr;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&raw), raw_bytes));
            void*& table = dir == 0 ? plan->d_fwd[i] : plan->d_inv[i];
            table = raw; // owned by the plan from here on (freed with it if anything below fails)
            (void)hipGetLastError();
            if (src.seed == 0) {
                HIP_TRY(hipMemcpy(raw, dir == 0 ? src.h_fwd : src.h_inv, raw_bytes, hipMemcpyHostToDevice));
            } else {
                const std::vector<uint64_t>& seeds = dir == 0 ? src.seed_fwd : src.seed_inv;
                const std::vector<uint64_t>& step = dir == 0 ? src.step_fwd : src.step_inv;
                HIP_TRY(hipMemcpy(raw, seeds.data(), src.seed * sizeof(uint64_t), hipMemcpyHostToDevice));
                size_t level = 0;
                for (uint64_t half = src.seed; half < plan->n; half <<= 1, level++)
                    hipLaunchKernelGGL(table_level_kernel, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, nullptr, raw,
                                       half, step[level], goldilocks ? 1 : 0, plan->mctx);
            }
            // the table in the kernels' format
            if (pairs) {
                nttx::TwPair* tw = nullptr;
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&tw), bytes));
                hipLaunchKernelGGL(shoup_pairs_kernel, dim3(grid), dim3(256), 0, nullptr, raw, tw, plan->n, plan->mctx);
                const hipError_t e = hipDeviceSynchronize();
                table = tw;
                (void)hipFree(raw);
                HIP_TRY(e);
            } else if (plain) {
                hipLaunchKernelGGL(plain_residues_kernel, dim3(grid), dim3(256), 0, nullptr, raw, plan->n, plan->mctx);
            }
            // regrouped copy for the pass that holds the last four stages
            void* perm = dir == 0 ? plan->d_fwd_perm[i] : plan->d_inv_perm[i];
            if (pairs)
                hipLaunchKernelGGL(regroup_kernel<nttx::TwPair>, dim3(grid), dim3(256), 0, nullptr,
                                   static_cast<const nttx::TwPair*>(table), static_cast<nttx::TwPair*>(perm), plan->n, plan->logn);
            else
                hipLaunchKernelGGL(regroup_kernel<uint64_t>, dim3(grid), dim3(256), 0, nullptr,
                                   static_cast<const uint64_t*>(table), static_cast<uint64_t*>(perm), plan->n, plan->logn);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipDeviceSynchronize());
    }
    // The split transform exchanges pieces between the plan's devices: let them reach each other's
    // memory directly over xGMI where the topology allows it (otherwise peer copies are staged by the
    // runtime).  NTTX_NO_PEER_ACCESS=1 leaves peer access off, which forces the staged form on any
    // topology (an alternative path for tests on multi-GPU hosts).
    static const bool no_peer = std::getenv("NTTX_NO_PEER_ACCESS") != nullptr;
    for (size_t i = 0; i < nd && !no_peer; i++) {
        for (size_t j = 0; j < nd; j++) {
            int can = 0;
            if (i == j || hipDeviceCanAccessPeer(&can, plan->devices[i], plan->devices[j]) != hipSuccess || !can)
                continue;
            HIP_TRY(hipSetDevice(plan->devices[i]));
            const hipError_t e = hipDeviceEnablePeerAccess(plan->devices[j], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                return fail(NTTX_EHIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            (void)hipGetLastError(); // "already enabled" is not an error worth keeping
        }
    }
    (void)hipSetDevice(prev);
    return NTTX_OK;
}

// Everything of plan creation but the origin of the tables.  psi1 / psi_inv1: entry 1 of the two
// tables (reference format).
int create_plan(int kind, uint64_t q, uint64_t k, uint64_t n, uint64_t n_inv, const TableSource& src, const int* devices,
                int n_devices, nttx_plan** out)
{
    const uint32_t logn = ntt::host::ilog2(n);
    nttx_plan* plan = new (std::nothrow) nttx_plan;
    if (!plan)
        return fail(NTTX_ENOMEM, "out of host memory");
    plan->kind = kind;
    plan->q = q;
    plan->k = k;
    plan->n = n;
    plan->n_inv = n_inv;
    plan->logn = logn;
    using u128 = unsigned __int128;
    if (kind == NTTX_GOLDILOCKS)
        plan->policy = nttx_plan::GOLDILOCKS;
    else if (q < (1ull << 61))
        // 8q <= 2^64: lazy butterflies; moduli just below 2^61 get the cheaper reductions (NTTX_NO_NEAR: not)
        plan->policy = nttx::shoup61_is_near(q) && std::getenv("NTTX_NO_NEAR") == nullptr ? nttx_plan::SHOUP61_NEAR
                                                                                           : nttx_plan::SHOUP61;
    else if (q == nttx::Goldilocks::Q)
        plan->policy = nttx_plan::GOLDILOCKS_CANONICAL; // Table::new_goldilock: the prime's own arithmetic
    else
        plan->policy = (q >> 63) ? nttx_plan::MONT_WIDE : nttx_plan::MONT;
    const uint64_t r_mod_q = (uint64_t)((((u128)1) << 64) % q);
    plan->mctx = nttx::MontCtx{q, k, r_mod_q, n_inv};
    plan->mctx_mul = plan->mctx;
    plan->mctx_mul.n_inv_m = (uint64_t)(((u128)n_inv * r_mod_q) % q); // n^-1 * 2^128: REDC leaves n^-1 * 2^64
    plan->gctx = nttx::GoldilocksCtx{n_inv, 0, 0};
    const uint64_t psi_inv1 = src.seed ? src.seed_inv[1] : src.h_inv[1];
    if (plan->policy == nttx_plan::SHOUP61 || plan->policy == nttx_plan::SHOUP61_NEAR) {
        // The kernels multiply plain data by plain twiddles with a precomputed quotient; the constants
        // leave Montgomery form here, the tables on the device (shoup_pairs_kernel).
        const ntt::utils::MontgomeryTransformer<uint64_t> mt{k, 0, q};
        const uint64_t n_inv_plain = mt.reduce((u128)n_inv);
        const uint64_t s1 = (uint64_t)(((u128)mt.reduce((u128)psi_inv1) * n_inv_plain) % q);
        plan->sctx = nttx::make_shoup61_ctx(q, n_inv_plain, s1, k);
        plan->sctx_mul = nttx::make_shoup61_ctx(q, (uint64_t)(((u128)n_inv_plain * r_mod_q) % q),
                                                (uint64_t)(((u128)s1 * r_mod_q) % q), k);
    }
    if (plan->policy == nttx_plan::GOLDILOCKS_CANONICAL) {
        const ntt::utils::MontgomeryTransformer<uint64_t> mt{k, 0, q};
        plan->gctx = nttx::GoldilocksCtx{mt.reduce((u128)n_inv), 0, 0};
    }
    if ((plan->policy == nttx_plan::GOLDILOCKS || plan->policy == nttx_plan::GOLDILOCKS_CANONICAL) && n >= 16 &&
        std::getenv("NTTX_NO_POW2_TWIDDLES") == nullptr) {
        // Twiddles of the first four stages as shifts: taken when entries 1..15 of both tables are exactly
        // the powers of two the kernels have compiled in (tables of every psi with psi^(n/32) = 2^39).
        const ntt::utils::MontgomeryTransformer<uint64_t> mt{k, 0, q};
        const uint64_t* f = src.seed ? src.seed_fwd.data() : src.h_fwd;
        const uint64_t* b = src.seed ? src.seed_inv.data() : src.h_inv;
        bool match = true;
        for (int x = 1; x < 16 && match; x++) {
            const uint64_t fx = kind == NTTX_MONT64 ? mt.reduce((u128)f[x]) : f[x];
            const uint64_t bx = kind == NTTX_MONT64 ? mt.reduce((u128)b[x]) : b[x];
            match = fx == ntt::utils::mod_exp_u64(2, (uint64_t)nttx::Goldilocks::pow2_exponent(x, false), q) &&
                    bx == ntt::utils::mod_exp_u64(2, (uint64_t)nttx::Goldilocks::pow2_exponent(x, true), q);
        }
        plan->pow2_twiddles = match;
        // with them n^-1 = 2^(192 - log2 n) can be a shift too (checked: n_inv is what the caller passed)
        if (match && logn < 32 && plan->gctx.n_inv == ntt::utils::mod_exp_u64(2, 192 - logn, q))
            plan->gctx.logn = logn;
    }
    int rc = guarded([&] {
        if (src.seed == 0) {
            plan->h_fwd.assign(src.h_fwd, src.h_fwd + n);
            plan->h_inv.assign(src.h_inv, src.h_inv + n);
        }
        plan->passes = make_schedule(logn);
        plan->split_passes = make_passes(logn);
        return finish_plan(plan, src, devices, n_devices);
    });
    if (rc != NTTX_OK) {
        std::string keep = g_err;
        nttx_plan_destroy(plan);
        g_err = keep;
        return rc;
    }
    *out = plan;
    return NTTX_OK;
}

// Arguments every plan constructor checks.
int check_plan_arguments(int kind, uint64_t q, uint64_t k, uint64_t n, uint64_t n_inv)
{
    if (kind != NTTX_MONT64 && kind != NTTX_GOLDILOCKS)
        return fail(NTTX_EINVAL, "unknown kind");
    if (n < 4 || n > (1ull << 32) || (n & (n - 1)) != 0)
        return fail(NTTX_EINVAL, "n must be a power of two in [4, 2^32]");
    if (kind == NTTX_MONT64) {
        if ((q & 1) == 0 || q < 3)
            return fail(NTTX_EINVAL, "modulus is not coprime to helper modulus R=2^64");
        if (k * q != 1)
            return fail(NTTX_EINVAL, "k * q != 1 mod 2^64: the reference derives a wrong k for this modulus "
                                     "(utils.rs:148-152)");
    } else if (q != nttx::Goldilocks::Q) {
        return fail(NTTX_EINVAL, "NTTX_GOLDILOCKS requires q = 0xffffffff00000001");
    }
    if (n_inv >= q)
        return fail(NTTX_EINVAL, "n_inv is not a residue");
    return NTTX_OK;
}

// Entry 0 of both tables is the unit in the table's form (ntt.rs:247-253: the chain starts from
// transform(1); ntt_goldilock.rs:200-206: from 1); entries 1 are psi^(n/2) and its inverse; n_inv * n
// is the unit.  A table of another modulus, form or root pair -- or a psi that is not a primitive
// 2n-th root -- fails these.  The first entries of a generated table are those of its seed table, so
// the same checks serve both origins.
int check_table_heads(int kind, uint64_t q, uint64_t k, uint64_t n, uint64_t n_inv, uint64_t f0, uint64_t f1, uint64_t b0,
                      uint64_t b1)
{
    using u128 = unsigned __int128;
    const uint64_t one = kind == NTTX_MONT64 ? (uint64_t)((((u128)1) << 64) % q) : 1;
    const auto mul = [&](uint64_t a, uint64_t b) -> uint64_t {
        if (kind == NTTX_MONT64)
            return ntt::utils::MontgomeryTransformer<uint64_t>{k, 0, q}.mul(a, b);
        return ntt::host::goldilocks_mul_reduce(a, b);
    };
    if (f0 != one || b0 != one)
        return fail(NTTX_EINVAL, "entry 0 of a twiddle table is not the unit (2^64 mod q for NTTX_MONT64, 1 for NTTX_GOLDILOCKS)");
    if (mul(f1, b1) != one)
        return fail(NTTX_EINVAL, "entries 1 of the twiddle tables are not inverse to each other modulo q");
    if (mul(f1, f1) != q - one)
        return fail(NTTX_EINVAL, "entry 1 of the forward table is not a square root of -1 modulo q (psi is not a primitive 2n-th root)");
    // n * n_inv = 1 in the table's form: n_inv * n (plain) = unit
    uint64_t acc = n_inv; // n_inv * n by doubling (n is a power of two), canonical throughout
    for (uint64_t m = n; m > 1; m >>= 1) {
        const u128 d = (u128)acc * 2;
        acc = (uint64_t)(d >= q ? d - q : d);
    }
    if (acc != one)
        return fail(NTTX_EINVAL, "n_inv is not the inverse of n modulo q (in Montgomery form for NTTX_MONT64)");
    return NTTX_OK;
}

// Plans from a root generate their tables on the devices from this size on (the host builds the
// first DEVICE_TABLES_SEED entries and the per-level factors).  NTTX_HOST_TABLES=1: never.
constexpr uint64_t DEVICE_TABLES_MIN_N = 1ull << 18;
constexpr uint64_t DEVICE_TABLES_SEED = 1ull << 12;

} // namespace

extern "C" {

int nttx_plan_create(int kind, uint64_t q, uint64_t k, uint64_t n, uint64_t n_inv, const uint64_t* psi_bo,
                     const uint64_t* psi_inv_bo, const int* devices, int n_devices, nttx_plan** out)
{
    if (!out)
        return fail(NTTX_EINVAL, "out is NULL");
    *out = nullptr;
    if (kind != NTTX_MONT64 && kind != NTTX_GOLDILOCKS)
        return fail(NTTX_EINVAL, "unknown kind");
    if (!psi_bo || !psi_inv_bo)
        return fail(NTTX_EINVAL, "twiddle tables are NULL");
    int rc = check_plan_arguments(kind, q, k, n, n_inv);
    if (rc != NTTX_OK)
        return rc;
    for (uint64_t i = 0; i < n; i++)
        if (psi_bo[i] >= q || psi_inv_bo[i] >= q)
            return fail(NTTX_EINVAL, "twiddle table entry is not a residue");
    rc = check_table_heads(kind, q, k, n, n_inv, psi_bo[0], psi_bo[1], psi_inv_bo[0], psi_inv_bo[1]);
    if (rc != NTTX_OK)
        return rc;
    TableSource src;
    src.h_fwd = psi_bo;
    src.h_inv = psi_inv_bo;
    return create_plan(kind, q, k, n, n_inv, src, devices, n_devices, out);
}

static int plan_from_root(int kind, uint64_t q, uint64_t psi, uint64_t n, const int* devices, int n_devices,
                          nttx_plan** out, bool exact_k)
{
    if (!out)
        return fail(NTTX_EINVAL, "out is NULL");
    *out = nullptr;
    if (n < 4 || n > (1ull << 32) || (n & (n - 1)) != 0)
        return fail(NTTX_EINVAL, "n must be a power of two in [4, 2^32]");
    // The reference's constructors panic for odd log2(n) (bit_reverse, utils.rs:14-17, is no reversal
    // then and with_precomputes indexes out of bounds, ntt.rs:253); only the corrected constructor
    // builds such plans, with the tables of the true bit reversal.
    if (!exact_k && ntt::host::ilog2(n) % 2 != 0)
        return fail(NTTX_EINVAL, "log2(n) must be even: the reference's bit_reverse (utils.rs:14-17) "
                                 "indexes out of bounds for odd log2(n)");
    static const bool host_tables = std::getenv("NTTX_HOST_TABLES") != nullptr;
    try {
        if (n >= DEVICE_TABLES_MIN_N && !host_tables && (kind == NTTX_MONT64 || kind == NTTX_GOLDILOCKS)) {
            // Large tables are generated on the devices: the host builds the table of the nested
            // size-DEVICE_TABLES_SEED transform (identical to the first entries of the big table:
            // out[j] depends on g^(n / 2^(k+1)) for 2^k <= j only) and the factor of every further level.
            if (kind == NTTX_MONT64 ? (q < 3 || psi == 0 || psi >= q) : (q != nttx::Goldilocks::Q || psi == 0 || psi >= q))
                return fail(NTTX_EINVAL, kind == NTTX_MONT64 ? "need 0 < psi < q" : "NTTX_GOLDILOCKS requires q = 0xffffffff00000001 and 0 < psi < q");
            const uint64_t seed = DEVICE_TABLES_SEED;
            const uint32_t logn = ntt::host::ilog2(n), logseed = ntt::host::ilog2(seed);
            TableSource src;
            src.seed = seed;
            uint64_t k = 0, n_inv = 0, g = 0, g_inv = 0;
            std::function<uint64_t(uint64_t, uint64_t)> mul;
            ntt::utils::MontgomeryTransformer<uint64_t> M{0, 0, q};
            if (kind == NTTX_MONT64) {
                // scalars exactly as build_montgomery_tables derives them (ntt.rs:90-120)
                auto small = ntt::host::build_montgomery_tables(q, ntt::utils::mod_exp_u64(psi, n / seed, q), seed, exact_k);
                M = small.montgomery;
                k = M.k;
                using ntt::utils::i128;
                const i128 n_inv_raw = std::get<1>(ntt::utils::egcd_i128((i128)n, (i128)q));
                n_inv = M.transform(n_inv_raw < 0 ? (uint64_t)(n_inv_raw + (i128)q) : (uint64_t)n_inv_raw);
                const i128 psi_inv_raw = std::get<1>(ntt::utils::egcd_i128((i128)psi, (i128)q));
                g = M.transform(psi);
                g_inv = M.transform(psi_inv_raw < 0 ? (uint64_t)(psi_inv_raw + (i128)q) : (uint64_t)psi_inv_raw);
                src.seed_fwd = std::move(small.powers_psi_bo);
                src.seed_inv = std::move(small.powers_psi_inv_bo);
                mul = [M](uint64_t a, uint64_t b) { return M.mul(a, b); };
            } else {
                auto small = ntt::host::build_goldilocks_tables(ntt::utils::mod_exp_u64(psi, n / seed, q), seed);
                if ((q - 1) % n != 0)
                    return fail(NTTX_EINVAL, "n must be a power of two dividing q - 1");
                n_inv = q - (q - 1) / n;
                g = psi;
                g_inv = ntt::utils::mod_exp_u64(psi, q - 2, q);
                src.seed_fwd = std::move(small.powers_psi_bo);
                src.seed_inv = std::move(small.powers_psi_inv_bo);
                mul = ntt::host::goldilocks_mul_reduce;
            }
            // step[k] = g^(n / 2^(k+1)) for k = logseed .. logn - 1 (fill_bit_reversed_powers)
            auto steps = [&](uint64_t base) {
                std::vector<uint64_t> all(logn);
                uint64_t sq = base;
                for (uint32_t kk = logn; kk-- > 0;) {
                    all[kk] = sq;
                    sq = mul(sq, sq);
                }
                return std::vector<uint64_t>(all.begin() + logseed, all.end());
            };
            src.step_fwd = steps(g);
            src.step_inv = steps(g_inv);
            int rc = check_plan_arguments(kind, q, k, n, n_inv);
            if (rc == NTTX_OK)
                rc = check_table_heads(kind, q, k, n, n_inv, src.seed_fwd[0], src.seed_fwd[1], src.seed_inv[0], src.seed_inv[1]);
            if (rc != NTTX_OK)
                return rc;
            rc = create_plan(kind, q, k, n, n_inv, src, devices, n_devices, out);
            if (rc == NTTX_OK) {
                (*out)->psi = psi;
                (*out)->exact_k = exact_k;
            }
            return rc;
        }
        if (kind == NTTX_MONT64) {
            if (q < 3 || psi == 0 || psi >= q)
                return fail(NTTX_EINVAL, "need 0 < psi < q");
            auto t = ntt::host::build_montgomery_tables(q, psi, n, exact_k);
            return nttx_plan_create(kind, q, t.montgomery.k, n, t.n_inv_montgomery, t.powers_psi_bo.data(),
                                    t.powers_psi_inv_bo.data(), devices, n_devices, out);
        } else if (kind == NTTX_GOLDILOCKS) {
            if (q != nttx::Goldilocks::Q)
                return fail(NTTX_EINVAL, "NTTX_GOLDILOCKS requires q = 0xffffffff00000001");
            if (psi == 0 || psi >= q)
                return fail(NTTX_EINVAL, "need 0 < psi < q");
            auto t = ntt::host::build_goldilocks_tables(psi, n, exact_k);
            return nttx_plan_create(kind, q, 0, n, t.n_inv, t.powers_psi_bo.data(), t.powers_psi_inv_bo.data(),
                                    devices, n_devices, out);
        }
    } catch (const ntt::Panic& p) {
        return fail(NTTX_EINVAL, p.what());
    } catch (const std::bad_alloc&) {
        return fail(NTTX_ENOMEM, "out of host memory");
    }
    return fail(NTTX_EINVAL, "unknown kind");
}

int nttx_plan_create_from_root(int kind, uint64_t q, uint64_t psi, uint64_t n, const int* devices, int n_devices,
                               nttx_plan** out)
{
    return plan_from_root(kind, q, psi, n, devices, n_devices, out, false);
}

int nttx_plan_create_from_root_exact_k(int kind, uint64_t q, uint64_t psi, uint64_t n, const int* devices,
                                       int n_devices, nttx_plan** out)
{
    return plan_from_root(kind, q, psi, n, devices, n_devices, out, true);
}

void nttx_plan_destroy(nttx_plan* plan)
{
    if (!plan)
        return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < plan->devices.size(); i++) {
        (void)hipSetDevice(plan->devices[i]);
        if (i < plan->d_fwd.size() && plan->d_fwd[i])
            (void)hipFree(plan->d_fwd[i]);
        if (i < plan->d_inv.size() && plan->d_inv[i])
            (void)hipFree(plan->d_inv[i]);
        if (i < plan->d_fwd_perm.size() && plan->d_fwd_perm[i])
            (void)hipFree(plan->d_fwd_perm[i]);
        if (i < plan->d_inv_perm.size() && plan->d_inv_perm[i])
            (void)hipFree(plan->d_inv_perm[i]);
        if (i < plan->host_slots.size() && plan->host_slots[i]) {
            if (plan->host_slots[i]->stream)
                (void)hipStreamDestroy(plan->host_slots[i]->stream);
            if (plan->host_slots[i]->buf)
                (void)hipFree(plan->host_slots[i]->buf);
            nttx_plan::HostSlot& hs = *plan->host_slots[i];
            if (hs.down)
                (void)hipStreamDestroy(hs.down);
            for (int b = 0; b < nttx_plan::HostSlot::STAGES; b++) {
                if (hs.pin_in[b])
                    (void)hipHostFree(hs.pin_in[b]);
                if (hs.pin_out[b])
                    (void)hipHostFree(hs.pin_out[b]);
                if (hs.dev[b])
                    (void)hipFree(hs.dev[b]);
                if (hs.computed[b])
                    (void)hipEventDestroy(hs.computed[b]);
                if (hs.downloaded[b])
                    (void)hipEventDestroy(hs.downloaded[b]);
            }
        }
    }
    (void)hipSetDevice(prev);
    delete plan;
}

int nttx_plan_info(const nttx_plan* plan, int* kind, uint64_t* q, uint64_t* k, uint64_t* n, uint64_t* n_inv,
                   int* n_devices)
{
    if (!plan)
        return fail(NTTX_EINVAL, "plan is NULL");
    if (kind) *kind = plan->kind;
    if (q) *q = plan->q;
    if (k) *k = plan->k;
    if (n) *n = plan->n;
    if (n_inv) *n_inv = plan->n_inv;
    if (n_devices) *n_devices = (int)plan->devices.size();
    return NTTX_OK;
}

int nttx_plan_tables(const nttx_plan* plan, uint64_t* psi_bo_out, uint64_t* psi_inv_bo_out)
{
    if (!plan)
        return fail(NTTX_EINVAL, "plan is NULL");
    if (plan->h_fwd.empty()) {
        // The tables were generated on the devices: read the first device's copy back and undo the
        // conversion to the kernels' format, so that the caller sees what the kernels multiply by.
        return guarded([&]() -> int {
            int prev = 0;
            (void)hipGetDevice(&prev);
            HIP_TRY(hipSetDevice(plan->devices[0]));
            const bool pairs = plan->policy == nttx_plan::SHOUP61 || plan->policy == nttx_plan::SHOUP61_NEAR;
            const bool plain = plan->policy == nttx_plan::GOLDILOCKS_CANONICAL;
            const ntt::utils::MontgomeryTransformer<uint64_t> mt =
                plan->kind == NTTX_MONT64 ? (plan->exact_k ? ntt::utils::MontgomeryTransformer<uint64_t>::from_modulus_exact(plan->q)
                                                           : ntt::utils::MontgomeryTransformer<uint64_t>::from_modulus(plan->q))
                                          : ntt::utils::MontgomeryTransformer<uint64_t>{0, 0, plan->q};
            int rc = NTTX_OK;
            for (int dir = 0; dir < 2 && rc == NTTX_OK; dir++) {
                uint64_t* out = dir == 0 ? psi_bo_out : psi_inv_bo_out;
                const void* dev = dir == 0 ? plan->d_fwd[0] : plan->d_inv[0];
                if (!out)
                    continue;
                auto copy = [&]() -> int {
                    if (pairs) {
                        std::vector<nttx::TwPair> tmp(plan->n);
                        HIP_TRY(hipMemcpy(tmp.data(), dev, plan->n * sizeof(nttx::TwPair), hipMemcpyDeviceToHost));
                        for (uint64_t i = 0; i < plan->n; i++)
                            out[i] = mt.transform(tmp[i].w); // plain twiddle -> the reference's Montgomery form
                    } else {
                        HIP_TRY(hipMemcpy(out, dev, plan->n * sizeof(uint64_t), hipMemcpyDeviceToHost));
                        if (plain)
                            for (uint64_t i = 0; i < plan->n; i++)
                                out[i] = mt.transform(out[i]);
                    }
                    return NTTX_OK;
                };
                rc = copy();
            }
            (void)hipSetDevice(prev);
            return rc;
        });
    }
    if (psi_bo_out)
        std::memcpy(psi_bo_out, plan->h_fwd.data(), plan->n * sizeof(uint64_t));
    if (psi_inv_bo_out)
        std::memcpy(psi_inv_bo_out, plan->h_inv.data(), plan->n * sizeof(uint64_t));
    return NTTX_OK;
}

int nttx_forward_host(const nttx_plan* plan, uint64_t* data, size_t batch)
{
    return guarded([&] { return run_host(plan, data, batch, true); });
}
int nttx_backward_host(const nttx_plan* plan, uint64_t* data, size_t batch)
{
    return guarded([&] { return run_host(plan, data, batch, false); });
}

int nttx_forward_device(const nttx_plan* plan, int device, void* dptr, size_t batch, void* hip_stream)
{
    return run_device(plan, device, dptr, batch, hip_stream, true);
}
int nttx_backward_device(const nttx_plan* plan, int device, void* dptr, size_t batch, void* hip_stream)
{
    return run_device(plan, device, dptr, batch, hip_stream, false);
}

int nttx_forward_host_lazy(const nttx_plan* plan, uint64_t* data, size_t batch)
{
    return guarded([&] { return run_host(plan, data, batch, true, nullptr, true); });
}
int nttx_backward_host_lazy(const nttx_plan* plan, uint64_t* data, size_t batch)
{
    return guarded([&] { return run_host(plan, data, batch, false, nullptr, true); });
}
int nttx_forward_device_lazy(const nttx_plan* plan, int device, void* dptr, size_t batch, void* hip_stream)
{
    return run_device(plan, device, dptr, batch, hip_stream, true, -1, true);
}
int nttx_backward_device_lazy(const nttx_plan* plan, int device, void* dptr, size_t batch, void* hip_stream)
{
    return run_device(plan, device, dptr, batch, hip_stream, false, -1, true);
}

const char* nttx_plan_policy(const nttx_plan* plan)
{
    if (!plan)
        return "";
    switch (plan->policy) {
    case nttx_plan::SHOUP61: return "shoup61";
    case nttx_plan::SHOUP61_NEAR: return "shoup61_near";
    case nttx_plan::MONT: return "mont";
    case nttx_plan::MONT_WIDE: return "mont_wide";
    case nttx_plan::GOLDILOCKS: return "goldilocks";
    case nttx_plan::GOLDILOCKS_CANONICAL: return "goldilocks_canonical";
    }
    return "";
}

int nttx_plan_feature(const nttx_plan* plan, const char* name)
{
    if (!plan || !name)
        return -1;
    if (std::strcmp(name, "pow2_twiddles") == 0)
        return plan->pow2_twiddles ? 1 : 0;
    if (std::strcmp(name, "shift_scaling") == 0)
        return (plan->policy == nttx_plan::SHOUP61 || plan->policy == nttx_plan::SHOUP61_NEAR) && plan->sctx.scale_shift != 0 ? 1 : 0;
    return -1;
}

int nttx_negacyclic_mul_host(const nttx_plan* plan, uint64_t* a, const uint64_t* b, size_t batch)
{
    if (batch != 0 && !b)
        return fail(NTTX_EINVAL, "second operand is NULL");
    if (plan && batch != 0 && a) {
        // as for the device entry point: the same pointer squares, partial overlap is refused (chunks
        // of b are read from the caller's memory after earlier chunks of a have been written back)
        const uint64_t total = (uint64_t)batch << plan->logn;
        if (a != b && a < b + total && b < a + total)
            return fail(NTTX_EINVAL, "operands overlap (pass the same pointer twice to square)");
    }
    return guarded([&] { return run_host(plan, a, batch, true, batch ? b : nullptr); });
}

int nttx_negacyclic_mul_device(const nttx_plan* plan, int device, void* a, void* b, size_t batch, void* hip_stream)
{
    if (!plan)
        return fail(NTTX_EINVAL, "plan is NULL");
    const int slot = device_slot(plan, device);
    if (slot < 0)
        return fail(NTTX_ENODEV, "device is not one of the plan's devices");
    if (batch == 0)
        return NTTX_OK;
    if (!a || !b || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15))
        return fail(NTTX_EINVAL, "device pointers must be non-NULL and 16-byte aligned");
    const uint64_t total = (uint64_t)batch << plan->logn;
    if ((total >> plan->logn) != batch)
        return fail(NTTX_EINVAL, "batch * n is too large");
    uint64_t* pa = static_cast<uint64_t*>(a);
    uint64_t* pb = static_cast<uint64_t*>(b);
    if (pa != pb && pa < pb + total && pb < pa + total)
        return fail(NTTX_EINVAL, "operands overlap (pass the same pointer twice to square)");
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != device)
        HIP_TRY(hipSetDevice(device));
    int rc = enqueue_product(plan, slot, pa, pb, batch, static_cast<hipStream_t>(hip_stream));
    if (cur != device && cur >= 0)
        (void)hipSetDevice(cur);
    return rc;
}

} // extern "C"

// ---- host-slice transform that allocates nothing ------------------------------------------------
//
// Staging memory of nttx_transform_host_noalloc: part of the code object, so every GPU the library
// runs on holds its own copy from the moment the module is loaded until the process ends.  In host
// code the array's name is only a handle for hipGetSymbolAddress; the address kernels and copies
// use is looked up once per device, with that device current, and kept in scratch_ptr.
namespace {
constexpr size_t NOALLOC_MAX_BYTES = 32ull << 20;
constexpr int NOALLOC_MAX_DEVICES = 64;
static __device__ __attribute__((used)) __align__(256) uint64_t scratch[NOALLOC_MAX_BYTES / sizeof(uint64_t)];
std::mutex scratch_lock;                            // guards scratch_ptr
uint64_t* scratch_ptr[NOALLOC_MAX_DEVICES] = {};
std::mutex scratch_in_use[NOALLOC_MAX_DEVICES];     // held for a whole call: one user of a device's array at a time

// Device address of this device's copy of `scratch`; `device` must be current.
int scratch_on_current_device(int device, uint64_t** out)
{
    if (device < 0 || device >= NOALLOC_MAX_DEVICES)
        return fail(NTTX_ENODEV, "device index out of range for the static staging buffer");
    std::lock_guard<std::mutex> guard(scratch_lock);
    if (!scratch_ptr[device]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p)
            return fail(NTTX_EHIP, "hipGetSymbolAddress returned no address for the static staging buffer");
        scratch_ptr[device] = static_cast<uint64_t*>(p);
    }
    *out = scratch_ptr[device];
    return NTTX_OK;
}
} // namespace

extern "C" {

int nttx_transform_host_noalloc(const nttx_plan* plan, uint64_t* data, size_t batch, int dir)
{
    if (!plan)
        return fail(NTTX_EINVAL, "plan is NULL");
    if (dir != 0 && dir != 1)
        return fail(NTTX_EINVAL, "direction must be 0 (forward) or 1 (backward)");
    if (batch == 0)
        return NTTX_OK;
    if (!data)
        return fail(NTTX_EINVAL, "data is NULL");
    const size_t max_batch = NOALLOC_MAX_BYTES >> (plan->logn + 3);
    if (batch > max_batch)
        return fail(NTTX_EINVAL, "batch * n * 8 bytes exceeds the static staging buffer of 32 MiB "
                                 "(use nttx_forward_host / nttx_backward_host, which allocate)");
    const size_t bytes = (batch << plan->logn) * sizeof(uint64_t);
    const int device = plan->devices[0];
    if (device < 0 || device >= NOALLOC_MAX_DEVICES)
        return fail(NTTX_ENODEV, "device index out of range for the static staging buffer");
    // Every plan and thread of this process shares the device's one array: calls take turns (a static
    // mutex per device: locking allocates nothing).
    std::lock_guard<std::mutex> turn(scratch_in_use[device]);
    int prev = -1;
    HIP_TRY(hipGetDevice(&prev));
    if (prev != device)
        HIP_TRY(hipSetDevice(device));
    auto body = [&]() -> int {
        uint64_t* ptr = nullptr;
        int rc = scratch_on_current_device(device, &ptr);
        if (rc != NTTX_OK)
            return rc;
        // everything on the default stream: no stream, event or buffer is created
        HIP_TRY(hipMemcpyAsync(ptr, data, bytes, hipMemcpyHostToDevice, nullptr));
        rc = enqueue(plan, 0, ptr, batch, dir == 0, nullptr);
        if (rc != NTTX_OK) {
            (void)hipStreamSynchronize(nullptr);
            return rc;
        }
        HIP_TRY(hipMemcpyAsync(data, ptr, bytes, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        return NTTX_OK;
    };
    const int rc = body();
    if (prev != device && prev >= 0)
        (void)hipSetDevice(prev);
    return rc;
}

int nttx_pass_count(const nttx_plan* plan) { return plan ? (int)plan->passes.size() : 0; }

int nttx_pass_describe(const nttx_plan* plan, int dir, int pass_index, char* buf, size_t buf_len)
{
    if (!plan || !buf || buf_len == 0)
        return fail(NTTX_EINVAL, "bad argument");
    const int np = (int)plan->passes.size();
    if (pass_index < 0 || pass_index >= np || (dir != 0 && dir != 1))
        return fail(NTTX_EINVAL, "pass index or direction out of range");
    const PassDesc& d = plan->passes[dir == 0 ? pass_index : np - 1 - pass_index];
    if (d.fused)
        std::snprintf(buf, buf_len, "%s fused %d+8 stages %u-%u", dir == 0 ? "fwd" : "bwd", d.fused, d.s0 - d.fused + 1,
                      d.s0 + 8);
    else
        std::snprintf(buf, buf_len, "%s %s %d+%d stages %u-%u", dir == 0 ? "fwd" : "bwd", d.contig ? "contig" : "strided",
                      d.k1, d.k2, d.s0 + 1, d.s0 + (unsigned)(d.k1 + d.k2));
    return NTTX_OK;
}

int nttx_run_pass_device(const nttx_plan* plan, int device, void* dptr, size_t batch, int dir, int pass_index,
                         void* hip_stream)
{
    if (!plan || pass_index < 0 || pass_index >= (int)plan->passes.size() || (dir != 0 && dir != 1))
        return fail(NTTX_EINVAL, "pass index or direction out of range");
    return run_device(plan, device, dptr, batch, hip_stream, dir == 0, pass_index);
}

} // extern "C"

namespace {
// One wave: elapsed shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) over a spin of
// `ticks` ticks.  The loop ends on the tick count or after a fixed number of polls, whichever
// comes first, so the wave finishes whatever the counters do.
__global__ void clock_probe_kernel(uint64_t* out2, uint64_t ticks)
{
    if (threadIdx.x != 0)
        return;
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    uint64_t r = r0;
    for (uint32_t polls = 0; polls < (1u << 26) && r - r0 < ticks; polls++) {
        __builtin_amdgcn_s_sleep(8);
        r = __builtin_amdgcn_s_memrealtime();
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    out2[0] = c1 - c0;
    out2[1] = r - r0;
}
} // namespace

extern "C" {

int nttx_clock_probe_device(int device, uint64_t* out2, int spin_us, void* hip_stream)
{
    if (!out2)
        return fail(NTTX_EINVAL, "out2 is NULL");
    if (spin_us < 1 || spin_us > 100000)
        return fail(NTTX_EINVAL, "spin_us must be in [1, 100000]");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return fail(NTTX_ENODEV, "device index out of range");
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != device)
        HIP_TRY(hipSetDevice(device));
    (void)hipGetLastError();
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(hip_stream), out2,
                       (uint64_t)spin_us * 100);
    const hipError_t e = hipGetLastError();
    if (cur != device && cur >= 0)
        (void)hipSetDevice(cur);
    if (e != hipSuccess)
        return fail(NTTX_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return NTTX_OK;
}

int nttx_forward_distributed(const nttx_plan* plan, uint64_t* host_data, int n_shards)
{
    return guarded([&] { return run_distributed(plan, host_data, true, n_shards); });
}
int nttx_backward_distributed(const nttx_plan* plan, uint64_t* host_data, int n_shards)
{
    return guarded([&] { return run_distributed(plan, host_data, false, n_shards); });
}

int nttx_shard_geometry(const nttx_plan* plan, int n_shards, uint64_t* local_elems, uint64_t* rows, uint64_t* cols,
                        uint64_t* rows_per_peer)
{
    ShardGeom g;
    int rc = shard_geometry(plan, n_shards, g);
    if (rc != NTTX_OK)
        return rc;
    if (local_elems) *local_elems = g.local;
    if (rows) *rows = g.rows;
    if (cols) *cols = g.W;
    if (rows_per_peer) *rows_per_peer = g.R;
    return NTTX_OK;
}

int nttx_shard_phase_device(const nttx_plan* plan, int device, void* dptr, int dir, int phase, int shard, int n_shards,
                            void* hip_stream)
{
    ShardGeom g;
    int rc = shard_geometry(plan, n_shards, g);
    if (rc != NTTX_OK)
        return rc;
    if ((dir != 0 && dir != 1) || (phase != 0 && phase != 1) || shard < 0 || shard >= n_shards)
        return fail(NTTX_EINVAL, "direction, phase or shard out of range");
    const int slot = device_slot(plan, device);
    if (slot < 0)
        return fail(NTTX_ENODEV, "device is not one of the plan's devices");
    if (!dptr || (reinterpret_cast<uintptr_t>(dptr) & 15))
        return fail(NTTX_EINVAL, "device pointer must be non-NULL and 16-byte aligned");
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != device)
        HIP_TRY(hipSetDevice(device));
    rc = enqueue_shard_phase(plan, slot, static_cast<uint64_t*>(dptr), dir == 0, phase, shard, g,
                             static_cast<hipStream_t>(hip_stream));
    if (cur != device && cur >= 0)
        (void)hipSetDevice(cur);
    return rc;
}

int nttx_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess)
        return 0;
    return count;
}

const char* nttx_last_error(void) { return g_err.c_str(); }
const char* nttx_version(void)
{
#ifdef NTTX_PROFILING
    return "nttx 0.3 (gfx950) PROFILING BUILD: honours NTTX_DEBUG_* switches that produce wrong results";
#else
    return "nttx 0.3 (gfx950)";
#endif
}

} // extern "C"
