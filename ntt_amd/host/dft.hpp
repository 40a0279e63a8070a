// C++ host mirror of the reference's `ntt::dft` module: the `DFT<O>` trait (src/dft.rs:6-11)
// and the three table types, with the transform bodies replaced by calls into libnttx.so
// (include/nttx.h).  Table construction stays on the host exactly as in the reference
// (tables.hpp); slices are (pointer, length) pairs or std::vector.  A reference panic is a
// thrown ntt::Panic.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/nttx.h"
#include "tables.hpp"
#include "utils.hpp"

namespace ntt {
namespace dft {

// src/dft.rs:6-11
template <class O>
struct DFT {
    virtual ~DFT() = default;
    virtual void forward_inplace(O* x, size_t len) const = 0;
    virtual void forward_inplace_lazy(O* x, size_t len) const = 0;
    virtual void backward_inplace(O* x, size_t len) const = 0;
    virtual void backward_inplace_lazy(O* x, size_t len) const = 0;

    void forward_inplace(std::vector<O>& x) const { forward_inplace(x.data(), x.size()); }
    void forward_inplace_lazy(std::vector<O>& x) const { forward_inplace_lazy(x.data(), x.size()); }
    void backward_inplace(std::vector<O>& x) const { backward_inplace(x.data(), x.size()); }
    void backward_inplace_lazy(std::vector<O>& x) const { backward_inplace_lazy(x.data(), x.size()); }
};

namespace detail {

inline void check(int rc)
{
    if (rc != NTTX_OK)
        throw Panic(std::string("nttx: ") + nttx_last_error());
}

// Owns one nttx_plan; shared by the three table types.
class PlanHolder {
  public:
    PlanHolder() = default;
    PlanHolder(const PlanHolder&) = delete;
    PlanHolder& operator=(const PlanHolder&) = delete;
    PlanHolder(PlanHolder&& o) noexcept : plan_(o.plan_) { o.plan_ = nullptr; }
    PlanHolder& operator=(PlanHolder&& o) noexcept
    {
        if (this != &o) {
            reset();
            plan_ = o.plan_;
            o.plan_ = nullptr;
        }
        return *this;
    }
    ~PlanHolder() { reset(); }
    void reset()
    {
        if (plan_)
            nttx_plan_destroy(plan_);
        plan_ = nullptr;
    }
    nttx_plan** out() { return &plan_; }
    const nttx_plan* get() const { return plan_; }

  private:
    nttx_plan* plan_ = nullptr;
};

// Length semantics: the reference sizes the transform by the slice (ntt.rs:123), but only
// len == n round-trips (ntt.rs:231-235) and len > n indexes out of bounds (ntt.rs:143).
// range_2q: outputs in [0, 2q) (nttx_*_lazy) instead of canonical residues -- an explicit opt-in; the
// trait's lazy methods return the strict outputs, as the reference's do
inline void run(const nttx_plan* plan, uint64_t n, void* x, size_t len, bool forward, bool range_2q = false)
{
    if (len != n)
        throw Panic("slice length does not match the table size");
    uint64_t* a = static_cast<uint64_t*>(x);
    if (range_2q)
        check(forward ? nttx_forward_host_lazy(plan, a, 1) : nttx_backward_host_lazy(plan, a, 1));
    else
        check(forward ? nttx_forward_host(plan, a, 1) : nttx_backward_host(plan, a, 1));
}

// Same transform computed by n_shards shards over the plan's devices (one exchange between them);
// n_shards = 0: one shard per device.  No counterpart in the reference, which runs any length in
// one loop nest; results are identical.
inline void run_distributed(const nttx_plan* plan, uint64_t n, void* x, size_t len, bool forward, int n_shards)
{
    if (len != n)
        throw Panic("slice length does not match the table size");
    check(forward ? nttx_forward_distributed(plan, static_cast<uint64_t*>(x), n_shards)
                  : nttx_backward_distributed(plan, static_cast<uint64_t*>(x), n_shards));
}

// a <- a * b in Z_q[X]/(X^n + 1): forward(a), forward(b), coefficient-wise product, backward(a),
// fused on the device.  No counterpart in the reference (it is what its callers compose).
inline void run_product(const nttx_plan* plan, uint64_t n, void* a, size_t len_a, const void* b, size_t len_b)
{
    if (len_a != n || len_b != n)
        throw Panic("slice length does not match the table size");
    check(nttx_negacyclic_mul_host(plan, static_cast<uint64_t*>(a), static_cast<const uint64_t*>(b), 1));
}

} // namespace detail

template <class O>
class Table;

// src/dft/ntt.rs:6-21
template <>
class Table<uint64_t> : public DFT<uint64_t> {
  public:
    uint64_t q;                              // ntt.rs:8
    uint64_t psi;                            // ntt.rs:10
    std::vector<uint64_t> powers_psi_bo;     // ntt.rs:17
    std::vector<uint64_t> powers_psi_inv_bo; // ntt.rs:20

    using DFT<uint64_t>::forward_inplace;
    using DFT<uint64_t>::forward_inplace_lazy;
    using DFT<uint64_t>::backward_inplace;
    using DFT<uint64_t>::backward_inplace_lazy;

    // ntt.rs:90-120.  exact_k (not in the reference): k = q^-1 mod 2^64 for every odd modulus, where
    // MontgomeryTransformer::from_modulus (utils.rs:148-152) derives a wrong k for about half of them
    // (12289, 65537 ...) and the library rejects the plan.
    static Table from_prime_and_root(uint64_t modulus, uint64_t root, uint64_t n, const std::vector<int>& devices = {},
                                     bool exact_k = false)
    {
        host::MontgomeryTables t = host::build_montgomery_tables(modulus, root, n, exact_k);
        Table out;
        out.q = modulus;
        out.psi = root;
        out.montgomery_ = t.montgomery;
        out.n_inv_montgomery_ = t.n_inv_montgomery;
        out.n_ = n;
        out.psi_inv_montgomery_ = t.psi_inv_montgomery;
        out.powers_psi_bo = std::move(t.powers_psi_bo);
        out.powers_psi_inv_bo = std::move(t.powers_psi_inv_bo);
        detail::check(nttx_plan_create(NTTX_MONT64, modulus, out.montgomery_.k, n, out.n_inv_montgomery_,
                                       out.powers_psi_bo.data(), out.powers_psi_inv_bo.data(),
                                       devices.empty() ? nullptr : devices.data(), (int)devices.size(),
                                       out.plan_.out()));
        return out;
    }
    static Table new_() { return from_prime_and_root(0x1fffffffffe00001ull, 0x15eb043c7aa2b01full, 1ull << 16); } // ntt.rs:55-57
    static Table default_() { return new_(); }                                                                      // ntt.rs:264-268
    static Table new_goldilock() { return from_prime_and_root(0xffffffff00000001ull, 0xabd0a6e8aa3d8a0eull, 1ull << 16); } // :60-62
    static Table new_u32_compatible() { return from_prime_and_root(4293918721ull, 2004365341ull, 1ull << 16); }     // :66-68
    static Table new_simple() { return from_prime_and_root(7681, 1925, 1ull << 3); }                                // :72-75 (panics)

    void forward_inplace(uint64_t* a, size_t len) const override { forward_inplace_core<false>(a, len); }      // ntt.rs:27-29
    void forward_inplace_lazy(uint64_t* a, size_t len) const override { forward_inplace_core<true>(a, len); }  // ntt.rs:34-36
    void backward_inplace(uint64_t* a, size_t len) const override { backward_inplace_core<false>(a, len); }    // ntt.rs:41-43
    void backward_inplace_lazy(uint64_t* a, size_t len) const override { backward_inplace_core<true>(a, len); } // ntt.rs:48-50

    // ntt.rs:122, 178.  The reference accepts LAZY and never reads it, so its lazy methods return the
    // strict outputs; the same here, bit for bit.
    template <bool LAZY>
    void forward_inplace_core(uint64_t* a, size_t len) const { detail::run(plan_.get(), n_, a, len, true); }
    template <bool LAZY>
    void backward_inplace_core(uint64_t* a, size_t len) const { detail::run(plan_.get(), n_, a, len, false); }
    // Explicit opt-in with no counterpart in the reference: outputs in [0, 2q), the range its lazy
    // methods document (ntt.rs:31-35, 45-49), congruent to the canonical ones (nttx_*_lazy).
    void forward_inplace_range_2q(uint64_t* a, size_t len) const { detail::run(plan_.get(), n_, a, len, true, true); }
    void backward_inplace_range_2q(uint64_t* a, size_t len) const { detail::run(plan_.get(), n_, a, len, false, true); }

    void forward_inplace_distributed(uint64_t* a, size_t len, int n_shards = 0) const
    {
        detail::run_distributed(plan_.get(), n_, a, len, true, n_shards);
    }
    void backward_inplace_distributed(uint64_t* a, size_t len, int n_shards = 0) const
    {
        detail::run_distributed(plan_.get(), n_, a, len, false, n_shards);
    }

    void negacyclic_mul_inplace(std::vector<uint64_t>& a, const std::vector<uint64_t>& b) const
    {
        detail::run_product(plan_.get(), n_, a.data(), a.size(), b.data(), b.size());
    }

    const nttx_plan* plan() const { return plan_.get(); }

  private:
    Table() : q(0), psi(0) {}
    utils::MontgomeryTransformer<uint64_t> montgomery_{};
    uint64_t n_inv_montgomery_ = 0;
    uint64_t n_ = 0;
    uint64_t psi_inv_montgomery_ = 0;
    detail::PlanHolder plan_;
};

// src/dft/ntt_avx2.rs:7-24.  Same transform on i64 slices; the GPU kernels are the ones
// Table<u64> uses (the reference's own test requires equal outputs, dft.rs:58-77).
class TableAVX2 : public DFT<int64_t> {
  public:
    int64_t q;   // ntt_avx2.rs:9
    int64_t psi; // ntt_avx2.rs:11

    using DFT<int64_t>::forward_inplace;
    using DFT<int64_t>::forward_inplace_lazy;
    using DFT<int64_t>::backward_inplace;
    using DFT<int64_t>::backward_inplace_lazy;

    // ntt_avx2.rs:74-105
    static TableAVX2 from_prime_and_root(int64_t modulus, int64_t root, int64_t n, const std::vector<int>& devices = {})
    {
        // signed compares at ntt_avx2.rs:563-589 need q < 2^62; len <= 8 takes a wrong scalar path (:264-268)
        if (modulus <= 0 || modulus >= ((int64_t)1 << 62))
            throw Panic("TableAVX2 supports positive moduli below 2^62");
        if (n < 16)
            throw Panic("TableAVX2 needs n >= 16");
        host::MontgomeryTables t = host::build_montgomery_tables((uint64_t)modulus, (uint64_t)root, (uint64_t)n);
        TableAVX2 out;
        out.q = modulus;
        out.psi = root;
        out.n_ = (size_t)n;
        out.k_ = (int64_t)t.montgomery.k;
        out.n_inv_montgomery_ = (int64_t)t.n_inv_montgomery;
        out.psi_inv_montgomery_ = (int64_t)t.psi_inv_montgomery;
        out.powers_psi_bo_.assign(t.powers_psi_bo.begin(), t.powers_psi_bo.end());
        out.powers_psi_inv_bo_.assign(t.powers_psi_inv_bo.begin(), t.powers_psi_inv_bo.end());
        detail::check(nttx_plan_create(NTTX_MONT64, (uint64_t)modulus, t.montgomery.k, (uint64_t)n,
                                       t.n_inv_montgomery, t.powers_psi_bo.data(), t.powers_psi_inv_bo.data(),
                                       devices.empty() ? nullptr : devices.data(), (int)devices.size(),
                                       out.plan_.out()));
        return out;
    }
    static TableAVX2 new_() { return from_prime_and_root(0x1fffffffffe00001ll, 0x15eb043c7aa2b01fll, 1ll << 16); } // :57-59

    void forward_inplace(int64_t* a, size_t len) const override { forward_inplace_core<false>(a, len); }
    void forward_inplace_lazy(int64_t* a, size_t len) const override { forward_inplace_core<true>(a, len); }
    void backward_inplace(int64_t* a, size_t len) const override { backward_inplace_core<false>(a, len); }
    void backward_inplace_lazy(int64_t* a, size_t len) const override { backward_inplace_core<true>(a, len); }

    template <bool LAZY> // never read by the reference (ntt_avx2.rs:108, 281): lazy == strict
    void forward_inplace_core(int64_t* a, size_t len) const { detail::run(plan_.get(), n_, a, len, true); }   // :108
    template <bool LAZY>
    void backward_inplace_core(int64_t* a, size_t len) const { detail::run(plan_.get(), n_, a, len, false); } // :281
    // opt-in, no counterpart in the reference: outputs in [0, 2q) (nttx_*_lazy)
    void forward_inplace_range_2q(int64_t* a, size_t len) const { detail::run(plan_.get(), n_, a, len, true, true); }
    void backward_inplace_range_2q(int64_t* a, size_t len) const { detail::run(plan_.get(), n_, a, len, false, true); }

    void forward_inplace_distributed(int64_t* a, size_t len, int n_shards = 0) const
    {
        detail::run_distributed(plan_.get(), n_, a, len, true, n_shards);
    }
    void backward_inplace_distributed(int64_t* a, size_t len, int n_shards = 0) const
    {
        detail::run_distributed(plan_.get(), n_, a, len, false, n_shards);
    }

    void negacyclic_mul_inplace(std::vector<int64_t>& a, const std::vector<int64_t>& b) const
    {
        detail::run_product(plan_.get(), n_, a.data(), a.size(), b.data(), b.size());
    }

    const nttx_plan* plan() const { return plan_.get(); }

  private:
    TableAVX2() : q(0), psi(0) {}
    int64_t psi_inv_montgomery_ = 0;
    size_t n_ = 0;
    int64_t n_inv_montgomery_ = 0;
    int64_t k_ = 0;
    std::vector<int64_t> powers_psi_bo_, powers_psi_inv_bo_;
    detail::PlanHolder plan_;
};

template <class O>
class TableGoldilock;

// src/dft/ntt_goldilock.rs:6-15
template <>
class TableGoldilock<uint64_t> : public DFT<uint64_t> {
  public:
    uint64_t q;   // ntt_goldilock.rs:8
    uint64_t psi; // ntt_goldilock.rs:10

    using DFT<uint64_t>::forward_inplace;
    using DFT<uint64_t>::forward_inplace_lazy;
    using DFT<uint64_t>::backward_inplace;
    using DFT<uint64_t>::backward_inplace_lazy;

    static TableGoldilock new_() { return from_root(0xabd0a6e8aa3d8a0eull, 1ull << 16); } // ntt_goldilock.rs:48-58

    // The same table for another (psi, n); the reference hard-codes both.  corrected: odd log2(n) with
    // the true bit reversal (with_precomputes, ntt_goldilock.rs:192-214, panics there).
    static TableGoldilock from_root(uint64_t psi, uint64_t n, const std::vector<int>& devices = {},
                                    bool corrected = false)
    {
        host::GoldilocksTables t = host::build_goldilocks_tables(psi, n, corrected);
        TableGoldilock out;
        out.q = host::GOLDILOCKS_Q;
        out.psi = psi;
        out.n_ = n;
        out.n_inv_ = t.n_inv;
        out.powers_psi_bo_ = std::move(t.powers_psi_bo);
        out.powers_psi_inv_bo_ = std::move(t.powers_psi_inv_bo);
        detail::check(nttx_plan_create(NTTX_GOLDILOCKS, out.q, 0, n, out.n_inv_, out.powers_psi_bo_.data(),
                                       out.powers_psi_inv_bo_.data(), devices.empty() ? nullptr : devices.data(),
                                       (int)devices.size(), out.plan_.out()));
        return out;
    }

    // ntt_goldilock.rs:63-79
    uint64_t reduce_159(uint64_t low, uint64_t middle, uint64_t high) const
    {
        return host::goldilocks_reduce_159(low, middle, high);
    }

    void forward_inplace(uint64_t* a, size_t len) const override { forward_inplace_core<false>(a, len); }
    void forward_inplace_lazy(uint64_t* a, size_t len) const override { forward_inplace_core<true>(a, len); }
    void backward_inplace(uint64_t* a, size_t len) const override { backward_inplace_core<false>(a, len); }
    void backward_inplace_lazy(uint64_t* a, size_t len) const override { backward_inplace_core<true>(a, len); }

    template <bool LAZY> // never read by the reference (ntt_goldilock.rs:81, 128): lazy == strict
    void forward_inplace_core(uint64_t* a, size_t len) const { detail::run(plan_.get(), n_, a, len, true); }   // :81
    template <bool LAZY>
    void backward_inplace_core(uint64_t* a, size_t len) const { detail::run(plan_.get(), n_, a, len, false); } // :128

    void forward_inplace_distributed(uint64_t* a, size_t len, int n_shards = 0) const
    {
        detail::run_distributed(plan_.get(), n_, a, len, true, n_shards);
    }
    void backward_inplace_distributed(uint64_t* a, size_t len, int n_shards = 0) const
    {
        detail::run_distributed(plan_.get(), n_, a, len, false, n_shards);
    }

    void negacyclic_mul_inplace(std::vector<uint64_t>& a, const std::vector<uint64_t>& b) const
    {
        detail::run_product(plan_.get(), n_, a.data(), a.size(), b.data(), b.size());
    }

    const nttx_plan* plan() const { return plan_.get(); }

  private:
    TableGoldilock() : q(0), psi(0) {}
    uint64_t n_ = 0, n_inv_ = 0;
    std::vector<uint64_t> powers_psi_bo_, powers_psi_inv_bo_;
    detail::PlanHolder plan_;
};

// ntt_goldilock.rs:218-227: despite its doc comment the pair is (high, low)
inline std::pair<uint64_t, uint64_t> u128_to_u64(utils::u128 a) { return {(uint64_t)(a >> 64), (uint64_t)a}; }
// ntt_goldilock.rs:231-238
inline utils::u128 u64_to_u128(uint64_t first, uint64_t second) { return ((utils::u128)first << 64) | second; }

} // namespace dft
} // namespace ntt
