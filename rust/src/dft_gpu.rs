//! GPU bodies for the three table types of the `ntt` crate: what replaces the CPU loop nests of
//! `Table<u64>` (src/dft/ntt.rs:122-236), `TableAVX2` (src/dft/ntt_avx2.rs:108-470) and
//! `TableGoldilock<u64>` (src/dft/ntt_goldilock.rs:81-180) when the crate is built against libnttx.
//!
//! How a maintainer applies it:
//!   1. each struct gains one private field, `plan: Option<crate::ffi::Plan>`, initialised to `None`
//!      where the struct literal is written (ntt.rs:106-116, ntt_avx2.rs:91-101,
//!      ntt_goldilock.rs:49-57);
//!   2. `with_precomputes` ends with `self.attach_plan(); self` instead of `self`
//!      (ntt.rs:260, ntt_avx2.rs:612, ntt_goldilock.rs:213);
//!   3. the bodies of `forward_inplace_core` / `backward_inplace_core` are deleted from the three files
//!      and this file is added to `src/dft.rs` as `mod dft_gpu;` -- the `impl` blocks below then are
//!      the only definitions of those methods.  The four `DFT` trait methods keep forwarding to them
//!      (ntt.rs:27-50 and the like) and need no change.
//!
//! Nothing else moves: constructors, `utils.rs`, the table loops and every public field stay, and the
//! tables the GPU plan is created from are the ones `with_precomputes` just filled in.
//!
//! Not compiled in this repository's build image (it has no Rust toolchain); the C++ mirror of the
//! same three types (ntt_amd/host/dft.hpp) is what the GPU tests run, and
//! tests/test_ffi_consistency.py checks that every `Plan` method used here exists in ffi.rs.
use crate::dft::ntt::Table;
use crate::dft::ntt_avx2::TableAVX2;
use crate::dft::ntt_goldilock::TableGoldilock;
use crate::ffi;

impl Table<u64> {
    /// Uploads the tables `with_precomputes` built (Montgomery form, bit-reversed order) once.
    pub(crate) fn attach_plan(&mut self) {
        self.plan = Some(ffi::Plan::create(
            ffi::NTTX_MONT64,
            self.q,
            self.montgomery.k,
            self.n,
            self.n_inv_montgomery,
            &self.powers_psi_bo,
            &self.powers_psi_inv_bo,
        ));
    }

    /// `LAZY` is accepted and not read, exactly as in the reference (ntt.rs:122): its lazy methods
    /// return the canonical outputs of the strict ones, and a drop-in has to return the same bits.
    pub fn forward_inplace_core<const LAZY: bool>(&self, a: &mut [u64]) {
        self.plan.as_ref().expect("table without a GPU plan").forward(a, self.n)
    }

    /// ntt.rs:178; includes the n^-1 scaling of ntt.rs:231-235.
    pub fn backward_inplace_core<const LAZY: bool>(&self, a: &mut [u64]) {
        self.plan.as_ref().expect("table without a GPU plan").backward(a, self.n)
    }

    /// Opt-in with no counterpart in the reference: outputs in `[0, 2q)`, the range the lazy methods
    /// document (ntt.rs:31-35, 45-49), congruent to the canonical ones; valid inputs of every method.
    pub fn forward_inplace_range_2q(&self, a: &mut [u64]) {
        self.plan.as_ref().expect("table without a GPU plan").forward_lazy(a, self.n)
    }
    pub fn backward_inplace_range_2q(&self, a: &mut [u64]) {
        self.plan.as_ref().expect("table without a GPU plan").backward_lazy(a, self.n)
    }
}

/// The AVX2 table works on `i64` lanes holding residues in `[0, q)`, q < 2^62 (ntt_avx2.rs:26-30): the
/// same bit patterns as `u64`, so the slice is reinterpreted and the MONT64 plan serves both types
/// (their outputs are equal by the reference's own test, dft.rs:45-78).
fn as_u64(a: &mut [i64]) -> &mut [u64] {
    unsafe { std::slice::from_raw_parts_mut(a.as_mut_ptr() as *mut u64, a.len()) }
}

impl TableAVX2 {
    pub(crate) fn attach_plan(&mut self) {
        let fwd: Vec<u64> = self.powers_psi_bo.iter().map(|&x| x as u64).collect();
        let inv: Vec<u64> = self.powers_psi_inv_bo.iter().map(|&x| x as u64).collect();
        self.plan = Some(ffi::Plan::create(
            ffi::NTTX_MONT64,
            self.q as u64,
            self.montgomery.k as u64,
            self.n as u64,
            self.n_inv_montgomery as u64,
            &fwd,
            &inv,
        ));
    }

    /// ntt_avx2.rs:108.  The scalar fallback the reference takes for slices of 8 or fewer elements is
    /// wrong there (ntt_avx2.rs:264-268 multiplies Montgomery-form values with a plain `%`); the plan
    /// is created for `self.n` and `Plan::forward` refuses any other length.
    pub fn forward_inplace_core<const LAZY: bool>(&self, a: &mut [i64]) {
        self.plan.as_ref().expect("table without a GPU plan").forward(as_u64(a), self.n as u64)
    }

    /// ntt_avx2.rs:281
    pub fn backward_inplace_core<const LAZY: bool>(&self, a: &mut [i64]) {
        self.plan.as_ref().expect("table without a GPU plan").backward(as_u64(a), self.n as u64)
    }
}

impl TableGoldilock<u64> {
    /// Plain residues, no Montgomery constant (k is ignored for this kind).
    pub(crate) fn attach_plan(&mut self) {
        self.plan = Some(ffi::Plan::create(
            ffi::NTTX_GOLDILOCKS,
            self.q,
            0,
            self.n,
            self.n_inv,
            &self.powers_psi_bo,
            &self.powers_psi_inv_bo,
        ));
    }

    /// ntt_goldilock.rs:81.  The kernels reproduce the reference's per-butterfly rules, including the
    /// non-canonical value q that `res > q` (ntt_goldilock.rs:107) lets through.
    pub fn forward_inplace_core<const LAZY: bool>(&self, a: &mut [u64]) {
        self.plan.as_ref().expect("table without a GPU plan").forward(a, self.n)
    }

    /// ntt_goldilock.rs:128
    pub fn backward_inplace_core<const LAZY: bool>(&self, a: &mut [u64]) {
        self.plan.as_ref().expect("table without a GPU plan").backward(a, self.n)
    }

    /// The reference fixes psi and n = 2^16 (ntt_goldilock.rs:48-58); BASELINE's 2^20 and 2^28 vectors
    /// need other sizes.  Same struct, same `with_precomputes`; n^-1 = q - (q - 1) / n for a power of
    /// two n dividing q - 1.
    pub fn from_root(psi: u64, n: u64) -> Self {
        let q: u64 = 0xffffffff00000001;
        assert!(n.is_power_of_two() && (q - 1) % n == 0, "n must be a power of two dividing q - 1");
        TableGoldilock {
            q,
            psi,
            n,
            n_inv: q - (q - 1) / n,
            powers_psi_bo: Vec::with_capacity(n as usize),
            powers_psi_inv_bo: Vec::with_capacity(n as usize),
            plan: None,
        }
        .with_precomputes()
    }
}
