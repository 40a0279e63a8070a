//! What changes inside the reference's three table types.  Constructors and `with_precomputes`
//! stay as they are (host precompute); each struct gains a `plan: ffi::Plan` field filled at the
//! end of construction, and the bodies of the `*_inplace_core` methods become one call.
//!
//! ```ignore
//! // src/dft/ntt.rs -- Table<u64>
//! fn with_precomputes(mut self) -> Self {
//!     /* unchanged table loop (ntt.rs:239-258) */
//!     self.plan = Some(ffi::Plan::create(ffi::NTTX_MONT64, self.q, self.montgomery.k, self.n,
//!                                        self.n_inv_montgomery, &self.powers_psi_bo, &self.powers_psi_inv_bo));
//!     self
//! }
//! pub fn forward_inplace_core<const LAZY: bool>(&self, a: &mut [u64]) {
//!     self.plan.as_ref().unwrap().forward(a, self.n)
//! }
//! pub fn backward_inplace_core<const LAZY: bool>(&self, a: &mut [u64]) {
//!     self.plan.as_ref().unwrap().backward(a, self.n)
//! }
//!
//! // src/dft/ntt_avx2.rs -- TableAVX2: same plan kind; the i64 slice is reinterpreted
//! pub fn forward_inplace_core<const LAZY: bool>(&self, a: &mut [i64]) {
//!     let a = unsafe { std::slice::from_raw_parts_mut(a.as_mut_ptr() as *mut u64, a.len()) };
//!     self.plan.as_ref().unwrap().forward(a, self.n as u64)
//! }
//!
//! // src/dft/ntt_goldilock.rs -- TableGoldilock<u64>
//! self.plan = Some(ffi::Plan::create(ffi::NTTX_GOLDILOCKS, self.q, 0, self.n, self.n_inv,
//!                                    &self.powers_psi_bo, &self.powers_psi_inv_bo));
//! ```
