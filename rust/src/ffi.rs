//! Raw bindings of include/nttx.h plus the one helper the table types share.
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct NttxPlan {
    _private: [u8; 0],
}

pub const NTTX_OK: c_int = 0;
pub const NTTX_MONT64: c_int = 0;
pub const NTTX_GOLDILOCKS: c_int = 1;

extern "C" {
    pub fn nttx_plan_create(
        kind: c_int, q: u64, k: u64, n: u64, n_inv: u64,
        psi_bo: *const u64, psi_inv_bo: *const u64,
        devices: *const c_int, n_devices: c_int, out: *mut *mut NttxPlan,
    ) -> c_int;
    pub fn nttx_plan_create_from_root(
        kind: c_int, q: u64, psi: u64, n: u64,
        devices: *const c_int, n_devices: c_int, out: *mut *mut NttxPlan,
    ) -> c_int;
    /// Not reference behaviour: k = q^-1 mod 2^64 for every odd q (utils.rs:148-152 gets about
    /// half of all moduli wrong, and the two constructors above reject those); also accepts odd
    /// log2 n, with tables at the true bit reversal (utils.rs:14-17 panics there).
    pub fn nttx_plan_create_from_root_exact_k(
        kind: c_int, q: u64, psi: u64, n: u64,
        devices: *const c_int, n_devices: c_int, out: *mut *mut NttxPlan,
    ) -> c_int;
    pub fn nttx_plan_destroy(plan: *mut NttxPlan);
    pub fn nttx_plan_info(plan: *const NttxPlan, kind: *mut c_int, q: *mut u64, k: *mut u64, n: *mut u64,
                          n_inv: *mut u64, n_devices: *mut c_int) -> c_int;
    pub fn nttx_plan_tables(plan: *const NttxPlan, psi_bo_out: *mut u64, psi_inv_bo_out: *mut u64) -> c_int;
    pub fn nttx_plan_policy(plan: *const NttxPlan) -> *const c_char;
    pub fn nttx_forward_host(plan: *const NttxPlan, data: *mut u64, batch: usize) -> c_int;
    pub fn nttx_backward_host(plan: *const NttxPlan, data: *mut u64, batch: usize) -> c_int;
    pub fn nttx_forward_device(plan: *const NttxPlan, device: c_int, dptr: *mut c_void, batch: usize, hip_stream: *mut c_void) -> c_int;
    pub fn nttx_backward_device(plan: *const NttxPlan, device: c_int, dptr: *mut c_void, batch: usize, hip_stream: *mut c_void) -> c_int;
    /// forward_host / backward_host staged through a static device array: no allocation, at most 32 MiB.
    pub fn nttx_transform_host_noalloc(plan: *const NttxPlan, data: *mut u64, batch: usize, dir: c_int) -> c_int;
    /// Outputs in [0, 2q): what DFT::forward_inplace_lazy / backward_inplace_lazy document (ntt.rs:31-35, 45-49).
    pub fn nttx_forward_host_lazy(plan: *const NttxPlan, data: *mut u64, batch: usize) -> c_int;
    pub fn nttx_backward_host_lazy(plan: *const NttxPlan, data: *mut u64, batch: usize) -> c_int;
    pub fn nttx_forward_device_lazy(plan: *const NttxPlan, device: c_int, dptr: *mut c_void, batch: usize, hip_stream: *mut c_void) -> c_int;
    pub fn nttx_backward_device_lazy(plan: *const NttxPlan, device: c_int, dptr: *mut c_void, batch: usize, hip_stream: *mut c_void) -> c_int;
    pub fn nttx_negacyclic_mul_host(plan: *const NttxPlan, a: *mut u64, b: *const u64, batch: usize) -> c_int;
    pub fn nttx_negacyclic_mul_device(plan: *const NttxPlan, device: c_int, a: *mut c_void, b: *mut c_void, batch: usize, hip_stream: *mut c_void) -> c_int;
    pub fn nttx_plan_feature(plan: *const NttxPlan, name: *const c_char) -> c_int;
    pub fn nttx_pass_count(plan: *const NttxPlan) -> c_int;
    pub fn nttx_pass_describe(plan: *const NttxPlan, dir: c_int, pass_index: c_int, buf: *mut c_char, buf_len: usize) -> c_int;
    pub fn nttx_run_pass_device(plan: *const NttxPlan, device: c_int, dptr: *mut c_void, batch: usize, dir: c_int,
                                pass_index: c_int, hip_stream: *mut c_void) -> c_int;
    /// {shader cycles, 100 MHz ticks} of a one-wave spin of spin_us microseconds, into device memory out2.
    pub fn nttx_clock_probe_device(device: c_int, out2: *mut u64, spin_us: c_int, hip_stream: *mut c_void) -> c_int;
    pub fn nttx_forward_distributed(plan: *const NttxPlan, host_data: *mut u64, n_shards: c_int) -> c_int;
    pub fn nttx_backward_distributed(plan: *const NttxPlan, host_data: *mut u64, n_shards: c_int) -> c_int;
    pub fn nttx_shard_geometry(plan: *const NttxPlan, n_shards: c_int, local_elems: *mut u64, rows: *mut u64,
                               cols: *mut u64, rows_per_peer: *mut u64) -> c_int;
    pub fn nttx_shard_phase_device(plan: *const NttxPlan, device: c_int, dptr: *mut c_void, dir: c_int, phase: c_int,
                                   shard: c_int, n_shards: c_int, hip_stream: *mut c_void) -> c_int;
    pub fn nttx_device_count() -> c_int;
    pub fn nttx_last_error() -> *const c_char;
    pub fn nttx_version() -> *const c_char;
}

/// Owned plan handle.  A plan is immutable after creation and the library allows concurrent
/// calls on it, so the table types stay `Send + Sync` as in the reference.
pub struct Plan(pub *mut NttxPlan);
unsafe impl Send for Plan {}
unsafe impl Sync for Plan {}

impl Drop for Plan {
    fn drop(&mut self) {
        unsafe { nttx_plan_destroy(self.0) }
    }
}

/// The reference reports errors by panicking; a non-zero status keeps that contract.
pub fn check(rc: c_int) {
    if rc != NTTX_OK {
        let msg = unsafe { CStr::from_ptr(nttx_last_error()) }.to_string_lossy().into_owned();
        panic!("nttx: {msg}");
    }
}

impl Plan {
    pub fn create(kind: c_int, q: u64, k: u64, n: u64, n_inv: u64, psi_bo: &[u64], psi_inv_bo: &[u64]) -> Self {
        assert_eq!(psi_bo.len() as u64, n);
        assert_eq!(psi_inv_bo.len() as u64, n);
        let mut p = std::ptr::null_mut();
        check(unsafe {
            nttx_plan_create(kind, q, k, n, n_inv, psi_bo.as_ptr(), psi_inv_bo.as_ptr(), std::ptr::null(), 0, &mut p)
        });
        Plan(p)
    }

    /// `a.len()` must equal the table size: the reference sizes the transform by the slice
    /// (ntt.rs:123) but only that length round-trips, and a longer one indexes out of bounds.
    pub fn forward(&self, a: &mut [u64], n: u64) {
        assert_eq!(a.len() as u64, n, "slice length does not match the table size");
        check(unsafe { nttx_forward_host(self.0, a.as_mut_ptr(), 1) });
    }
    pub fn backward(&self, a: &mut [u64], n: u64) {
        assert_eq!(a.len() as u64, n, "slice length does not match the table size");
        check(unsafe { nttx_backward_host(self.0, a.as_mut_ptr(), 1) });
    }
    /// `*_inplace_core::<true>`: outputs in [0, 2q) (ntt.rs:31-35, 45-49).
    pub fn forward_lazy(&self, a: &mut [u64], n: u64) {
        assert_eq!(a.len() as u64, n, "slice length does not match the table size");
        check(unsafe { nttx_forward_host_lazy(self.0, a.as_mut_ptr(), 1) });
    }
    pub fn backward_lazy(&self, a: &mut [u64], n: u64) {
        assert_eq!(a.len() as u64, n, "slice length does not match the table size");
        check(unsafe { nttx_backward_host_lazy(self.0, a.as_mut_ptr(), 1) });
    }
}
