/*
 * nttx -- MI355X (gfx950) engine for the transform path of robertkiel/ntt.
 *
 * C ABI of libnttx.so.  The reference is a pure-Rust crate with no FFI of its
 * own: the boundary it exposes is the `DFT<O>` trait (src/dft.rs:6-11) and the
 * three table types that implement it.  Each entry point below names the
 * reference interface whose body it replaces.  A Rust `extern "C"` block that
 * binds exactly these symbols is rust/src/ffi.rs.
 *
 * Conventions: plain pointers and sizes; no exceptions cross the boundary;
 * every function returns an nttx_status (0 = ok) and leaves a message readable
 * through nttx_last_error() on failure; the caller owns every buffer it passes;
 * a plan is immutable after creation and may be used from several host threads.
 * The reference reports errors by panicking; a Rust shim maps a non-zero status
 * to panic!() to keep that contract.
 */
#ifndef NTTX_H
#define NTTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nttx_plan nttx_plan;

enum nttx_status {
    NTTX_OK = 0,
    NTTX_EINVAL = 1, /* argument outside the reference's own domain (would panic or be wrong there) */
    NTTX_ENODEV = 2, /* no usable HIP device / bad device index */
    NTTX_ENOMEM = 3, /* host or device allocation failed */
    NTTX_EHIP = 4    /* a HIP runtime call failed */
};

enum nttx_kind {
    NTTX_MONT64 = 0,    /* Table<u64> (src/dft/ntt.rs) and TableAVX2 (src/dft/ntt_avx2.rs): identical outputs */
    NTTX_GOLDILOCKS = 1 /* TableGoldilock<u64> (src/dft/ntt_goldilock.rs) */
};

/*
 * Creates a plan from tables the host computed exactly as the reference does.
 *
 * Replaces the table-owning half of Table::from_prime_and_root (ntt.rs:90-120),
 * TableAVX2::from_prime_and_root (ntt_avx2.rs:74-105) and TableGoldilock::new
 * (ntt_goldilock.rs:48-58): the struct keeps q/psi and calls this with
 *   NTTX_MONT64:     k = MontgomeryTransformer.k (utils.rs:130), n_inv = n_inv_montgomery
 *                    (ntt.rs:13), psi_bo / psi_inv_bo = powers_psi_bo / powers_psi_inv_bo
 *                    (ntt.rs:17-20), all in Montgomery form as with_precomputes (ntt.rs:239-261)
 *                    leaves them;
 *   NTTX_GOLDILOCKS: q must be 0xffffffff00000001, k is ignored, n_inv and both tables
 *                    are plain residues (ntt_goldilock.rs:12-14, 192-214).
 * n must be a power of two, 4 <= n <= 2^32.  The reference only builds tables for even log2 n
 * (utils.rs:14-17 makes every odd log2 n panic); tables for odd log2 n are taken to be in true
 * bit-reversed order (entry bit_reverse(i) = psi^i), the order the index structure of
 * ntt.rs:129-171 needs.  For NTTX_MONT64 q must be odd (utils.rs:144-146) and k*q
 * must be 1 mod 2^64 (utils.rs:148-152 yields a wrong k, hence wrong transforms, otherwise).
 * Tables (n entries each) are copied to every listed device; devices == NULL / n_devices == 0
 * means device 0 only.
 */
int nttx_plan_create(int kind, uint64_t q, uint64_t k, uint64_t n, uint64_t n_inv,
                     const uint64_t *psi_bo, const uint64_t *psi_inv_bo,
                     const int *devices, int n_devices, nttx_plan **out);

/*
 * Same, with the host precompute done inside the library from (q, psi, n):
 * a C++ restatement of MontgomeryTransformer::from_modulus (utils.rs:141-162),
 * Table::from_prime_and_root + with_precomputes (ntt.rs:90-120, 239-261) and
 * TableGoldilock::with_precomputes (ntt_goldilock.rs:192-214).  For callers with no
 * host-side table code (tests, the benchmark, C and Python users).  Like those constructors it
 * refuses odd log2 n (they panic: utils.rs:14-17, ntt.rs:253).
 */
int nttx_plan_create_from_root(int kind, uint64_t q, uint64_t psi, uint64_t n,
                               const int *devices, int n_devices, nttx_plan **out);

/*
 * Same as nttx_plan_create_from_root, but with k = q^-1 mod 2^64 for every odd modulus.  No
 * counterpart in the reference: MontgomeryTransformer::from_modulus (utils.rs:148-152) lifts a
 * negative Bezout coefficient by the modulus instead of 2^64 and so derives a k with k*q != 1 for
 * about half of all moduli (12289, 65537, 786433 ...), for which the reference's transforms are
 * wrong and nttx_plan_create / nttx_plan_create_from_root return NTTX_EINVAL.  With this entry
 * point those moduli work; where the reference's k is right the plan is identical.  It also lifts
 * the reference's other setup defect: odd log2 n is accepted, with tables in true bit-reversed
 * order (utils.rs:14-17 returns twice the reversal for an odd bit count).  Such plans run the
 * batched entry points and the product; they are not split over shards.
 */
int nttx_plan_create_from_root_exact_k(int kind, uint64_t q, uint64_t psi, uint64_t n,
                                       const int *devices, int n_devices, nttx_plan **out);

void nttx_plan_destroy(nttx_plan *plan);

/* Read-back of what a plan holds (for parity tests against the reference's pub fields
 * Table::powers_psi_bo / powers_psi_inv_bo, ntt.rs:17-20).  Any out pointer may be NULL. */
int nttx_plan_info(const nttx_plan *plan, int *kind, uint64_t *q, uint64_t *k, uint64_t *n,
                   uint64_t *n_inv, int *n_devices);
int nttx_plan_tables(const nttx_plan *plan, uint64_t *psi_bo_out, uint64_t *psi_inv_bo_out);
/* Name of the arithmetic the plan's kernels use (no counterpart in the reference; for tests and
 * logs): "shoup61" / "shoup61_near" (NTTX_MONT64, q < 2^61: lazy butterflies on plain residues with
 * precomputed quotients; "_near" for q within 2^29 of 2^61), "mont" / "mont_wide" (NTTX_MONT64,
 * 2^61 <= q < 2^63 / q >= 2^63: strict REDC with R = 2^64 as utils.rs:165-187),
 * "goldilocks_canonical" (NTTX_MONT64 with q = 0xffffffff00000001, Table::new_goldilock,
 * ntt.rs:60-62), "goldilocks" (NTTX_GOLDILOCKS).  Static string; "" for a NULL plan. */
const char *nttx_plan_policy(const nttx_plan *plan);
/*
 * Whether a plan's kernels take one of the shortcuts that depend on its constants (1 / 0; -1 for an
 * unknown name or a NULL plan).  No counterpart in the reference; tests use it to assert that the
 * path they mean to check is the one that runs.
 *   "pow2_twiddles"  NTTX_GOLDILOCKS arithmetic: the twiddles of the first four forward / last four
 *                    inverse stages are applied as shifts (entries 1..15 of both tables are the powers
 *                    of two of a table with psi^(n/32) = 2^39, e.g. psi = 7^((q-1)/2n))
 *   "shift_scaling"  q < 2^61: the n^-1 factor of the inverse transform's sums (ntt.rs:231-235) is applied
 *                    as an exact division by n = 2^k instead of a product (n^-1 = 2^-k mod q, 8 <= k <= 24)
 */
int nttx_plan_feature(const nttx_plan *plan, const char *name);

/*
 * In-place transforms of `batch` independent length-n vectors stored back to back in HOST
 * memory (batch*n u64, little endian).  batch == 1 is DFT::forward_inplace / backward_inplace
 * (dft.rs:7,9) for Table<u64> (ntt.rs:27-29, 41-43), TableAVX2 (ntt_avx2.rs:30-32, 44-46; pass
 * the i64 slice reinterpreted: same outputs as the AVX2 code for its documented inputs, residues
 * in [0, q) -- other bit patterns are read as u64 and reduced mod q, where the AVX2 lanes are signed) and TableGoldilock<u64> (ntt_goldilock.rs:21-23, 35-37); the
 * *_lazy trait methods (dft.rs:8,10) may bind the same symbols (the reference's LAZY flag is never
 * read, ntt.rs:122, 178) or the *_lazy entry points below.  batch > 1 has no counterpart in the reference (it loops).
 * Copies to the plan's devices, splits the batch evenly across them, transforms, copies back.
 * Forward: natural order in, bit-reversed order out.  Backward: bit-reversed in, natural out,
 * scaled by n^-1.
 * A share of 16 MiB or more per device is streamed in chunks of whole transforms (2 to 16 MiB)
 * through pinned memory, host threads copying between the caller's slice and the pinned chunks
 * (NTTX_HOST_COPY_THREADS, default 8) while uploads, kernels and downloads of neighbouring
 * chunks overlap; the plan then keeps 128 MiB of pinned host memory and 64 MiB of device memory
 * per device until it is destroyed.  Smaller calls copy straight from the caller's memory.
 */
int nttx_forward_host(const nttx_plan *plan, uint64_t *data, size_t batch);
int nttx_backward_host(const nttx_plan *plan, uint64_t *data, size_t batch);

/*
 * Same transforms on data already resident in the memory of `device` (one of the plan's
 * devices), enqueued on `hip_stream` (a hipStream_t, NULL = default stream) without any
 * host synchronisation.  This is the measured path; no counterpart in the reference.
 */
int nttx_forward_device(const nttx_plan *plan, int device, void *dptr, size_t batch, void *hip_stream);
int nttx_backward_device(const nttx_plan *plan, int device, void *dptr, size_t batch, void *hip_stream);

/*
 * nttx_forward_host (dir = 0) / nttx_backward_host (dir = 1) for callers that cannot allocate.
 * No counterpart in the reference.  It is for harnesses that forbid hipMalloc, tight serving loops
 * and similar callers: the batch is staged in a 32 MiB array that is part of the library's code
 * object (every GPU the library runs on holds one from load time to process exit), and the call
 * makes no hipMalloc, hipFree, stream or event: copies and kernels run on the default stream of the
 * plan's FIRST device, and the call returns when the result is back in `data`.  Results are those of
 * nttx_forward_host / nttx_backward_host bit for bit.  batch * n * 8 bytes must not exceed 32 MiB
 * (64 vectors of 2^16, one of 2^22): larger calls fail with NTTX_EINVAL.  Every call on a GPU
 * shares that GPU's array -- the upload overwrites whatever the last call left there, nothing needs
 * clearing -- so two calls in flight on one GPU would race on it: the library takes a per-device lock
 * for the whole call, which makes calls from several threads or plans of one process take turns.
 */
int nttx_transform_host_noalloc(const nttx_plan *plan, uint64_t *data, size_t batch, int dir);

/*
 * The same transforms with LAZY outputs: DFT::forward_inplace_lazy / backward_inplace_lazy
 * (dft.rs:8,10), whose contract is "elements in [0, 2q)" (ntt.rs:31-35, 45-49; the reference's
 * LAZY parameter is never read, so its lazy methods return canonical residues, which satisfy that
 * contract too).  For NTTX_MONT64 plans with q < 2^61 the final canonicalising step is replaced by
 * a cheaper one that leaves every output in [0, 2q), congruent to the canonical value; such outputs
 * are valid inputs of every entry point here (inputs below 4q are taken as they are).  All other
 * plans (q >= 2^61, NTTX_GOLDILOCKS) return exactly what the strict entry points return.
 */
int nttx_forward_host_lazy(const nttx_plan *plan, uint64_t *data, size_t batch);
int nttx_backward_host_lazy(const nttx_plan *plan, uint64_t *data, size_t batch);
int nttx_forward_device_lazy(const nttx_plan *plan, int device, void *dptr, size_t batch, void *hip_stream);
int nttx_backward_device_lazy(const nttx_plan *plan, int device, void *dptr, size_t batch, void *hip_stream);

/*
 * Negacyclic product a <- a * b in Z_q[X]/(X^n + 1) for `batch` pairs of polynomials (batch*n
 * coefficients each, natural order in and out).  No counterpart in the reference: it is the
 * composition its callers write around the transforms -- forward_inplace(a), forward_inplace(b)
 * (ntt.rs:27-29 / ntt_goldilock.rs:21-23), coefficient-wise product mod q, backward_inplace
 * (ntt.rs:41-43 / ntt_goldilock.rs:35-37) -- with the product fused into the first pass of the
 * backward transform, so it costs no pass over memory of its own; where the last forward pass and
 * the first backward pass work on the same tiles (log2 n a multiple of 8, 16 pairs or more) those
 * two passes of a, the last pass of b and the product are one launch.  The coefficient-wise
 * product is the canonical residue (for NTTX_GOLDILOCKS: mul_reduce, ntt_goldilock.rs:183-190,
 * which is canonical even for the value q the forward transform can emit), so the result equals
 * that composition bit for bit.  The device entry point uses b as scratch: its contents afterwards
 * are unspecified (the host entry point leaves the caller's b untouched).  Passing the same
 * pointer as a and b squares; partially overlapping operands are rejected (both entry points).
 */
int nttx_negacyclic_mul_host(const nttx_plan *plan, uint64_t *a, const uint64_t *b, size_t batch);
int nttx_negacyclic_mul_device(const nttx_plan *plan, int device, void *a, void *b, size_t batch,
                               void *hip_stream);

/*
 * Profiling hooks (no counterpart in the reference).  One forward or backward call issues
 * nttx_pass_count() kernel launches ("passes"); dir = 0 forward, 1 backward, and pass indices
 * follow execution order for that direction.  nttx_pass_describe writes a short name such as
 * "fwd strided 4+4 stages 1-8" into buf ("fwd fused 4+8 stages 9-20": the launch that runs the
 * last 8 + 2 or 8 + 4 stages when log2 n is 2 or 4 mod 8).  nttx_run_pass_device enqueues ONE pass alone so a
 * benchmark can bracket it with events on its own stream; run out of sequence it leaves
 * meaningless data in dptr, so it is for timing only.
 */
int nttx_pass_count(const nttx_plan *plan);
int nttx_pass_describe(const nttx_plan *plan, int dir, int pass_index, char *buf, size_t buf_len);
int nttx_run_pass_device(const nttx_plan *plan, int device, void *dptr, size_t batch, int dir,
                         int pass_index, void *hip_stream);
/*
 * Shader clock under load (no counterpart in the reference; a measurement aid like the three above).
 * Enqueues on hip_stream a one-wave kernel that reads the shader-clock counter and the constant
 * 100 MHz counter, spins for about spin_us microseconds (at most 100 000) and reads both again;
 * out2 (device memory, two uint64_t) receives {shader cycles, 100 MHz ticks} elapsed, so the clock
 * in MHz is 100 * out2[0] / out2[1].  Enqueued on a second stream while transforms run on the
 * first, it reports the clock the chip holds under that load, which is what prices a kernel's
 * instruction count in time.
 */
int nttx_clock_probe_device(int device, uint64_t *out2, int spin_us, void *hip_stream);

/*
 * ONE transform of n elements split over n_shards = 2^k shards (a single vector too large or
 * too slow for one GPU; the reference runs any length in one loop nest, ntt.rs:122-176 and
 * :178-236, ntt_goldilock.rs:81-180 -- these calls compute the same result).  The split
 * follows the reference's index structure: the first f butterfly stages (f = log2 n mod 8, or 8
 * when that is 0; nttx_shard_geometry reports rows = 2^f) only couple elements whose indices
 * agree modulo n/2^f, and after them the vector is 2^f independent contiguous blocks.
 *
 *   column layout of shard d: a [rows = 2^f][cols = n/(2^f * n_shards)] array holding, of every
 *     row of n/2^f consecutive elements of the vector, the columns [d*cols, (d+1)*cols).
 *   block layout of shard c: elements [c*n/n_shards, (c+1)*n/n_shards) of the vector.
 *   exchange: shard d's rows [c*rows_per_peer, (c+1)*rows_per_peer) (one contiguous piece) go to
 *     shard c, which stores row rho of that piece at offset (rho*n_shards + d)*cols.
 *
 * Forward (natural order in column layout -> bit-reversed order in block layout):
 *   phase 0 on every shard, exchange, phase 1.   Backward is the mirror image: phase 1 on the
 *   block layout, the inverse exchange, phase 0 on the column layout (which applies n^-1).
 *
 * nttx_forward_distributed / nttx_backward_distributed do all of it for a vector in HOST memory
 * from one process: shard i runs on plan device i % n_devices (n_shards = 0 means one shard per
 * plan device), pieces move with device-to-device copies.  With a single device all shards share
 * it, so the same code path can be tested on one GPU.
 *
 * nttx_shard_geometry and nttx_shard_phase_device are the building blocks for callers that run
 * one process per GPU and do the exchange themselves (an all-to-all over RCCL):
 * nttx_shard_phase_device enqueues the passes of one phase on one shard's device-resident buffer
 * of local_elems elements.  dir = 0 forward, 1 backward.
 *
 * Limits: n_shards a power of two <= 2^f (f = 8 when log2 n is a multiple of 8, else log2 n mod
 * 8), n >= 2^10 and n/n_shards >= 4096.  For the N = 2^28 vector over 8 GPUs f = 4.
 */
int nttx_forward_distributed(const nttx_plan *plan, uint64_t *host_data, int n_shards);
int nttx_backward_distributed(const nttx_plan *plan, uint64_t *host_data, int n_shards);
int nttx_shard_geometry(const nttx_plan *plan, int n_shards, uint64_t *local_elems, uint64_t *rows,
                        uint64_t *cols, uint64_t *rows_per_peer);
int nttx_shard_phase_device(const nttx_plan *plan, int device, void *dptr, int dir, int phase, int shard,
                            int n_shards, void *hip_stream);

int nttx_device_count(void);
const char *nttx_last_error(void); /* thread-local, never NULL */
const char *nttx_version(void);

#ifdef __cplusplus
}
#endif
#endif
