/*
 * CPU oracle for the robertkiel/ntt transform path -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's three CPU implementations (scalar
 * Montgomery `Table<u64>`, AVX2 `TableAVX2`, scalar `TableGoldilock<u64>`) and
 * of the host precompute in `src/utils.rs`.  It is the checker for the HIP
 * path and the timed CPU baseline.  Nothing in the shipped library may link,
 * load or call it: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * The reference is Rust and no Rust toolchain exists in the build image, so the
 * reference itself was never executed here.  The restatement is pinned by the
 * known-answer tests the reference carries (utils.rs:200-222, ntt_avx2.rs:
 * 826-1030, ntt_goldilock.rs:247-274, the hard-coded n_inv at
 * ntt_goldilock.rs:53), by the reference's cross-implementation and round-trip
 * properties (dft.rs:23-189), and by an independent O(N^2) evaluation of the
 * closed form for small N.  The reference holds no full-transform golden
 * vector, so full transforms are pinned only through those.
 *
 * Every function cites the reference lines it follows.  "panic" in the
 * reference becomes a non-zero return here with a message in orc_last_panic().
 */
#ifndef NTT_ORACLE_H
#define NTT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned __int128 orc_u128;
typedef __int128 orc_i128;

/* ---- src/utils.rs ------------------------------------------------------- */
uint64_t orc_bit_reverse(uint64_t n, uint32_t bits);                 /* utils.rs:14-17 */
void orc_egcd_i128(orc_i128 a, orc_i128 m, orc_i128 out[3]);         /* utils.rs:51-64 */
/* 64-bit-halves view of egcd for ctypes callers: inputs/outputs as (hi, lo) two's complement */
void orc_egcd_i128_parts(int64_t a_hi, uint64_t a_lo, int64_t m_hi, uint64_t m_lo,
                         int64_t out_hi[3], uint64_t out_lo[3]);
uint64_t orc_mod_exp_u64(uint64_t base, uint64_t exp, uint64_t modulus); /* utils.rs:76-92 */
int64_t orc_mod_exp_i64(int64_t base, int64_t exp, int64_t modulus);     /* utils.rs:97 */

typedef struct {
    uint64_t k;        /* utils.rs:130 */
    uint64_t r_square; /* utils.rs:133 */
    uint64_t modulus;  /* utils.rs:134 */
} orc_montgomery;

int orc_montgomery_from_modulus(uint64_t modulus, orc_montgomery *out); /* utils.rs:141-162 */
uint64_t orc_montgomery_reduce(const orc_montgomery *m, uint64_t a_hi, uint64_t a_lo); /* utils.rs:165-178 */
uint64_t orc_montgomery_transform(const orc_montgomery *m, uint64_t a); /* utils.rs:181-183 */
uint64_t orc_montgomery_mul(const orc_montgomery *m, uint64_t a, uint64_t b); /* utils.rs:185-187 */

/* ---- src/dft/ntt.rs : Table<u64> ------------------------------------------ */
typedef struct {
    uint64_t q, psi;                 /* ntt.rs:8-10 */
    orc_montgomery montgomery;       /* ntt.rs:11 */
    uint64_t n_inv_montgomery;       /* ntt.rs:13 */
    uint64_t n;                      /* ntt.rs:14 */
    uint64_t psi_inv_montgomery;     /* ntt.rs:16 */
    uint64_t *powers_psi_bo;         /* ntt.rs:17 */
    uint64_t *powers_psi_inv_bo;     /* ntt.rs:20 */
} orc_table;

int orc_table_from_prime_and_root(uint64_t modulus, uint64_t root, uint64_t n, orc_table **out); /* ntt.rs:90-120 */
int orc_table_new(orc_table **out);                 /* ntt.rs:55-57, 264-268 */
int orc_table_new_goldilock(orc_table **out);       /* ntt.rs:60-62 */
int orc_table_new_u32_compatible(orc_table **out);  /* ntt.rs:66-68 */
int orc_table_new_simple(orc_table **out);          /* ntt.rs:72-75 (panics in the reference) */
void orc_table_free(orc_table *t);
int orc_table_forward_inplace(const orc_table *t, uint64_t *a, size_t len);  /* ntt.rs:122-176 */
int orc_table_backward_inplace(const orc_table *t, uint64_t *a, size_t len); /* ntt.rs:178-236 */
const uint64_t *orc_table_powers_psi_bo(const orc_table *t);
const uint64_t *orc_table_powers_psi_inv_bo(const orc_table *t);
uint64_t orc_table_n_inv_montgomery(const orc_table *t);
uint64_t orc_table_k(const orc_table *t);
uint64_t orc_table_r_square(const orc_table *t);

/* ---- src/dft/ntt_goldilock.rs : TableGoldilock<u64> ------------------------- */
typedef struct {
    uint64_t q, psi, n, n_inv;       /* ntt_goldilock.rs:8-12 */
    uint64_t *powers_psi_bo;         /* ntt_goldilock.rs:13 */
    uint64_t *powers_psi_inv_bo;     /* ntt_goldilock.rs:14 */
} orc_goldilock;

int orc_goldilock_new(orc_goldilock **out);         /* ntt_goldilock.rs:48-58 */
/* Same construction with psi / n supplied by the caller (the reference fixes them);
 * n_inv = q - (q-1)/n, which is 0xfffeffff00010001 for n = 2^16 as at ntt_goldilock.rs:53. */
int orc_goldilock_from_root(uint64_t psi, uint64_t n, orc_goldilock **out);
void orc_goldilock_free(orc_goldilock *t);
uint64_t orc_goldilock_reduce_159(const orc_goldilock *t, uint64_t low, uint64_t middle, uint64_t high); /* :63-79 */
uint64_t orc_goldilock_mul_reduce(const orc_goldilock *t, uint64_t a, uint64_t b); /* :183-190 */
int orc_goldilock_forward_inplace(const orc_goldilock *t, uint64_t *a, size_t len);  /* :81-126 */
int orc_goldilock_backward_inplace(const orc_goldilock *t, uint64_t *a, size_t len); /* :128-180 */
const uint64_t *orc_goldilock_powers_psi_bo(const orc_goldilock *t);
const uint64_t *orc_goldilock_powers_psi_inv_bo(const orc_goldilock *t);
uint64_t orc_goldilock_n_inv(const orc_goldilock *t);
void orc_u128_to_u64(uint64_t hi, uint64_t lo, uint64_t out[2]); /* :218-227: returns (high, low) */
void orc_u64_to_u128(uint64_t first, uint64_t second, uint64_t out_hi_lo[2]); /* :231-238 */

/* ---- src/dft/ntt_avx2.rs : TableAVX2 (compiled with -mavx2) ------------------- */
typedef struct {
    int64_t q, psi;                  /* ntt_avx2.rs:9-11 */
    int64_t psi_inv_montgomery;      /* :13 */
    size_t n;                        /* :15 */
    int64_t n_inv_montgomery;        /* :18 */
    orc_montgomery montgomery;       /* :19 (MontgomeryTransformer<i64>; same bits as u64) */
    int64_t *powers_psi_bo;          /* :21 */
    int64_t *powers_psi_inv_bo;      /* :23 */
} orc_avx2;

int orc_avx2_from_prime_and_root(int64_t modulus, int64_t root, int64_t n, orc_avx2 **out); /* :74-105 */
int orc_avx2_new(orc_avx2 **out);                    /* :57-59 */
void orc_avx2_free(orc_avx2 *t);
int orc_avx2_forward_inplace(const orc_avx2 *t, int64_t *a, size_t len);  /* :108-279 */
int orc_avx2_backward_inplace(const orc_avx2 *t, int64_t *a, size_t len); /* :281-470 */
const int64_t *orc_avx2_powers_psi_bo(const orc_avx2 *t);
const int64_t *orc_avx2_powers_psi_inv_bo(const orc_avx2 *t);
/* 4-lane primitives, arrays of 4 in / out, for the known-answer tests */
void orc_avx2_a_mul_b_karatsuba(const int64_t a[4], const int64_t b[4], int64_t high[4], int64_t low[4]); /* :747-779 */
void orc_avx2_a_lt_b(const int64_t a[4], const int64_t b[4], int64_t out[4]);                           /* :728-744 */
void orc_avx2_a_sub_b(const int64_t ah[4], const int64_t al[4], const int64_t bh[4], const int64_t bl[4],
                      int64_t high[4], int64_t low[4]);                                                   /* :782-796 */
void orc_avx2_a_lt_b_wide(const int64_t ah[4], const int64_t al[4], const int64_t bh[4], const int64_t bl[4],
                          int64_t out[4]);                                                                /* :799-813 */
void orc_avx2_montgomery_reduce_64(const orc_avx2 *t, const int64_t th[4], const int64_t tl[4], int64_t out[4]); /* :623-644 */
void orc_avx2_montgomery_mul_vec(const orc_avx2 *t, const int64_t a[4], const int64_t b[4], int64_t out[4]);     /* :554-559 */
void orc_avx2_to_montgomery(const orc_avx2 *t, const int64_t a[4], int64_t out[4]);                              /* :674-678 */
void orc_avx2_reduce_if_greater_equal_q(const orc_avx2 *t, int64_t a[4]);                                        /* :563-577 */
void orc_avx2_reduce_if_negative(const orc_avx2 *t, int64_t a[4]);                                               /* :581-589 */
void orc_avx2_cs_butterfly_simd_4(const orc_avx2 *t, const int64_t s[4], int64_t a_j[4], int64_t a_j_t[4]);      /* :518-533 */
void orc_avx2_gs_butterfly_simd_4(const orc_avx2 *t, const int64_t s[4], int64_t a_j[4], int64_t a_j_t[4]);      /* :537-550 */

/* ---- independent closed form (not in the reference; pins the restatement) ---- */
/* out[i] = sum_j a[j] * psi^(j*(2*brev(i)+1)) mod q, O(N^2), true bit reversal. */
void orc_closed_form_forward(uint64_t q, uint64_t psi, const uint64_t *a, uint64_t *out, size_t n);

/* ---- CPU baseline timing (main.rs:24-54 shape) ------------------------------ */
/* Runs `pairs` x {forward_inplace; backward_inplace} on each of `threads` private
 * buffers of length n filled with SplitMix64 residues; kind: 0 base, 1 avx2, 2 goldilock.
 * `table` is the matching orc_* table.  Returns wall seconds, <0 on error. */
double orc_bench_pairs(int kind, const void *table, size_t n, int pairs, int threads);

const char *orc_last_panic(void);

#ifdef __cplusplus
}
#endif
#endif
