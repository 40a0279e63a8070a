/*
 * CPU oracle, scalar paths -- TEST INFRASTRUCTURE ONLY (see ntt_oracle.h).
 *
 * Restates src/utils.rs, src/dft/ntt.rs and src/dft/ntt_goldilock.rs of the
 * reference in plain C with unsigned __int128.  Kept deliberately dumb: loop
 * structure, operation order and the reference's quirks are preserved so that
 * its outputs are the reference's outputs bit for bit.
 */
#define _GNU_SOURCE
#include "ntt_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static __thread char g_panic[256];

const char *orc_last_panic(void) { return g_panic; }

static int panic_(const char *msg)
{
    snprintf(g_panic, sizeof g_panic, "%s", msg);
    return 1;
}

/* ------------------------------------------------------------------ utils.rs */

/* u64::reverse_bits */
static uint64_t reverse_bits64(uint64_t x)
{
    uint64_t r = 0;
    for (int i = 0; i < 64; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

/* utils.rs:14-17.  shift = (64 - bits) / 2 -- an integer division, so the result
 * is the true bit reversal only for even `bits`; odd `bits` gives twice it. */
uint64_t orc_bit_reverse(uint64_t n, uint32_t bits)
{
    uint32_t shift = (64 - bits) / 2;
    return reverse_bits64(n << shift) >> shift;
}

/* utils.rs:51-64 */
void orc_egcd_i128(orc_i128 a, orc_i128 m, orc_i128 out[3])
{
    orc_i128 r0 = a, r1 = m;
    orc_i128 s0 = 1, s1 = 0;
    orc_i128 t0 = 0, t1 = 1;

    while (r1 != 0) {
        orc_i128 q = r0 / r1;
        orc_i128 tmp;
        tmp = r0 - q * r1; r0 = r1; r1 = tmp;
        tmp = s0 - q * s1; s0 = s1; s1 = tmp;
        tmp = t0 - q * t1; t0 = t1; t1 = tmp;
    }
    out[0] = r0;
    out[1] = s0;
    out[2] = t0;
}

void orc_egcd_i128_parts(int64_t a_hi, uint64_t a_lo, int64_t m_hi, uint64_t m_lo,
                         int64_t out_hi[3], uint64_t out_lo[3])
{
    orc_i128 a = ((orc_i128)a_hi << 64) | (orc_i128)(orc_u128)a_lo;
    orc_i128 m = ((orc_i128)m_hi << 64) | (orc_i128)(orc_u128)m_lo;
    orc_i128 o[3];
    orc_egcd_i128(a, m, o);
    for (int i = 0; i < 3; i++) {
        out_hi[i] = (int64_t)(o[i] >> 64);
        out_lo[i] = (uint64_t)o[i];
    }
}

/* utils.rs:76-92 with ($t, $t_larger) = (u64, u128) */
uint64_t orc_mod_exp_u64(uint64_t base, uint64_t exp, uint64_t modulus)
{
    uint64_t out = 1;
    uint64_t acc = base;
    while (exp > 0) {
        if (exp % 2 == 1)
            out = (uint64_t)(((orc_u128)out * (orc_u128)acc) % (orc_u128)modulus);
        acc = (uint64_t)(((orc_u128)acc * (orc_u128)acc) % (orc_u128)modulus);
        exp >>= 1;
    }
    return out;
}

/* utils.rs:76-92 with (i64, i128) */
int64_t orc_mod_exp_i64(int64_t base, int64_t exp, int64_t modulus)
{
    int64_t out = 1;
    int64_t acc = base;
    while (exp > 0) {
        if (exp % 2 == 1)
            out = (int64_t)(((orc_i128)out * (orc_i128)acc) % (orc_i128)modulus);
        acc = (int64_t)(((orc_i128)acc * (orc_i128)acc) % (orc_i128)modulus);
        exp >>= 1;
    }
    return out;
}

/* utils.rs:141-162 */
int orc_montgomery_from_modulus(uint64_t modulus, orc_montgomery *out)
{
    orc_i128 g[3];
    orc_egcd_i128((orc_i128)modulus, (orc_i128)1 << 64, g);
    if (g[0] != 1)
        return panic_("modulus is not coprime to helper modulus R=2^64");

    orc_i128 x = g[1];
    /* utils.rs:148-152: a negative x is lifted by the modulus, not by 2^64 */
    uint64_t k = (x < 0) ? (uint64_t)(x + (orc_i128)modulus) : (uint64_t)x;

    orc_u128 r_mod_n = ((orc_u128)1 << 64) % (orc_u128)modulus;
    uint64_t r_square = (uint64_t)((r_mod_n * r_mod_n) % (orc_u128)modulus);

    out->k = k;
    out->r_square = r_square;
    out->modulus = modulus;
    return 0;
}

/* utils.rs:165-178 */
static inline uint64_t mont_reduce(const orc_montgomery *mt, orc_u128 a)
{
    const orc_u128 mask = (orc_u128)0xffffffffffffffffULL;
    orc_u128 m = ((a & mask) * (orc_u128)mt->k) & mask;
    orc_u128 m_n = m * (orc_u128)mt->modulus;
    uint64_t y = (uint64_t)((a - m_n) >> 64); /* overflowing_sub(..).0 >> 64 */
    if (a < m_n)
        return y + mt->modulus; /* overflowing_add */
    return y;
}

uint64_t orc_montgomery_reduce(const orc_montgomery *m, uint64_t a_hi, uint64_t a_lo)
{
    return mont_reduce(m, ((orc_u128)a_hi << 64) | a_lo);
}

/* utils.rs:181-183 */
static inline uint64_t mont_transform(const orc_montgomery *mt, uint64_t a)
{
    return mont_reduce(mt, (orc_u128)a * (orc_u128)mt->r_square);
}

/* utils.rs:185-187 */
static inline uint64_t mont_mul(const orc_montgomery *mt, uint64_t a, uint64_t b)
{
    return mont_reduce(mt, (orc_u128)a * (orc_u128)b);
}

uint64_t orc_montgomery_transform(const orc_montgomery *m, uint64_t a) { return mont_transform(m, a); }
uint64_t orc_montgomery_mul(const orc_montgomery *m, uint64_t a, uint64_t b) { return mont_mul(m, a, b); }

/* usize::ilog2 for n > 0 */
static uint32_t ilog2_u64(uint64_t n)
{
    uint32_t l = 0;
    while (n >>= 1)
        l++;
    return l;
}

/* -------------------------------------------------------------- dft/ntt.rs */

void orc_table_free(orc_table *t)
{
    if (!t)
        return;
    free(t->powers_psi_bo);
    free(t->powers_psi_inv_bo);
    free(t);
}

/* ntt.rs:239-261 */
static int table_with_precomputes(orc_table *t)
{
    uint64_t psi_montgomery = mont_transform(&t->montgomery, t->psi);
    uint64_t tmp_psi = mont_transform(&t->montgomery, 1);
    uint64_t tmp_psi_inv = mont_transform(&t->montgomery, 1);

    uint64_t n = t->n; /* capacity() == n for with_capacity(n) */
    if (n == 0)
        return panic_("ilog2 of zero");
    t->powers_psi_bo = calloc(n, sizeof(uint64_t));
    t->powers_psi_inv_bo = calloc(n, sizeof(uint64_t));
    if (!t->powers_psi_bo || !t->powers_psi_inv_bo)
        return panic_("allocation failed");

    uint32_t log_n = ilog2_u64(n);

    for (uint64_t i = 0; i < t->n; i++) {
        uint64_t reversed_index = orc_bit_reverse(i, log_n);
        if (reversed_index >= n)
            return panic_("index out of bounds in with_precomputes (odd log2 n)");
        t->powers_psi_bo[reversed_index] = tmp_psi;
        t->powers_psi_inv_bo[reversed_index] = tmp_psi_inv;

        tmp_psi = mont_mul(&t->montgomery, tmp_psi, psi_montgomery);
        tmp_psi_inv = mont_mul(&t->montgomery, tmp_psi_inv, t->psi_inv_montgomery);
    }
    return 0;
}

/* ntt.rs:90-120 */
int orc_table_from_prime_and_root(uint64_t modulus, uint64_t root, uint64_t n, orc_table **out)
{
    *out = NULL;
    orc_table *t = calloc(1, sizeof *t);
    if (!t)
        return panic_("allocation failed");
    if (orc_montgomery_from_modulus(modulus, &t->montgomery)) {
        free(t);
        return 1;
    }

    orc_i128 g[3];
    orc_egcd_i128((orc_i128)n, (orc_i128)modulus, g);
    orc_i128 n_inv_raw = g[1];
    uint64_t n_inv = (n_inv_raw < 0) ? (uint64_t)(n_inv_raw + (orc_i128)modulus) : (uint64_t)n_inv_raw;

    orc_egcd_i128((orc_i128)root, (orc_i128)modulus, g);
    orc_i128 psi_inv_raw = g[1];
    uint64_t psi_inv = (psi_inv_raw < 0) ? (uint64_t)(psi_inv_raw + (orc_i128)modulus) : (uint64_t)psi_inv_raw;

    t->q = modulus;
    t->psi = root;
    t->n_inv_montgomery = mont_transform(&t->montgomery, n_inv);
    t->psi_inv_montgomery = mont_transform(&t->montgomery, psi_inv);
    t->n = n;

    if (table_with_precomputes(t)) {
        orc_table_free(t);
        return 1;
    }
    *out = t;
    return 0;
}

/* ntt.rs:264-268 */
int orc_table_new(orc_table **out)
{
    return orc_table_from_prime_and_root(0x1fffffffffe00001ULL, 0x15eb043c7aa2b01fULL, 1ULL << 16, out);
}

/* ntt.rs:60-62 */
int orc_table_new_goldilock(orc_table **out)
{
    return orc_table_from_prime_and_root(0xffffffff00000001ULL, 0xabd0a6e8aa3d8a0eULL, 1ULL << 16, out);
}

/* ntt.rs:66-68 */
int orc_table_new_u32_compatible(orc_table **out)
{
    return orc_table_from_prime_and_root(4293918721ULL, 2004365341ULL, 1ULL << 16, out);
}

/* ntt.rs:72-75 */
int orc_table_new_simple(orc_table **out)
{
    return orc_table_from_prime_and_root(7681, 1925, 1ULL << 3, out);
}

const uint64_t *orc_table_powers_psi_bo(const orc_table *t) { return t->powers_psi_bo; }
const uint64_t *orc_table_powers_psi_inv_bo(const orc_table *t) { return t->powers_psi_inv_bo; }
uint64_t orc_table_n_inv_montgomery(const orc_table *t) { return t->n_inv_montgomery; }
uint64_t orc_table_k(const orc_table *t) { return t->montgomery.k; }
uint64_t orc_table_r_square(const orc_table *t) { return t->montgomery.r_square; }

/* ntt.rs:149-157 and ntt_goldilock.rs:104-112: U + V with the reference's compare */
static inline uint64_t add_mod_ref(uint64_t u, uint64_t v, uint64_t q)
{
    uint64_t res = u + v;
    if (res < u)            /* (res, true) */
        return res - q;     /* overflowing_sub */
    if (res > q)            /* strictly greater: res == q stays q */
        res -= q;
    return res;
}

/* ntt.rs:161-164: U - V, +q on borrow */
static inline uint64_t sub_mod_ref(uint64_t u, uint64_t v, uint64_t q)
{
    uint64_t res = u - v;
    if (u < v)
        return res + q;
    return res;
}

/* ntt.rs:122-176 */
int orc_table_forward_inplace(const orc_table *tb, uint64_t *a, size_t a_len)
{
    const orc_montgomery *mt = &tb->montgomery;

    for (size_t j = 0; j < a_len; j++)
        a[j] = mont_transform(mt, a[j]);

    size_t t = a_len;
    size_t m = 1;

    for (;;) {
        if (m >= a_len)
            break;
        t /= 2;

        for (size_t i = 0; i < m; i++) {
            size_t j_1 = 2 * i * t;
            size_t j_2 = j_1 + t - 1;

            if (m + i >= tb->n)
                return panic_("index out of bounds: powers_psi_bo");
            uint64_t cap_s = tb->powers_psi_bo[m + i];

            for (size_t j = j_1; j <= j_2; j++) {
                uint64_t cap_u = a[j];
                uint64_t cap_v = mont_mul(mt, a[j + t], cap_s);
                a[j] = add_mod_ref(cap_u, cap_v, tb->q);
                a[j + t] = sub_mod_ref(cap_u, cap_v, tb->q);
            }
        }
        m *= 2;
    }

    for (size_t j = 0; j < a_len; j++)
        a[j] = mont_reduce(mt, (orc_u128)a[j]);
    return 0;
}

/* ntt.rs:178-236 */
int orc_table_backward_inplace(const orc_table *tb, uint64_t *a, size_t a_len)
{
    const orc_montgomery *mt = &tb->montgomery;

    for (size_t j = 0; j < a_len; j++)
        a[j] = mont_transform(mt, a[j]);

    size_t t = 1;
    size_t m = a_len;

    for (;;) {
        if (m == 1)
            break;
        if (m == 0)
            return 0; /* empty slice: the reference loops forever on m = 0; callers never pass it */

        size_t j_1 = 0;
        size_t h = m / 2;

        for (size_t i = 0; i < h; i++) {
            size_t j_2 = j_1 + t - 1;
            if (h + i >= tb->n)
                return panic_("index out of bounds: powers_psi_inv_bo");
            uint64_t cap_s = tb->powers_psi_inv_bo[h + i];

            for (size_t j = j_1; j <= j_2; j++) {
                uint64_t cap_u = a[j];
                uint64_t cap_v = a[j + t];
                a[j] = add_mod_ref(cap_u, cap_v, tb->q);
                a[j + t] = mont_mul(mt, sub_mod_ref(cap_u, cap_v, tb->q), cap_s);
            }
            j_1 += 2 * t;
        }
        t *= 2;
        m /= 2;
    }

    for (size_t j = 0; j < a_len; j++)
        a[j] = mont_reduce(mt, (orc_u128)mont_mul(mt, a[j], tb->n_inv_montgomery));
    return 0;
}

/* ---------------------------------------------------- dft/ntt_goldilock.rs */

/* ntt_goldilock.rs:218-227: big-endian bytes 0..8 first, so the pair is (high, low) */
void orc_u128_to_u64(uint64_t hi, uint64_t lo, uint64_t out[2])
{
    out[0] = hi;
    out[1] = lo;
}

/* ntt_goldilock.rs:231-238: first argument lands in the high 8 bytes */
void orc_u64_to_u128(uint64_t first, uint64_t second, uint64_t out_hi_lo[2])
{
    out_hi_lo[0] = first;
    out_hi_lo[1] = second;
}

/* ntt_goldilock.rs:63-79 */
static inline uint64_t gl_reduce_159(uint64_t q, uint64_t low, uint64_t middle, uint64_t high)
{
    uint64_t low_2 = low - high;
    if (low < high)
        low_2 += q;

    uint64_t product = middle << 32;
    product -= product >> 32;

    uint64_t res = low_2 + product;
    if ((res < product) || (res >= q))
        res -= q;
    return res;
}

/* ntt_goldilock.rs:183-190 */
static inline uint64_t gl_mul_reduce(uint64_t q, uint64_t a, uint64_t b)
{
    orc_u128 x = (orc_u128)a * (orc_u128)b;
    uint64_t tmp0 = (uint64_t)(x >> 64); /* u128_to_u64(..).0 = high */
    uint64_t tmp1 = (uint64_t)x;         /* .1 = low */
    uint64_t high = tmp0 >> 32;
    uint64_t middle = tmp0 & 0xffffffffULL;
    return gl_reduce_159(q, tmp1, middle, high);
}

uint64_t orc_goldilock_reduce_159(const orc_goldilock *t, uint64_t low, uint64_t middle, uint64_t high)
{
    return gl_reduce_159(t->q, low, middle, high);
}

uint64_t orc_goldilock_mul_reduce(const orc_goldilock *t, uint64_t a, uint64_t b)
{
    return gl_mul_reduce(t->q, a, b);
}

void orc_goldilock_free(orc_goldilock *t)
{
    if (!t)
        return;
    free(t->powers_psi_bo);
    free(t->powers_psi_inv_bo);
    free(t);
}

/* ntt_goldilock.rs:192-214 */
static int goldilock_with_precomputes(orc_goldilock *t)
{
    uint64_t psi_inv = orc_mod_exp_u64(t->psi, t->q - 2, t->q);

    uint64_t tmp_psi = 1;
    uint64_t tmp_psi_inv = 1;

    if (t->n == 0)
        return panic_("ilog2 of zero");
    t->powers_psi_bo = calloc(t->n, sizeof(uint64_t));
    t->powers_psi_inv_bo = calloc(t->n, sizeof(uint64_t));
    if (!t->powers_psi_bo || !t->powers_psi_inv_bo)
        return panic_("allocation failed");

    uint32_t log_n = ilog2_u64(t->n);

    for (uint64_t i = 0; i < t->n; i++) {
        uint64_t reversed_index = orc_bit_reverse(i, log_n);
        if (reversed_index >= t->n)
            return panic_("index out of bounds in with_precomputes (odd log2 n)");
        t->powers_psi_bo[reversed_index] = tmp_psi;
        t->powers_psi_inv_bo[reversed_index] = tmp_psi_inv;

        tmp_psi = gl_mul_reduce(t->q, tmp_psi, t->psi);
        tmp_psi_inv = gl_mul_reduce(t->q, tmp_psi_inv, psi_inv);
    }
    return 0;
}

int orc_goldilock_from_root(uint64_t psi, uint64_t n, orc_goldilock **out)
{
    *out = NULL;
    orc_goldilock *t = calloc(1, sizeof *t);
    if (!t)
        return panic_("allocation failed");
    t->q = 0xffffffff00000001ULL;
    t->psi = psi;
    t->n = n;
    if (n == 0 || (n & (n - 1)) != 0 || ((t->q - 1) % n) != 0) {
        free(t);
        return panic_("n must be a power of two dividing q-1");
    }
    t->n_inv = t->q - (t->q - 1) / n;
    if (goldilock_with_precomputes(t)) {
        orc_goldilock_free(t);
        return 1;
    }
    *out = t;
    return 0;
}

/* ntt_goldilock.rs:48-58 */
int orc_goldilock_new(orc_goldilock **out)
{
    int rc = orc_goldilock_from_root(0xabd0a6e8aa3d8a0eULL, 1ULL << 16, out);
    if (rc == 0 && (*out)->n_inv != 0xfffeffff00010001ULL) {
        orc_goldilock_free(*out);
        *out = NULL;
        return panic_("n_inv differs from the constant at ntt_goldilock.rs:53");
    }
    return rc;
}

const uint64_t *orc_goldilock_powers_psi_bo(const orc_goldilock *t) { return t->powers_psi_bo; }
const uint64_t *orc_goldilock_powers_psi_inv_bo(const orc_goldilock *t) { return t->powers_psi_inv_bo; }
uint64_t orc_goldilock_n_inv(const orc_goldilock *t) { return t->n_inv; }

/* ntt_goldilock.rs:81-126 */
int orc_goldilock_forward_inplace(const orc_goldilock *tb, uint64_t *a, size_t a_len)
{
    size_t t = a_len;
    size_t m = 1;

    for (;;) {
        if (m >= a_len)
            break;
        t /= 2;

        for (size_t i = 0; i < m; i++) {
            size_t j_1 = 2 * i * t;
            size_t j_2 = j_1 + t - 1;
            if (m + i >= tb->n)
                return panic_("index out of bounds: powers_psi_bo");
            uint64_t cap_s = tb->powers_psi_bo[m + i];

            for (size_t j = j_1; j <= j_2; j++) {
                uint64_t cap_u = a[j];
                uint64_t cap_v = gl_mul_reduce(tb->q, a[j + t], cap_s);
                a[j] = add_mod_ref(cap_u, cap_v, tb->q);
                a[j + t] = sub_mod_ref(cap_u, cap_v, tb->q) % tb->q; /* :120 */
            }
        }
        m *= 2;
    }
    return 0;
}

/* ntt_goldilock.rs:128-180 */
int orc_goldilock_backward_inplace(const orc_goldilock *tb, uint64_t *a, size_t a_len)
{
    size_t t = 1;
    size_t m = a_len;

    for (;;) {
        if (m == 1)
            break;
        if (m == 0)
            return 0;

        size_t j_1 = 0;
        size_t h = m / 2;

        for (size_t i = 0; i < h; i++) {
            size_t j_2 = j_1 + t - 1;
            if (h + i >= tb->n)
                return panic_("index out of bounds: powers_psi_inv_bo");
            uint64_t cap_s = tb->powers_psi_inv_bo[h + i];

            for (size_t j = j_1; j <= j_2; j++) {
                uint64_t cap_u = a[j];
                uint64_t cap_v = a[j + t];
                a[j] = add_mod_ref(cap_u, cap_v, tb->q);
                a[j + t] = gl_mul_reduce(tb->q, sub_mod_ref(cap_u, cap_v, tb->q), cap_s);
            }
            j_1 += 2 * t;
        }
        t *= 2;
        m /= 2;
    }

    for (size_t j = 0; j < a_len; j++)
        a[j] = gl_mul_reduce(tb->q, a[j], tb->n_inv);
    return 0;
}

/* ------------------------------------------ independent closed form, O(N^2) */

static uint64_t true_bit_reverse(uint64_t x, uint32_t bits)
{
    uint64_t r = 0;
    for (uint32_t i = 0; i < bits; i++)
        r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

void orc_closed_form_forward(uint64_t q, uint64_t psi, const uint64_t *a, uint64_t *out, size_t n)
{
    uint32_t log_n = ilog2_u64(n);
    for (size_t i = 0; i < n; i++) {
        uint64_t e = 2 * true_bit_reverse(i, log_n) + 1;
        uint64_t w = orc_mod_exp_u64(psi % q, e, q); /* psi^(2 brev(i) + 1) */
        orc_u128 acc = 0;
        uint64_t p = 1;                              /* w^j */
        for (size_t j = 0; j < n; j++) {
            acc = (acc + ((orc_u128)(a[j] % q) * p) % q) % q;
            p = (uint64_t)(((orc_u128)p * w) % q);
        }
        out[i] = (uint64_t)acc;
    }
}

/* ------------------------------------------------------ CPU baseline timing */

extern int orc_avx2_forward_inplace(const orc_avx2 *t, int64_t *a, size_t len);
extern int orc_avx2_backward_inplace(const orc_avx2 *t, int64_t *a, size_t len);

typedef struct {
    int kind;
    const void *table;
    size_t n;
    int pairs;
    uint64_t seed;
    uint64_t q;
    int rc;
} bench_arg;

static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static void *bench_worker(void *p)
{
    bench_arg *b = p;
    uint64_t *a = malloc(b->n * sizeof(uint64_t));
    if (!a) {
        b->rc = 1;
        return NULL;
    }
    uint64_t s = b->seed;
    for (size_t j = 0; j < b->n; j++)
        a[j] = splitmix64(&s) % b->q;
    for (int r = 0; r < b->pairs && !b->rc; r++) {
        switch (b->kind) {
        case 0:
            b->rc |= orc_table_forward_inplace(b->table, a, b->n);
            b->rc |= orc_table_backward_inplace(b->table, a, b->n);
            break;
        case 1:
            b->rc |= orc_avx2_forward_inplace(b->table, (int64_t *)a, b->n);
            b->rc |= orc_avx2_backward_inplace(b->table, (int64_t *)a, b->n);
            break;
        default:
            b->rc |= orc_goldilock_forward_inplace(b->table, a, b->n);
            b->rc |= orc_goldilock_backward_inplace(b->table, a, b->n);
            break;
        }
    }
    free(a);
    return NULL;
}

double orc_bench_pairs(int kind, const void *table, size_t n, int pairs, int threads)
{
    if (threads < 1 || threads > 1024 || !table)
        return -1.0;
    uint64_t q = kind == 0 ? ((const orc_table *)table)->q
               : kind == 1 ? (uint64_t)((const orc_avx2 *)table)->q
                           : ((const orc_goldilock *)table)->q;
    pthread_t *th = calloc((size_t)threads, sizeof *th);
    bench_arg *args = calloc((size_t)threads, sizeof *args);
    if (!th || !args)
        return -1.0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < threads; i++) {
        args[i] = (bench_arg){kind, table, n, pairs, 0x5EED0000ULL + (uint64_t)i, q, 0};
        pthread_create(&th[i], NULL, bench_worker, &args[i]);
    }
    int rc = 0;
    for (int i = 0; i < threads; i++) {
        pthread_join(th[i], NULL);
        rc |= args[i].rc;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    free(args);
    if (rc)
        return -1.0;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
