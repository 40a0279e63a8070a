/*
 * CPU oracle, AVX2 path -- TEST INFRASTRUCTURE ONLY (see ntt_oracle.h).
 *
 * Restates src/dft/ntt_avx2.rs of the reference with the same x86 intrinsics
 * (compile with -mavx2).  All 128-bit arithmetic is emulated from
 * _mm256_mul_epu32 exactly as the reference does, and every load and store is
 * the reference's per-lane set/extract, so this is also a fair CPU baseline for
 * that path.  The scalar fallbacks at ntt_avx2.rs:473-514 (reached only for
 * len <= 8, and wrong there) are restated as they stand.
 */
#include "ntt_oracle.h"

#include <immintrin.h>
#include <stdlib.h>

/* mirrors of the scalar helpers the AVX2 table borrows from utils.rs */
static inline uint64_t mont_reduce_(const orc_montgomery *mt, orc_u128 a)
{
    const orc_u128 mask = (orc_u128)0xffffffffffffffffULL;
    orc_u128 m = ((a & mask) * (orc_u128)mt->k) & mask;
    orc_u128 m_n = m * (orc_u128)mt->modulus;
    uint64_t y = (uint64_t)((a - m_n) >> 64);
    return (a < m_n) ? y + mt->modulus : y;
}
static inline int64_t mont_transform_(const orc_montgomery *mt, int64_t a)
{
    /* impl_montgomery_transformer!(i64): `a as u128` sign-extends, values here are >= 0 */
    return (int64_t)mont_reduce_(mt, (orc_u128)(orc_i128)a * (orc_u128)(orc_i128)(int64_t)mt->r_square);
}
static inline int64_t mont_mul_(const orc_montgomery *mt, int64_t a, int64_t b)
{
    return (int64_t)mont_reduce_(mt, (orc_u128)(orc_i128)a * (orc_u128)(orc_i128)b);
}

static inline __m256i ld4(const int64_t *p) { return _mm256_setr_epi64x(p[0], p[1], p[2], p[3]); }
static inline void st4(int64_t *p, __m256i v)
{
    p[0] = _mm256_extract_epi64(v, 0);
    p[1] = _mm256_extract_epi64(v, 1);
    p[2] = _mm256_extract_epi64(v, 2);
    p[3] = _mm256_extract_epi64(v, 3);
}

/* ntt_avx2.rs:728-744 */
static inline __m256i a_lt_b(__m256i a, __m256i b)
{
    __m256i zero = _mm256_setzero_si256();
    __m256i a_lt_0 = _mm256_cmpgt_epi64(zero, a);
    __m256i b_lt_0 = _mm256_cmpgt_epi64(zero, b);

    __m256i a_lt_b_s = _mm256_cmpgt_epi64(b, a);
    __m256i b_lt_a_s = _mm256_cmpgt_epi64(b, a); /* :733 computes the same compare twice */

    __m256i first = _mm256_andnot_si256(_mm256_or_si256(a_lt_0, b_lt_0), a_lt_b_s);
    __m256i second = _mm256_and_si256(_mm256_andnot_si256(a_lt_0, b_lt_0), _mm256_set1_epi64x(-1));
    __m256i third = _mm256_and_si256(_mm256_andnot_si256(b_lt_0, a_lt_0), zero);
    __m256i fourth = _mm256_and_si256(_mm256_and_si256(a_lt_0, b_lt_0), b_lt_a_s);

    return _mm256_or_si256(_mm256_or_si256(first, second), _mm256_or_si256(third, fourth));
}

/* ntt_avx2.rs:747-779: schoolbook 64x64 -> (high, low) from four 32x32 products */
static inline void a_mul_b_karatsuba(__m256i a, __m256i b, __m256i *high_out, __m256i *low_out)
{
    const __m256i lo32 = _mm256_set1_epi64x(0xffffffffLL);
    __m256i a_low = _mm256_and_si256(lo32, a);
    __m256i a_high = _mm256_srli_epi64(a, 32);
    __m256i b_low = _mm256_and_si256(lo32, b);
    __m256i b_high = _mm256_srli_epi64(b, 32);

    __m256i high = _mm256_mul_epu32(a_high, b_high);
    __m256i low = _mm256_mul_epu32(a_low, b_low);
    __m256i low_low = _mm256_and_si256(lo32, low);
    __m256i low_high = _mm256_srli_epi64(low, 32);

    __m256i middle_0 = _mm256_mul_epu32(a_high, b_low);
    __m256i middle_1 = _mm256_mul_epu32(a_low, b_high);

    __m256i middle_0_low = _mm256_and_si256(lo32, middle_0);
    __m256i middle_0_high = _mm256_srli_epi64(middle_0, 32);
    __m256i middle_1_low = _mm256_and_si256(lo32, middle_1);
    __m256i middle_1_high = _mm256_srli_epi64(middle_1, 32);

    high = _mm256_add_epi64(high, middle_0_high);
    high = _mm256_add_epi64(high, middle_1_high);

    low_high = _mm256_add_epi64(low_high, middle_0_low);
    low_high = _mm256_add_epi64(low_high, middle_1_low);

    high = _mm256_add_epi64(high, _mm256_srli_epi64(low_high, 32));

    *high_out = high;
    *low_out = _mm256_or_si256(low_low, _mm256_slli_epi64(low_high, 32));
}

/* ntt_avx2.rs:782-796 */
static inline void a_sub_b(__m256i a_high, __m256i a_low, __m256i b_high, __m256i b_low,
                           __m256i *high_out, __m256i *low_out)
{
    __m256i low = _mm256_sub_epi64(a_low, b_low);
    __m256i low_carry = _mm256_and_si256(_mm256_set1_epi64x(1), a_lt_b(a_low, b_low));
    __m256i high = _mm256_sub_epi64(_mm256_sub_epi64(a_high, low_carry), b_high);
    *high_out = high;
    *low_out = low;
}

/* ntt_avx2.rs:799-813 */
static inline __m256i a_lt_b_wide(__m256i a_high, __m256i a_low, __m256i b_high, __m256i b_low)
{
    __m256i a_low_lt_b_low = a_lt_b(a_low, b_low);
    __m256i a_high_lt_b_high = a_lt_b(a_high, b_high);
    __m256i a_high_eq_b_high = _mm256_cmpeq_epi64(a_high, b_high);
    return _mm256_or_si256(a_high_lt_b_high, _mm256_and_si256(a_high_eq_b_high, a_low_lt_b_low));
}

/* ntt_avx2.rs:623-644 */
static inline __m256i montgomery_reduce_64(const orc_avx2 *t, __m256i t_high, __m256i t_low)
{
    __m256i unused, m_low;
    a_mul_b_karatsuba(t_low, _mm256_set1_epi64x((int64_t)t->montgomery.k), &unused, &m_low);

    __m256i m_n_high, m_n_low;
    a_mul_b_karatsuba(m_low, _mm256_set1_epi64x(t->q), &m_n_high, &m_n_low);

    __m256i y, y_low;
    a_sub_b(t_high, t_low, m_n_high, m_n_low, &y, &y_low);

    __m256i mask = _mm256_and_si256(a_lt_b_wide(t_high, t_low, m_n_high, m_n_low),
                                    _mm256_set1_epi64x(t->q));
    return _mm256_add_epi64(mask, y);
}

/* ntt_avx2.rs:554-559 */
static inline __m256i montgomery_mul_vec(const orc_avx2 *t, __m256i a, __m256i b)
{
    __m256i high, low;
    a_mul_b_karatsuba(a, b, &high, &low);
    return montgomery_reduce_64(t, high, low);
}

/* ntt_avx2.rs:563-577 */
static inline __m256i reduce_if_greater_equal_q(const orc_avx2 *t, __m256i a)
{
    __m256i q = _mm256_set1_epi64x(t->q);
    __m256i masked = _mm256_and_si256(_mm256_cmpgt_epi64(a, q), q);
    a = _mm256_sub_epi64(a, masked);
    masked = _mm256_and_si256(_mm256_cmpeq_epi64(a, q), q);
    return _mm256_sub_epi64(a, masked);
}

/* ntt_avx2.rs:581-589 */
static inline __m256i reduce_if_negative(const orc_avx2 *t, __m256i a)
{
    __m256i masked = _mm256_and_si256(_mm256_set1_epi64x(t->q),
                                      _mm256_cmpgt_epi64(_mm256_setzero_si256(), a));
    return _mm256_add_epi64(a, masked);
}

/* ntt_avx2.rs:518-533 */
static inline void cs_butterfly_simd_4(const orc_avx2 *t, __m256i cap_s, __m256i *a_j, __m256i *a_j_t)
{
    __m256i cap_u = *a_j;
    __m256i cap_v = montgomery_mul_vec(t, *a_j_t, cap_s);
    *a_j = reduce_if_greater_equal_q(t, _mm256_add_epi64(cap_u, cap_v));
    *a_j_t = reduce_if_negative(t, _mm256_sub_epi64(cap_u, cap_v));
}

/* ntt_avx2.rs:537-550 */
static inline void gs_butterfly_simd_4(const orc_avx2 *t, __m256i cap_s, __m256i *a_j, __m256i *a_j_t)
{
    __m256i cap_u = *a_j;
    __m256i cap_v = *a_j_t;
    *a_j = reduce_if_greater_equal_q(t, _mm256_add_epi64(cap_u, cap_v));
    __m256i d = reduce_if_negative(t, _mm256_sub_epi64(cap_u, cap_v));
    *a_j_t = montgomery_mul_vec(t, d, cap_s);
}

/* ntt_avx2.rs:617-619 */
static inline int64_t mul_reduce(const orc_avx2 *t, int64_t a, int64_t b)
{
    return (int64_t)(((orc_u128)(orc_i128)a * (orc_u128)(orc_i128)b) % (orc_u128)(orc_i128)t->q);
}

/* ntt_avx2.rs:473-493 */
static inline void cs_butterfly(const orc_avx2 *t, int64_t cap_s, int64_t *a_j, int64_t *a_j_t)
{
    int64_t cap_u = *a_j;
    int64_t cap_v = mul_reduce(t, *a_j_t, cap_s);
    int64_t s = cap_u + cap_v;
    if (s >= t->q)
        s -= t->q;
    int64_t d = cap_u - cap_v;
    if (d < 0)
        d += t->q;
    *a_j = s;
    *a_j_t = d;
}

/* ntt_avx2.rs:495-514 */
static inline void gs_butterfly(const orc_avx2 *t, int64_t cap_s, int64_t *a_j, int64_t *a_j_t)
{
    int64_t cap_u = *a_j;
    int64_t cap_v = *a_j_t;
    int64_t s = cap_u + cap_v;
    if (s >= t->q)
        s -= t->q;
    int64_t d = cap_u - cap_v;
    if (d < 0)
        d += t->q;
    *a_j = s;
    *a_j_t = mul_reduce(t, d, cap_s);
}

/* ntt_avx2.rs:674-678 */
static inline __m256i to_montgomery(const orc_avx2 *t, __m256i a)
{
    __m256i high, low;
    a_mul_b_karatsuba(a, _mm256_set1_epi64x((int64_t)t->montgomery.r_square), &high, &low);
    return montgomery_reduce_64(t, high, low);
}

/* ntt_avx2.rs:682-701; steps by 4 with no tail handling */
static void to_montgomery_vec(const orc_avx2 *t, int64_t *a, size_t a_len)
{
    for (size_t j = 0; j < a_len; j += 4)
        st4(a + j, to_montgomery(t, ld4(a + j)));
}

/* ntt_avx2.rs:648-670 */
static void montgomery_reduce_vec(const orc_avx2 *t, int64_t *a, size_t a_len)
{
    for (size_t j = 0; j < a_len; j += 4)
        st4(a + j, montgomery_reduce_64(t, _mm256_setzero_si256(), ld4(a + j)));
}

/* The three per-stage shapes shared by forward (ntt_avx2.rs:123-271) and
 * backward (:297-443).  `tw` points at powers[m] (forward) or powers_inv[h]
 * (backward), `blocks` is m or h, CT selects the butterfly. */
static void stage_t1(const orc_avx2 *tb, int64_t *a, const int64_t *tw, size_t blocks, int ct)
{
    for (size_t i = 0; i < blocks; i += 4) {
        __m256i a_j = _mm256_setr_epi64x(a[2 * i], a[2 * (i + 1)], a[2 * (i + 2)], a[2 * (i + 3)]);
        __m256i a_j_t = _mm256_setr_epi64x(a[2 * i + 1], a[2 * (i + 1) + 1], a[2 * (i + 2) + 1], a[2 * (i + 3) + 1]);
        __m256i s = _mm256_setr_epi64x(tw[i], tw[i + 1], tw[i + 2], tw[i + 3]);
        if (ct)
            cs_butterfly_simd_4(tb, s, &a_j, &a_j_t);
        else
            gs_butterfly_simd_4(tb, s, &a_j, &a_j_t);
        a[2 * i] = _mm256_extract_epi64(a_j, 0);
        a[2 * (i + 1)] = _mm256_extract_epi64(a_j, 1);
        a[2 * (i + 2)] = _mm256_extract_epi64(a_j, 2);
        a[2 * (i + 3)] = _mm256_extract_epi64(a_j, 3);
        a[2 * i + 1] = _mm256_extract_epi64(a_j_t, 0);
        a[2 * (i + 1) + 1] = _mm256_extract_epi64(a_j_t, 1);
        a[2 * (i + 2) + 1] = _mm256_extract_epi64(a_j_t, 2);
        a[2 * (i + 3) + 1] = _mm256_extract_epi64(a_j_t, 3);
    }
}

static void stage_t2(const orc_avx2 *tb, int64_t *a, const int64_t *tw, size_t blocks, int ct)
{
    for (size_t i = 0; i < blocks; i += 2) {
        size_t j_i1 = 2 * i * 2, j_i1_t = j_i1 + 2;
        size_t j_i2 = 2 * (i + 1) * 2, j_i2_t = j_i2 + 2;
        __m256i a_j = _mm256_setr_epi64x(a[j_i1], a[j_i1 + 1], a[j_i2], a[j_i2 + 1]);
        __m256i a_j_t = _mm256_setr_epi64x(a[j_i1_t], a[j_i1_t + 1], a[j_i2_t], a[j_i2_t + 1]);
        __m256i s = _mm256_setr_epi64x(tw[i], tw[i], tw[i + 1], tw[i + 1]);
        if (ct)
            cs_butterfly_simd_4(tb, s, &a_j, &a_j_t);
        else
            gs_butterfly_simd_4(tb, s, &a_j, &a_j_t);
        a[j_i1] = _mm256_extract_epi64(a_j, 0);
        a[j_i1 + 1] = _mm256_extract_epi64(a_j, 1);
        a[j_i2] = _mm256_extract_epi64(a_j, 2);
        a[j_i2 + 1] = _mm256_extract_epi64(a_j, 3);
        a[j_i1_t] = _mm256_extract_epi64(a_j_t, 0);
        a[j_i1_t + 1] = _mm256_extract_epi64(a_j_t, 1);
        a[j_i2_t] = _mm256_extract_epi64(a_j_t, 2);
        a[j_i2_t + 1] = _mm256_extract_epi64(a_j_t, 3);
    }
}

/* One block [j_1, j_1 + t) of the generic shape: groups of four while t >= 4,
 * the scalar butterfly otherwise.  In the reference the `for mut j` loop is left
 * by `break` after the first pass of its inner loop whenever t >= 4, so each
 * element is visited once. */
static void block_generic(const orc_avx2 *tb, int64_t *a, size_t j_1, size_t t, int64_t cap_s, int ct)
{
    size_t j_2 = j_1 + t - 1;
    for (size_t j0 = j_1; j0 <= j_2; j0++) {
        size_t j = j0;
        while (!(t < 4 || j >= j_2)) {
            __m256i a_j = ld4(a + j);
            __m256i a_j_t = ld4(a + j + t);
            if (ct)
                cs_butterfly_simd_4(tb, _mm256_set1_epi64x(cap_s), &a_j, &a_j_t);
            else
                gs_butterfly_simd_4(tb, _mm256_set1_epi64x(cap_s), &a_j, &a_j_t);
            st4(a + j, a_j);
            st4(a + j + t, a_j_t);
            j += 4;
        }
        if (j > j_2)
            break;
        if (ct)
            cs_butterfly(tb, cap_s, &a[j], &a[j + t]);
        else
            gs_butterfly(tb, cap_s, &a[j], &a[j + t]);
    }
}

/* ntt_avx2.rs:108-279 */
int orc_avx2_forward_inplace(const orc_avx2 *tb, int64_t *a, size_t a_len)
{
    if (a_len % 4 != 0 || a_len > tb->n)
        return 1; /* the reference reads past the slice / table here */
    size_t t = a_len;
    size_t m = 1;

    to_montgomery_vec(tb, a, a_len);

    while (m < a_len) {
        t /= 2;
        if (t == 1 && m >= 4) {
            stage_t1(tb, a, tb->powers_psi_bo + m, m, 1);
        } else if (t == 2 && m >= 4) {
            stage_t2(tb, a, tb->powers_psi_bo + m, m, 1);
        } else {
            for (size_t i = 0; i < m; i++)
                block_generic(tb, a, 2 * i * t, t, tb->powers_psi_bo[m + i], 1);
        }
        m *= 2;
    }

    montgomery_reduce_vec(tb, a, a_len);
    return 0;
}

/* ntt_avx2.rs:281-470 */
int orc_avx2_backward_inplace(const orc_avx2 *tb, int64_t *a, size_t a_len)
{
    if (a_len % 4 != 0 || a_len > tb->n || a_len == 0)
        return 1;
    size_t t = 1;
    size_t m = a_len;

    to_montgomery_vec(tb, a, a_len);

    while (m != 1) {
        size_t h = m / 2;
        if (t == 1 && h >= 4) {
            stage_t1(tb, a, tb->powers_psi_inv_bo + h, h, 0);
        } else if (t == 2 && h >= 4) {
            stage_t2(tb, a, tb->powers_psi_inv_bo + h, h, 0);
        } else {
            size_t j_1 = 0;
            for (size_t i = 0; i < h; i++) {
                block_generic(tb, a, j_1, t, tb->powers_psi_inv_bo[h + i], 0);
                j_1 += 2 * t;
            }
        }
        t *= 2;
        m /= 2;
    }

    /* ntt_avx2.rs:450-469 */
    for (size_t j = 0; j < a_len; j += 4) {
        __m256i a_j = ld4(a + j);
        __m256i res = montgomery_reduce_64(
            tb, _mm256_setzero_si256(),
            montgomery_mul_vec(tb, a_j, _mm256_set1_epi64x(tb->n_inv_montgomery)));
        st4(a + j, res);
    }
    return 0;
}

void orc_avx2_free(orc_avx2 *t)
{
    if (!t)
        return;
    free(t->powers_psi_bo);
    free(t->powers_psi_inv_bo);
    free(t);
}

/* ntt_avx2.rs:592-613 */
static int avx2_with_precomputes(orc_avx2 *t)
{
    int64_t psi_montgomery = mont_transform_(&t->montgomery, t->psi);
    int64_t tmp_psi = mont_transform_(&t->montgomery, 1);
    int64_t tmp_psi_inv = mont_transform_(&t->montgomery, 1);

    if (t->n == 0)
        return 1;
    t->powers_psi_bo = calloc(t->n, sizeof(int64_t));
    t->powers_psi_inv_bo = calloc(t->n, sizeof(int64_t));
    if (!t->powers_psi_bo || !t->powers_psi_inv_bo)
        return 1;

    uint32_t log_n = 0;
    for (size_t x = t->n; x >>= 1;)
        log_n++;

    for (size_t i = 0; i < t->n; i++) {
        size_t reversed_index = (size_t)orc_bit_reverse((uint64_t)i, log_n);
        if (reversed_index >= t->n)
            return 1; /* index out of bounds (odd log2 n) */
        t->powers_psi_bo[reversed_index] = tmp_psi;
        t->powers_psi_inv_bo[reversed_index] = tmp_psi_inv;
        tmp_psi = mont_mul_(&t->montgomery, tmp_psi, psi_montgomery);
        tmp_psi_inv = mont_mul_(&t->montgomery, tmp_psi_inv, t->psi_inv_montgomery);
    }
    return 0;
}

/* ntt_avx2.rs:74-105 */
int orc_avx2_from_prime_and_root(int64_t modulus, int64_t root, int64_t n, orc_avx2 **out)
{
    *out = NULL;
    orc_avx2 *t = calloc(1, sizeof *t);
    if (!t)
        return 1;
    if (orc_montgomery_from_modulus((uint64_t)modulus, &t->montgomery)) {
        free(t);
        return 1;
    }

    orc_i128 g[3];
    orc_egcd_i128((orc_i128)n, (orc_i128)modulus, g);
    orc_i128 n_inv_raw = g[1];
    int64_t n_inv = (n_inv_raw < 0) ? (int64_t)n_inv_raw + modulus : (int64_t)n_inv_raw;

    orc_egcd_i128((orc_i128)root, (orc_i128)modulus, g);
    orc_i128 psi_inv_raw = g[1];
    int64_t psi_inv = (psi_inv_raw < 0) ? (int64_t)(psi_inv_raw + (orc_i128)modulus) : (int64_t)psi_inv_raw;

    t->q = modulus;
    t->psi = root;
    t->psi_inv_montgomery = mont_transform_(&t->montgomery, psi_inv);
    t->n_inv_montgomery = mont_transform_(&t->montgomery, n_inv);
    t->n = (size_t)n;

    if (avx2_with_precomputes(t)) {
        orc_avx2_free(t);
        return 1;
    }
    *out = t;
    return 0;
}

/* ntt_avx2.rs:57-59 */
int orc_avx2_new(orc_avx2 **out)
{
    return orc_avx2_from_prime_and_root(0x1fffffffffe00001LL, 0x15eb043c7aa2b01fLL, 1LL << 16, out);
}

const int64_t *orc_avx2_powers_psi_bo(const orc_avx2 *t) { return t->powers_psi_bo; }
const int64_t *orc_avx2_powers_psi_inv_bo(const orc_avx2 *t) { return t->powers_psi_inv_bo; }

/* ---- array-of-4 entry points for the known-answer tests ------------------- */
void orc_avx2_a_mul_b_karatsuba(const int64_t a[4], const int64_t b[4], int64_t high[4], int64_t low[4])
{
    __m256i h, l;
    a_mul_b_karatsuba(ld4(a), ld4(b), &h, &l);
    st4(high, h);
    st4(low, l);
}
void orc_avx2_a_lt_b(const int64_t a[4], const int64_t b[4], int64_t out[4]) { st4(out, a_lt_b(ld4(a), ld4(b))); }
void orc_avx2_a_sub_b(const int64_t ah[4], const int64_t al[4], const int64_t bh[4], const int64_t bl[4],
                      int64_t high[4], int64_t low[4])
{
    __m256i h, l;
    a_sub_b(ld4(ah), ld4(al), ld4(bh), ld4(bl), &h, &l);
    st4(high, h);
    st4(low, l);
}
void orc_avx2_a_lt_b_wide(const int64_t ah[4], const int64_t al[4], const int64_t bh[4], const int64_t bl[4],
                          int64_t out[4])
{
    st4(out, a_lt_b_wide(ld4(ah), ld4(al), ld4(bh), ld4(bl)));
}
void orc_avx2_montgomery_reduce_64(const orc_avx2 *t, const int64_t th[4], const int64_t tl[4], int64_t out[4])
{
    st4(out, montgomery_reduce_64(t, ld4(th), ld4(tl)));
}
void orc_avx2_montgomery_mul_vec(const orc_avx2 *t, const int64_t a[4], const int64_t b[4], int64_t out[4])
{
    st4(out, montgomery_mul_vec(t, ld4(a), ld4(b)));
}
void orc_avx2_to_montgomery(const orc_avx2 *t, const int64_t a[4], int64_t out[4])
{
    st4(out, to_montgomery(t, ld4(a)));
}
void orc_avx2_reduce_if_greater_equal_q(const orc_avx2 *t, int64_t a[4])
{
    st4(a, reduce_if_greater_equal_q(t, ld4(a)));
}
void orc_avx2_reduce_if_negative(const orc_avx2 *t, int64_t a[4]) { st4(a, reduce_if_negative(t, ld4(a))); }
void orc_avx2_cs_butterfly_simd_4(const orc_avx2 *t, const int64_t s[4], int64_t a_j[4], int64_t a_j_t[4])
{
    __m256i x = ld4(a_j), y = ld4(a_j_t);
    cs_butterfly_simd_4(t, ld4(s), &x, &y);
    st4(a_j, x);
    st4(a_j_t, y);
}
void orc_avx2_gs_butterfly_simd_4(const orc_avx2 *t, const int64_t s[4], int64_t a_j[4], int64_t a_j_t[4])
{
    __m256i x = ld4(a_j), y = ld4(a_j_t);
    gs_butterfly_simd_4(t, ld4(s), &x, &y);
    st4(a_j, x);
    st4(a_j_t, y);
}
