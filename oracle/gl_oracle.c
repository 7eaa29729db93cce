/*
 * gl_oracle.c -- CPU restatement of the reference's polynomial hot path (arithmetic part).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (stark_brainfuck_amd/) never links,
 * imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against the
 * fixtures in tests/golden/ (JSON), which were produced by running the reference itself
 * (tests/golden/gen_golden.py).
 *
 * Each function cites the reference file:line it restates (paths relative to /root/reference/code).
 * Plain C, single thread, unsigned __int128 for the 64x64 product.  Build: see oracle/Makefile.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
#define GL_P 0xFFFFFFFF00000001ULL /* algebra.py:110-115  p = 2^64 - 2^32 + 1 */

/* ---- base field: algebra.py:89-99 (canonical residues in [0,p)) ---- */
uint64_t glo_add(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + b) % GL_P); }
uint64_t glo_sub(uint64_t a, uint64_t b) { return (uint64_t)(((u128)GL_P + a - b) % GL_P); }
uint64_t glo_mul(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % GL_P); }
uint64_t glo_neg(uint64_t a) { return (GL_P - a) % GL_P; }

/* algebra.py:39-46  square-and-multiply, most significant bit first */
uint64_t glo_pow(uint64_t a, uint64_t e) {
    uint64_t acc = 1;
    for (int i = 63; i >= 0; --i) {
        acc = glo_mul(acc, acc);
        if ((e >> i) & 1) acc = glo_mul(acc, a);
    }
    return acc;
}

/* algebra.py:101-103  inverse via extended Euclid on (a, p); inverse(0) = 0 there as well */
uint64_t glo_inv(uint64_t a) {
    /* signed 128-bit Bezout coefficient tracking, same recurrence as algebra.py:1-12 */
    __int128 old_r = a, r = GL_P, old_s = 1, s = 0;
    while (r != 0) {
        __int128 q = old_r / r, t;
        t = old_r - q * r; old_r = r; r = t;
        t = old_s - q * s; old_s = s; s = t;
    }
    __int128 m = old_s % (__int128)GL_P;
    if (m < 0) m += GL_P;
    return (uint64_t)m;
}

/* algebra.py:122-136  w_{2^32} = 1753635133440165772 squared down to order n = 2^log_n */
uint64_t glo_primitive_nth_root(uint32_t log_n) {
    uint64_t root = 1753635133440165772ULL;
    for (uint32_t k = 32; k > log_n; --k) root = glo_mul(root, root);
    return root;
}

/* algebra.py:138-142  big-endian bytes -> int mod p */
uint64_t glo_sample(const uint8_t* bytes, size_t len) {
    u128 acc = 0;
    for (size_t i = 0; i < len; ++i) acc = ((acc << 8) | bytes[i]) % GL_P;
    return (uint64_t)acc;
}

/* ---- ntt.py:4-23  recursive radix-2, natural order in and out ----
 * out[i] = evens[i mod half] + w^i * odds[i mod half]; w^i is accumulated instead of re-exponentiated.
 * `in` is read with stride (the reference slices values[::2] / values[1::2]). */
static void ntt_rec(uint64_t w, const uint64_t* in, size_t stride, size_t n, uint64_t* out, uint64_t* scratch) {
    if (n == 1) { out[0] = in[0]; return; }
    size_t half = n / 2;
    uint64_t* evens = scratch;
    uint64_t* odds = scratch + half;
    uint64_t w2 = glo_mul(w, w);
    ntt_rec(w2, in + stride, 2 * stride, half, odds, scratch + n);  /* ntt.py:20 */
    ntt_rec(w2, in, 2 * stride, half, evens, scratch + n);          /* ntt.py:21 */
    uint64_t wi = 1;
    for (size_t i = 0; i < n; ++i) {                                /* ntt.py:23 */
        out[i] = glo_add(evens[i % half], glo_mul(wi, odds[i % half]));
        wi = glo_mul(wi, w);
    }
}

/* returns 0 ok; 1 = not power of two (ntt.py:5-6); 2 = w^n != 1 (ntt.py:13-14); 3 = w^(n/2) == 1 (ntt.py:15-16) */
int glo_ntt(uint64_t w, const uint64_t* in, uint64_t* out, size_t n) {
    if (n & (n - 1)) return 1;
    if (n <= 1) { if (n) out[0] = in[0]; return 0; }
    if (glo_pow(w, n) != 1) return 2;
    if (glo_pow(w, n / 2) == 1) return 3;
    uint64_t* scratch = (uint64_t*)malloc(sizeof(uint64_t) * 2 * n);
    if (!scratch) return -1;
    ntt_rec(w, in, 1, n, out, scratch);
    free(scratch);
    return 0;
}

/* ntt.py:26-42  ntt with w^-1, then multiply by n^-1 */
int glo_intt(uint64_t w, const uint64_t* in, uint64_t* out, size_t n) {
    if (n & (n - 1)) return 1;
    if (glo_pow(w, n) != 1) return 2;
    if (n == 1) { out[0] = in[0]; return 0; }
    if (glo_pow(w, n / 2) == 1) return 3;
    int rc = glo_ntt(glo_inv(w), in, out, n);
    if (rc) return rc;
    uint64_t ninv = glo_inv((uint64_t)n % GL_P);
    for (size_t i = 0; i < n; ++i) out[i] = glo_mul(ninv, out[i]);
    return 0;
}

/* univariate.py:168-169  c_i <- factor^i * c_i */
void glo_scale(uint64_t factor, const uint64_t* in, uint64_t* out, size_t n) {
    uint64_t f = 1;
    for (size_t i = 0; i < n; ++i) { out[i] = glo_mul(f, in[i]); f = glo_mul(f, factor); }
}

/* ntt.py:164-168  scale by offset, zero-pad to order, ntt.  ncoef <= order */
int glo_fast_coset_evaluate(const uint64_t* coef, size_t ncoef, uint64_t offset, uint64_t generator, size_t order, uint64_t* out) {
    if (ncoef > order) return 4;
    uint64_t* tmp = (uint64_t*)calloc(order ? order : 1, sizeof(uint64_t));
    if (!tmp) return -1;
    glo_scale(offset, coef, tmp, ncoef);
    int rc = glo_ntt(generator, tmp, out, order);
    free(tmp);
    return rc;
}

/* ntt.py:171-174  intt, then scale by offset^-1 */
int glo_fast_coset_interpolate(uint64_t offset, uint64_t generator, const uint64_t* values, size_t n, uint64_t* out) {
    uint64_t* tmp = (uint64_t*)malloc(sizeof(uint64_t) * (n ? n : 1));
    if (!tmp) return -1;
    int rc = glo_intt(generator, values, tmp, n);
    if (!rc) glo_scale(glo_inv(offset), tmp, out, n);
    free(tmp);
    return rc;
}

/* ntt.py:177-188  Montgomery batch inversion; returns 5 if any input is zero (assert at :178) */
int glo_batch_inverse(const uint64_t* in, uint64_t* out, size_t n) {
    if (n == 0) return 0;
    for (size_t i = 0; i < n; ++i) if (in[i] == 0) return 5;
    out[0] = in[0];
    for (size_t i = 1; i < n; ++i) out[i] = glo_mul(out[i - 1], in[i]);
    uint64_t acc = glo_inv(out[n - 1]);
    for (size_t i = n - 1; i >= 1; --i) {
        out[i] = glo_mul(acc, out[i - 1]);
        acc = glo_mul(acc, in[i]);
    }
    out[0] = acc;
    return 0;
}

void glo_hadamard(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) { /* ntt.py:76 */
    for (size_t i = 0; i < n; ++i) out[i] = glo_mul(a[i], b[i]);
}

/* ---- extension field F_p[X]/(X^3 - X + 1): extension_field.py:65-98 ----
 * elements are 3 limbs (c0,c1,c2), low degree first.  X^3 = X - 1, X^4 = X^2 - X. */
void xo_add(const uint64_t a[3], const uint64_t b[3], uint64_t r[3]) { for (int i = 0; i < 3; ++i) r[i] = glo_add(a[i], b[i]); }
void xo_sub(const uint64_t a[3], const uint64_t b[3], uint64_t r[3]) { for (int i = 0; i < 3; ++i) r[i] = glo_sub(a[i], b[i]); }

/* extension_field.py:65-66  schoolbook product (univariate.py:40-51) reduced mod X^3 - X + 1 (univariate.py:90-109) */
void xo_mul(const uint64_t a[3], const uint64_t b[3], uint64_t r[3]) {
    uint64_t c[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c[i + j] = glo_add(c[i + j], glo_mul(a[i], b[j]));
    /* c4*X^4 = c4*(X^2 - X); c3*X^3 = c3*(X - 1) */
    uint64_t r0 = glo_sub(c[0], c[3]);
    uint64_t r1 = glo_sub(glo_add(c[1], c[3]), c[4]);
    uint64_t r2 = glo_add(c[2], c[4]);
    r[0] = r0; r[1] = r1; r[2] = r2;
}

/* extension_field.py:77-81  inverse.  The reference runs a polynomial xgcd against the modulus; the field
 * inverse is unique, so it is restated as a 3x3 linear solve: find b with a*b = 1, i.e. M(a) b = e0 where
 * column j of M(a) is a * X^j reduced.  Solved with Cramer's rule over F_p. */
void xo_inv(const uint64_t a[3], uint64_t r[3]) {
    uint64_t col[3][3];
    uint64_t x[3] = {1, 0, 0};
    for (int j = 0; j < 3; ++j) {
        xo_mul(a, x, col[j]);
        uint64_t nx[3] = {0, 0, 0}; /* x <- x * X */
        uint64_t X1[3] = {0, 1, 0};
        xo_mul(x, X1, nx);
        memcpy(x, nx, sizeof nx);
    }
    /* M[i][j] = col[j][i]; solve M b = e0:  b = adj(M)[:,0] / det(M) */
#define M(i, j) col[j][i]
    uint64_t c00 = glo_sub(glo_mul(M(1, 1), M(2, 2)), glo_mul(M(1, 2), M(2, 1)));
    uint64_t c01 = glo_sub(glo_mul(M(1, 2), M(2, 0)), glo_mul(M(1, 0), M(2, 2)));
    uint64_t c02 = glo_sub(glo_mul(M(1, 0), M(2, 1)), glo_mul(M(1, 1), M(2, 0)));
    uint64_t det = glo_add(glo_add(glo_mul(M(0, 0), c00), glo_mul(M(0, 1), c01)), glo_mul(M(0, 2), c02));
    uint64_t dinv = glo_inv(det);
    /* b_j = cofactor(0,j) / det */
    r[0] = glo_mul(c00, dinv);
    r[1] = glo_mul(c01, dinv);
    r[2] = glo_mul(c02, dinv);
#undef M
}

/* extension_field.py:30-37 */
void xo_pow(const uint64_t a[3], uint64_t e, uint64_t r[3]) {
    uint64_t acc[3] = {1, 0, 0};
    for (int i = 63; i >= 0; --i) {
        xo_mul(acc, acc, acc);
        if ((e >> i) & 1) xo_mul(acc, a, acc);
    }
    memcpy(r, acc, sizeof acc);
}

/* ntt.py:76 and ntt.py:177-188 on ExtensionFieldElement operands (the way table.py:133-134 reaches them through fast_interpolate):
 * limb-major arrays c0[n] c1[n] c2[n].  xo_batch_inverse returns 5 if any element is zero (assert at ntt.py:178). */
void xo_hadamard(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        uint64_t x[3] = {a[i], a[n + i], a[2 * n + i]}, y[3] = {b[i], b[n + i], b[2 * n + i]}, r[3];
        xo_mul(x, y, r);
        out[i] = r[0]; out[n + i] = r[1]; out[2 * n + i] = r[2];
    }
}

static void xo_get(const uint64_t* a, size_t n, size_t i, uint64_t r[3]) { r[0] = a[i]; r[1] = a[n + i]; r[2] = a[2 * n + i]; }
static void xo_put(uint64_t* a, size_t n, size_t i, const uint64_t r[3]) { a[i] = r[0]; a[n + i] = r[1]; a[2 * n + i] = r[2]; }

int xo_batch_inverse(const uint64_t* in, uint64_t* out, size_t n) {
    if (n == 0) return 0;
    uint64_t x[3], y[3], acc[3], t[3];
    for (size_t i = 0; i < n; ++i) { xo_get(in, n, i, x); if (!(x[0] | x[1] | x[2])) return 5; }
    xo_get(in, n, 0, x); xo_put(out, n, 0, x);
    for (size_t i = 1; i < n; ++i) { xo_get(out, n, i - 1, x); xo_get(in, n, i, y); xo_mul(x, y, t); xo_put(out, n, i, t); }
    xo_get(out, n, n - 1, x); xo_inv(x, acc);
    for (size_t i = n - 1; i >= 1; --i) {
        xo_get(out, n, i - 1, x); xo_mul(acc, x, t); xo_put(out, n, i, t);
        xo_get(in, n, i, y); xo_mul(acc, y, t); memcpy(acc, t, sizeof t);
    }
    xo_put(out, n, 0, acc);
    return 0;
}

/* ---- fri.py:127-128  one split-and-fold round over an SoA codeword (limb-major: c0[n], c1[n], c2[n]) ----
 * out[i] = 2^-1 * ((1 + alpha/(offset*omega^i)) * cw[i] + (1 - alpha/(offset*omega^i)) * cw[n/2+i]),  i < n/2.
 * offset and omega are base-field elements lifted into the extension (fri.py:94-95). */
void xo_fri_fold(const uint64_t* in, size_t n, const uint64_t alpha[3], uint64_t offset, uint64_t omega, uint64_t* out) {
    size_t h = n / 2;
    uint64_t two_inv[3] = {glo_inv(2), 0, 0};
    uint64_t one[3] = {1, 0, 0};
    uint64_t wi = 1;
    for (size_t i = 0; i < h; ++i) {
        uint64_t x[3] = {glo_mul(offset, wi), 0, 0}, xinv[3], q[3], lp[3], lm[3], a[3], b[3], t0[3], t1[3], s[3], r[3];
        xo_inv(x, xinv);
        xo_mul(alpha, xinv, q);          /* alpha / (offset * omega^i) */
        xo_add(one, q, lp);
        xo_sub(one, q, lm);
        for (int k = 0; k < 3; ++k) { a[k] = in[k * n + i]; b[k] = in[k * n + h + i]; }
        xo_mul(lp, a, t0);
        xo_mul(lm, b, t1);
        xo_add(t0, t1, s);
        xo_mul(two_inv, s, r);
        for (int k = 0; k < 3; ++k) out[k * h + i] = r[k];
        wi = glo_mul(wi, omega);
    }
}

/* SURVEY 8d input recipe: felt(seed, i) = splitmix64(seed + i) mod p */
uint64_t glo_felt(uint64_t seed, uint64_t i) {
    uint64_t x = seed + i + 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return z % GL_P;
}
void glo_felt_fill(uint64_t seed, uint64_t start, uint64_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = glo_felt(seed, start + i);
}
