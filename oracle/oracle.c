/*
 * oracle.c - CPU restatement of the RLWE ciphertext-arithmetic hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED BY THE REFERENCE.  deeppowers/deeppowers holds no FHE code to restate (SURVEY.md
 * section 0: no NTT / modular arithmetic / Ciphertext / Evaluator anywhere under /root/reference;
 * FHE is README prose only, /root/reference/README.md:42,93,197).  What this file follows instead:
 *
 *   (1) orc_schoolbook_negacyclic - the mathematical definition c = a*b in Z_q[X]/(X^N+1) with
 *       canonical residues (unique answer; 128-bit accumulation).  This is the ground truth.
 *   (2) the "build CPU evaluator" of BASELINE.md section 3: radix-2 Harvey lazy-reduction
 *       negacyclic NTT / inverse NTT with Shoup twiddles (published algorithm: D. Harvey, "Faster
 *       arithmetic for number-theoretic transforms", J. Symb. Comp. 2014; ordering convention of
 *       SURVEY.md section 8(a) A1/A2), Barrett dyadic multiply, and the ct x ct tensor product
 *       built from them.  It doubles as the timed CPU baseline (bench.py cpu_baseline, kind "port").
 *
 * It is pinned against oracle/pyoracle.py (Python big ints) and the known answers of SURVEY.md
 * Appendix B in tests/test_oracle_*.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from this file; the product path never does.
 *
 * Layout everywhere: [batch][component][limb][N] little-endian u64 words, canonical residues.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

typedef struct {
    uint32_t log2n;
    uint64_t n, q, psi;
    uint64_t *rp, *rp_sh;   /* psi^brv(i),            floor(rp[i]  * 2^64 / q) */
    uint64_t *irp, *irp_sh; /* psi^-brv(i),           Shoup companions         */
    uint64_t ninv, ninv_sh; /* N^-1 mod q                                      */
    uint64_t br_hi, br_lo;  /* floor(2^128 / q) as two words (Barrett)         */
} orc_limb;

typedef struct {
    uint32_t log2n, n_limbs;
    orc_limb* limb;
} orc_ctx;

/* ------------------------------------------------------------------ scalar modular arithmetic */
static inline uint64_t mulmod_slow(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((u128)a * b % q); }

static uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) {
    uint64_t r = 1;
    b %= q;
    while (e) {
        if (e & 1) r = mulmod_slow(r, b, q);
        b = mulmod_slow(b, b, q);
        e >>= 1;
    }
    return r;
}

static inline uint64_t shoup_of(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }

/* w*y mod q in [0, 2q) for any y < 2^64 (Harvey); w < q, wsh = floor(w*2^64/q) */
static inline uint64_t mul_shoup_lazy(uint64_t y, uint64_t w, uint64_t wsh, uint64_t q) {
    uint64_t hi = (uint64_t)(((u128)y * wsh) >> 64);
    return y * w - hi * q;
}

/* Barrett reduction of a 128-bit value with the two-word ratio floor(2^128/q); out in [0,q) */
static inline uint64_t barrett128(u128 z, const orc_limb* t) {
    uint64_t z0 = (uint64_t)z, z1 = (uint64_t)(z >> 64);
    /* floor(z * ratio / 2^128), dropping only the z0*br_lo low word (error <= 2 quotients) */
    uint64_t carry = (uint64_t)(((u128)z0 * t->br_lo) >> 64);
    u128 t1 = (u128)z0 * t->br_hi + carry;
    u128 t2 = (u128)z1 * t->br_lo + (uint64_t)t1;
    uint64_t qhat = z1 * t->br_hi + (uint64_t)(t1 >> 64) + (uint64_t)(t2 >> 64);
    uint64_t r = z0 - qhat * t->q;
    while (r >= t->q) r -= t->q;
    return r;
}

static inline uint64_t mulmod_barrett(uint64_t a, uint64_t b, const orc_limb* t) { return barrett128((u128)a * b, t); }

static uint32_t brv(uint32_t x, uint32_t bits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

/* ------------------------------------------------------------------ context */
static int limb_init(orc_limb* t, uint32_t log2n, uint64_t q, uint64_t psi) {
    uint64_t n = 1ull << log2n;
    memset(t, 0, sizeof *t);
    if (q < 3 || (q >> 62) || (q - 1) % (2 * n) != 0) return 2000;
    if (psi == 0 || psi >= q || powmod(psi, n, q) != q - 1) return 2000;
    t->log2n = log2n; t->n = n; t->q = q; t->psi = psi;
    t->rp = malloc(4 * n * sizeof(uint64_t));
    if (!t->rp) return 1001;
    t->rp_sh = t->rp + n; t->irp = t->rp + 2 * n; t->irp_sh = t->rp + 3 * n;
    uint64_t ipsi = powmod(psi, q - 2, q);
    uint64_t pw = 1, ipw = 1;
    for (uint64_t i = 0; i < n; ++i) { /* natural powers scattered to bit-reversed slots */
        uint32_t r = brv((uint32_t)i, log2n);
        t->rp[r] = pw;   t->rp_sh[r] = shoup_of(pw, q);
        t->irp[r] = ipw; t->irp_sh[r] = shoup_of(ipw, q);
        pw = mulmod_slow(pw, psi, q); ipw = mulmod_slow(ipw, ipsi, q);
    }
    t->ninv = powmod(n % q, q - 2, q);
    t->ninv_sh = shoup_of(t->ninv, q);
    u128 ratio = (~(u128)0) / q; /* floor((2^128-1)/q) == floor(2^128/q) since q is not a power of two */
    t->br_hi = (uint64_t)(ratio >> 64); t->br_lo = (uint64_t)ratio;
    return 0;
}

int orc_ctx_create(orc_ctx** out, uint32_t log2n, uint32_t n_limbs, const uint64_t* moduli, const uint64_t* psi) {
    if (!out || !moduli || !psi || n_limbs == 0 || log2n < 1 || log2n > 20) return 2000;
    orc_ctx* c = calloc(1, sizeof *c);
    if (!c) return 1001;
    c->log2n = log2n; c->n_limbs = n_limbs;
    c->limb = calloc(n_limbs, sizeof(orc_limb));
    if (!c->limb) { free(c); return 1001; }
    for (uint32_t l = 0; l < n_limbs; ++l) {
        int rc = limb_init(&c->limb[l], log2n, moduli[l], psi[l]);
        if (rc) {
            for (uint32_t k = 0; k <= l; ++k) free(c->limb[k].rp);
            free(c->limb); free(c);
            return rc;
        }
    }
    *out = c;
    return 0;
}

void orc_ctx_destroy(orc_ctx* c) {
    if (!c) return;
    for (uint32_t l = 0; l < c->n_limbs; ++l) free(c->limb[l].rp);
    free(c->limb); free(c);
}

/* copies rp (forward table, psi^brv(i)) of one limb out - lets tests pin the table itself */
void orc_get_root_powers(const orc_ctx* c, uint32_t limb, uint64_t* out_rp, uint64_t* out_irp) {
    const orc_limb* t = &c->limb[limb];
    if (out_rp) memcpy(out_rp, t->rp, t->n * sizeof(uint64_t));
    if (out_irp) memcpy(out_irp, t->irp, t->n * sizeof(uint64_t));
}

/* ------------------------------------------------------------------ ground truth */
void orc_schoolbook_negacyclic(uint64_t* out, const uint64_t* a, const uint64_t* b, uint64_t n, uint64_t q) {
    /* positive (i+j = k) and negative (i+j = k+N, X^N = -1) parts accumulated separately in 128 bits;
     * `lim` raw products (each < q^2) fit a u128 next to an already reduced value, then we fold mod q. */
    int bits = 64 - __builtin_clzll(q);
    int sh = 127 - 2 * bits;
    uint64_t lim = sh <= 0 ? 1 : (sh > 20 ? (1ull << 20) : (1ull << sh));
    for (uint64_t k = 0; k < n; ++k) {
        u128 pos = 0, neg = 0;
        uint64_t cnt = 0;
        for (uint64_t i = 0; i <= k; ++i) {
            pos += (u128)a[i] * b[k - i];
            if (++cnt == lim) { pos %= q; cnt = 0; }
        }
        cnt = 0;
        for (uint64_t i = k + 1; i < n; ++i) {
            neg += (u128)a[i] * b[n + k - i];
            if (++cnt == lim) { neg %= q; cnt = 0; }
        }
        uint64_t p = (uint64_t)(pos % q), m = (uint64_t)(neg % q);
        out[k] = p >= m ? p - m : p + q - m;
    }
}

/* ------------------------------------------------------------------ Harvey NTT (one residue polynomial) */
static void ntt_fwd_poly(const orc_limb* T, uint64_t* a) {
    const uint64_t q = T->q, two_q = 2 * q, n = T->n;
    uint64_t t = n;
    for (uint64_t m = 1; m < n; m <<= 1) {
        t >>= 1;
        for (uint64_t i = 0; i < m; ++i) {
            const uint64_t w = T->rp[m + i], wsh = T->rp_sh[m + i];
            uint64_t* x = a + 2 * i * t;
            uint64_t* y = x + t;
            for (uint64_t j = 0; j < t; ++j) {
                uint64_t u = x[j];                  /* [0,4q) */
                u -= (u >= two_q) ? two_q : 0;      /* [0,2q) */
                uint64_t v = mul_shoup_lazy(y[j], w, wsh, q); /* [0,2q) */
                x[j] = u + v;                       /* [0,4q) */
                y[j] = u - v + two_q;               /* [0,4q) */
            }
        }
    }
    for (uint64_t j = 0; j < n; ++j) {
        uint64_t u = a[j];
        u -= (u >= two_q) ? two_q : 0;
        u -= (u >= q) ? q : 0;
        a[j] = u;
    }
}

static void ntt_inv_poly(const orc_limb* T, uint64_t* a) {
    const uint64_t q = T->q, two_q = 2 * q, n = T->n;
    uint64_t t = 1;
    for (uint64_t m = n; m > 1; m >>= 1) {
        const uint64_t h = m >> 1;
        uint64_t* x = a;
        for (uint64_t i = 0; i < h; ++i, x += 2 * t) {
            const uint64_t w = T->irp[h + i], wsh = T->irp_sh[h + i];
            uint64_t* y = x + t;
            for (uint64_t j = 0; j < t; ++j) {
                uint64_t u = x[j], v = y[j];        /* both [0,2q) */
                uint64_t s = u + v;                 /* [0,4q) */
                s -= (s >= two_q) ? two_q : 0;
                x[j] = s;                           /* [0,2q) */
                y[j] = mul_shoup_lazy(u - v + two_q, w, wsh, q); /* [0,2q) */
            }
        }
        t <<= 1;
    }
    for (uint64_t j = 0; j < n; ++j) {
        uint64_t u = mul_shoup_lazy(a[j], T->ninv, T->ninv_sh, q);
        a[j] = u - ((u >= q) ? q : 0);
    }
}

/* ------------------------------------------------------------------ batched public entry points */
static int clamp_threads(int threads) {
#ifdef _OPENMP
    int mx = omp_get_max_threads();
    if (threads <= 0 || threads > mx) threads = mx;
    return threads;
#else
    (void)threads;
    return 1;
#endif
}

int orc_max_threads(void) { return clamp_threads(0); }

void orc_ntt_fwd(const orc_ctx* c, uint64_t* io, size_t n_rns_polys, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long p = 0; p < (long long)(n_rns_polys * L); ++p) ntt_fwd_poly(&c->limb[p % L], io + (size_t)p * n);
}

void orc_ntt_inv(const orc_ctx* c, uint64_t* io, size_t n_rns_polys, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long p = 0; p < (long long)(n_rns_polys * L); ++p) ntt_inv_poly(&c->limb[p % L], io + (size_t)p * n);
}

/* op: 0 mul, 1 mul_add (out += a*b), 2 add, 3 sub, 4 negate (b ignored) */
void orc_dyadic(const orc_ctx* c, int op, uint64_t* out, const uint64_t* a, const uint64_t* b, size_t n_rns_polys, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long p = 0; p < (long long)(n_rns_polys * L); ++p) {
        const orc_limb* T = &c->limb[p % L];
        const uint64_t q = T->q;
        const size_t o = (size_t)p * n;
        for (size_t j = 0; j < n; ++j) {
            uint64_t x = a[o + j], y = b ? b[o + j] : 0, r;
            switch (op) {
                case 0: r = mulmod_barrett(x, y, T); break;
                case 1: r = mulmod_barrett(x, y, T) + out[o + j]; r -= (r >= q) ? q : 0; break;
                case 2: r = x + y; r -= (r >= q) ? q : 0; break;
                case 3: r = x >= y ? x - y : x + q - y; break;
                default: r = x ? q - x : 0; break;
            }
            out[o + j] = r;
        }
    }
}

/* ct x ct, coefficient domain in/out, through the NTT path.  a2,b2: [batch][2][L][N]; out3: [batch][3][L][N] */
void orc_ct_mul(const orc_ctx* c, uint64_t* out3, const uint64_t* a2, const uint64_t* b2, size_t batch, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    threads = clamp_threads(threads);
#pragma omp parallel num_threads(threads)
    {
        uint64_t* w = malloc(4 * n * sizeof(uint64_t));
#pragma omp for schedule(static)
        for (long long it = 0; it < (long long)(batch * L); ++it) {
            const size_t bi = (size_t)it / L, l = (size_t)it % L;
            const orc_limb* T = &c->limb[l];
            const uint64_t q = T->q;
            uint64_t *A0 = w, *A1 = w + n, *B0 = w + 2 * n, *B1 = w + 3 * n;
            memcpy(A0, a2 + ((bi * 2 + 0) * L + l) * n, n * 8);
            memcpy(A1, a2 + ((bi * 2 + 1) * L + l) * n, n * 8);
            memcpy(B0, b2 + ((bi * 2 + 0) * L + l) * n, n * 8);
            memcpy(B1, b2 + ((bi * 2 + 1) * L + l) * n, n * 8);
            ntt_fwd_poly(T, A0); ntt_fwd_poly(T, A1); ntt_fwd_poly(T, B0); ntt_fwd_poly(T, B1);
            uint64_t* c0 = out3 + ((bi * 3 + 0) * L + l) * n;
            uint64_t* c1 = out3 + ((bi * 3 + 1) * L + l) * n;
            uint64_t* c2 = out3 + ((bi * 3 + 2) * L + l) * n;
            for (size_t j = 0; j < n; ++j) {
                c0[j] = mulmod_barrett(A0[j], B0[j], T);
                uint64_t s = mulmod_barrett(A0[j], B1[j], T) + mulmod_barrett(A1[j], B0[j], T);
                c1[j] = s - ((s >= q) ? q : 0);
                c2[j] = mulmod_barrett(A1[j], B1[j], T);
            }
            ntt_inv_poly(T, c0); ntt_inv_poly(T, c1); ntt_inv_poly(T, c2);
        }
        free(w);
    }
}

/* Timed variant for bench.py's cpu_baseline leg: the same evaluator (orc_ct_mul), but measured the way a CPU deployment
 * would run it - private, page-aligned working copies first-touched by the thread that will use them (static schedule:
 * thread t owns a contiguous range of (ciphertext, limb) items, so its pages sit on its own NUMA node), `reps` timed
 * passes, best pass reported.  Without this the all-threads figure measured first-touch page faults on one node's
 * memory, not arithmetic (round-1 VERDICT).  Returns the best wall time of one pass in seconds; out3 receives the result. */
double orc_ct_mul_timed(const orc_ctx* c, uint64_t* out3, const uint64_t* a2, const uint64_t* b2, size_t batch, int threads, int reps) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    const size_t in_words = batch * 2 * L * n, out_words = batch * 3 * L * n;
    threads = clamp_threads(threads);
    if (reps < 1) reps = 1;
    uint64_t *a = NULL, *b = NULL, *o = NULL;
    if (posix_memalign((void**)&a, 4096, in_words * 8) || posix_memalign((void**)&b, 4096, in_words * 8) || posix_memalign((void**)&o, 4096, out_words * 8)) {
        free(a); free(b); free(o);
        return -1.0;
    }
    /* first touch with the schedule of the compute loop: item (bi, l) touches its 2 + 2 input and 3 output polynomials */
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long it = 0; it < (long long)(batch * L); ++it) {
        const size_t bi = (size_t)it / L, l = (size_t)it % L;
        for (int comp = 0; comp < 2; ++comp) {
            memcpy(a + ((bi * 2 + comp) * L + l) * n, a2 + ((bi * 2 + comp) * L + l) * n, n * 8);
            memcpy(b + ((bi * 2 + comp) * L + l) * n, b2 + ((bi * 2 + comp) * L + l) * n, n * 8);
        }
        for (int comp = 0; comp < 3; ++comp) memset(o + ((bi * 3 + comp) * L + l) * n, 0, n * 8);
    }
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        const double t0 = omp_get_wtime();
        orc_ct_mul(c, o, a, b, batch, threads);
        const double t = omp_get_wtime() - t0;
        if (t < best) best = t;
    }
    memcpy(out3, o, out_words * 8);
    free(a); free(b); free(o);
    return best;
}

/* what the hot loops of this build run on (reported next to the CPU baseline) */
const char* orc_isa(void) {
#if defined(__AVX512F__)
    return "x86-64 scalar 64x64->128 multiplies (mulx), compiled with AVX-512 enabled but not used by the modular arithmetic";
#elif defined(__BMI2__)
    return "x86-64-v3 scalar code: 64x64->128 multiplies (mulx/BMI2), Harvey lazy butterflies; no SIMD (60-bit moduli do not fit AVX-512 IFMA's 52-bit lanes)";
#else
    return "baseline x86-64 scalar code";
#endif
}

/* same tensor product by schoolbook convolution (ground truth; O(N^2), small sizes only) */
void orc_ct_mul_schoolbook(const orc_ctx* c, uint64_t* out3, const uint64_t* a2, const uint64_t* b2, size_t batch, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    threads = clamp_threads(threads);
#pragma omp parallel num_threads(threads)
    {
        uint64_t* w = malloc(n * sizeof(uint64_t));
#pragma omp for schedule(static)
        for (long long it = 0; it < (long long)(batch * L); ++it) {
            const size_t bi = (size_t)it / L, l = (size_t)it % L;
            const uint64_t q = c->limb[l].q;
            const uint64_t* a0 = a2 + ((bi * 2 + 0) * L + l) * n;
            const uint64_t* a1 = a2 + ((bi * 2 + 1) * L + l) * n;
            const uint64_t* b0 = b2 + ((bi * 2 + 0) * L + l) * n;
            const uint64_t* b1 = b2 + ((bi * 2 + 1) * L + l) * n;
            uint64_t* c0 = out3 + ((bi * 3 + 0) * L + l) * n;
            uint64_t* c1 = out3 + ((bi * 3 + 1) * L + l) * n;
            uint64_t* c2 = out3 + ((bi * 3 + 2) * L + l) * n;
            orc_schoolbook_negacyclic(c0, a0, b0, n, q);
            orc_schoolbook_negacyclic(c1, a0, b1, n, q);
            orc_schoolbook_negacyclic(w, a1, b0, n, q);
            for (size_t j = 0; j < n; ++j) { uint64_t s = c1[j] + w[j]; c1[j] = s - ((s >= q) ? q : 0); }
            orc_schoolbook_negacyclic(c2, a1, b1, n, q);
        }
        free(w);
    }
}

/* N1 relinearisation with RNS-digit keys (published algorithm: Bajard-Eynard-Hasan-Zucca RNS
 * decomposition / "BV" key switching without special prime):  (c0',c1') = (c0,c1) + sum_j [c2]_{q_j} (.) evk_j.
 * in3 [batch][3][L][N], out2 [batch][2][L][N] coefficient domain; evk [L][2][L][N] NTT domain. */
void orc_relinearize(const orc_ctx* c, uint64_t* out2, const uint64_t* in3, const uint64_t* evk, size_t batch, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    threads = clamp_threads(threads);
#pragma omp parallel num_threads(threads)
    {
        uint64_t* w = malloc(3 * n * sizeof(uint64_t));
#pragma omp for schedule(static)
        for (long long it = 0; it < (long long)(batch * L); ++it) {
            const size_t bi = (size_t)it / L, i = (size_t)it % L;
            const orc_limb* T = &c->limb[i];
            const uint64_t q = T->q;
            uint64_t *d = w, *acc0 = w + n, *acc1 = w + 2 * n;
            memset(acc0, 0, 2 * n * sizeof(uint64_t));
            for (size_t j = 0; j < L; ++j) {
                const uint64_t* c2j = in3 + ((bi * 3 + 2) * L + j) * n;
                for (size_t k = 0; k < n; ++k) d[k] = c2j[k] % q;
                ntt_fwd_poly(T, d);
                const uint64_t* k0 = evk + ((j * 2 + 0) * L + i) * n;
                const uint64_t* k1 = evk + ((j * 2 + 1) * L + i) * n;
                for (size_t k = 0; k < n; ++k) {
                    uint64_t s0 = acc0[k] + mulmod_barrett(d[k], k0[k], T); acc0[k] = s0 - ((s0 >= q) ? q : 0);
                    uint64_t s1 = acc1[k] + mulmod_barrett(d[k], k1[k], T); acc1[k] = s1 - ((s1 >= q) ? q : 0);
                }
            }
            ntt_inv_poly(T, acc0); ntt_inv_poly(T, acc1);
            const uint64_t* c0 = in3 + ((bi * 3 + 0) * L + i) * n;
            const uint64_t* c1 = in3 + ((bi * 3 + 1) * L + i) * n;
            uint64_t* o0 = out2 + ((bi * 2 + 0) * L + i) * n;
            uint64_t* o1 = out2 + ((bi * 2 + 1) * L + i) * n;
            for (size_t k = 0; k < n; ++k) {
                uint64_t s0 = acc0[k] + c0[k]; o0[k] = s0 - ((s0 >= q) ? q : 0);
                uint64_t s1 = acc1[k] + c1[k]; o1[k] = s1 - ((s1 >= q) ? q : 0);
            }
        }
        free(w);
    }
}

/* N1 second half: rescale = exact RNS divide-by-q_last-and-round (Cheon-Han-Kim-Kim-Song full-RNS CKKS; SEAL's
 * divide_and_round_q_last):  out_i = ((x_i + h) - ((x_last + h) mod q_last)) * q_last^-1 mod q_i,  h = floor(q_last/2). */
void orc_rescale(const orc_ctx* c, uint64_t* out, const uint64_t* in, size_t n_rns_polys) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    const uint64_t ql = c->limb[L - 1].q, h = ql / 2;
    for (size_t p = 0; p < n_rns_polys; ++p)
        for (size_t i = 0; i + 1 < L; ++i) {
            const uint64_t q = c->limb[i].q;
            const uint64_t inv = powmod(ql % q, q - 2, q);
            for (size_t k = 0; k < n; ++k) {
                const uint64_t t = (in[(p * L + L - 1) * n + k] + h) % ql;
                const uint64_t a = (uint64_t)(((u128)in[(p * L + i) * n + k] + h % q) % q);
                const uint64_t d = (a + q - t % q) % q;
                out[(p * (L - 1) + i) * n + k] = mulmod_slow(d, inv, q);
            }
        }
}

/* Hybrid key switching with one special prime P = last limb of the (extended) context c; data on the first Ld = L-1 limbs.
 * t = sum_{j<Ld} [digit_j] (.) key_j over all L limbs, then out = round(t / P) (+ c0, and + c1 when in_comps == 3).
 * in: [batch][in_comps][Ld][N], key: [Ld][2][L][N] (NTT domain), out2: [batch][2][Ld][N].  (GHS / hybrid key switching,
 * Gentry-Halevi-Smart 2012; RNS form as in Han-Ki 2020 with dnum = Ld.) */
void orc_keyswitch_hybrid(const orc_ctx* c, uint64_t* out2, const uint64_t* in, const uint64_t* key, size_t batch, int in_comps, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs, Ld = L - 1;
    const uint64_t P = c->limb[Ld].q, h = P / 2;
    threads = clamp_threads(threads);
#pragma omp parallel num_threads(threads)
    {
        uint64_t* t = malloc((2 * L + 1) * n * sizeof(uint64_t));   /* t[c][i][k] and a digit buffer */
        uint64_t* d = t + 2 * L * n;
#pragma omp for schedule(static)
        for (long long bi = 0; bi < (long long)batch; ++bi) {
            const uint64_t* digits = in + (((size_t)bi * in_comps + (in_comps - 1)) * Ld) * n;
            for (size_t i = 0; i < L; ++i) {
                const orc_limb* T = &c->limb[i];
                const uint64_t q = T->q;
                uint64_t *acc0 = t + (0 * L + i) * n, *acc1 = t + (1 * L + i) * n;
                memset(acc0, 0, n * sizeof(uint64_t)); memset(acc1, 0, n * sizeof(uint64_t));
                for (size_t j = 0; j < Ld; ++j) {
                    for (size_t k = 0; k < n; ++k) d[k] = digits[j * n + k] % q;
                    ntt_fwd_poly(T, d);
                    const uint64_t* k0 = key + ((j * 2 + 0) * L + i) * n;
                    const uint64_t* k1 = key + ((j * 2 + 1) * L + i) * n;
                    for (size_t k = 0; k < n; ++k) {
                        uint64_t s0 = acc0[k] + mulmod_barrett(d[k], k0[k], T); acc0[k] = s0 - ((s0 >= q) ? q : 0);
                        uint64_t s1 = acc1[k] + mulmod_barrett(d[k], k1[k], T); acc1[k] = s1 - ((s1 >= q) ? q : 0);
                    }
                }
                ntt_inv_poly(T, acc0); ntt_inv_poly(T, acc1);
            }
            for (int comp = 0; comp < 2; ++comp)
                for (size_t i = 0; i < Ld; ++i) {
                    const uint64_t q = c->limb[i].q;
                    const uint64_t inv = powmod(P % q, q - 2, q);
                    const int add = (in_comps == 3) || (comp == 0);
                    for (size_t k = 0; k < n; ++k) {
                        const uint64_t tl = (t[(comp * L + Ld) * n + k] + h) % P;
                        const uint64_t a = (uint64_t)(((u128)t[(comp * L + i) * n + k] + h % q) % q);
                        uint64_t r = mulmod_slow((a + q - tl % q) % q, inv, q);
                        if (add) { r += in[(((size_t)bi * in_comps + comp) * Ld + i) * n + k]; r -= (r >= q) ? q : 0; }
                        out2[(((size_t)bi * 2 + comp) * Ld + i) * n + k] = r;
                    }
                }
        }
        free(t);
    }
}

/* N3: Galois automorphism a(X) -> a(X^g) on n_rns_polys RNS polynomials (coefficient domain), scatter form:
 * coefficient i goes to index i*g mod 2N, negated when that index is >= N (X^N = -1). */
/* N3, HOISTED rotations (Halevi-Shoup): k rotations of ONE ciphertext share the digit decomposition.  The digits of c1 are
 * lifted to every limb of the extended basis FIRST and the automorphism is applied to the lifted digit (in the NTT domain it is
 * a permutation, which is what the GPU exploits):  t_g = sum_j sigma_g(lift([c1]_{q_j})) (.) key_{g,j},
 * out_g = (sigma_g(c0), 0) + round(t_g / P).  This differs from orc_keyswitch_hybrid applied to sigma_g(ct) by a multiple of
 * q_j in the lifted digits (sigma_g negates before the lift there, after it here) - both are valid key switches of the same
 * ciphertext, but the words differ, so the hoisted GPU path has its own restatement.
 * c = the EXTENDED context; in2: [2][Ld][N] (one item); keys: [k][Ld][2][L][N] (NTT domain); out2: [k][2][Ld][N]. */
void orc_rotate_hoisted(const orc_ctx* c, uint64_t* out2, const uint64_t* in2, const uint32_t* elts, const uint64_t* keys, size_t k_rot, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs, Ld = L - 1;
    const uint64_t P = c->limb[Ld].q, h = P / 2;
    threads = clamp_threads(threads);
#pragma omp parallel num_threads(threads)
    {
        uint64_t* t = malloc((2 * L + 2) * n * sizeof(uint64_t));
        uint64_t *d = t + 2 * L * n, *dg = d + n;
#pragma omp for schedule(static)
        for (long long it = 0; it < (long long)k_rot; ++it) {
            const uint32_t g = elts[it];
            const uint64_t* key = keys + (size_t)it * Ld * 2 * L * n;
            const uint64_t* c1 = in2 + Ld * n;
            for (size_t i = 0; i < L; ++i) {
                const orc_limb* T = &c->limb[i];
                const uint64_t q = T->q;
                uint64_t *acc0 = t + (0 * L + i) * n, *acc1 = t + (1 * L + i) * n;
                memset(acc0, 0, n * sizeof(uint64_t)); memset(acc1, 0, n * sizeof(uint64_t));
                for (size_t j = 0; j < Ld; ++j) {
                    for (size_t x = 0; x < n; ++x) d[x] = c1[j * n + x] % q;                 /* lift */
                    for (size_t x = 0; x < n; ++x) {                                         /* then rotate, mod q_i */
                        const size_t idx = (x * (size_t)g) & (2 * n - 1);
                        if (idx < n) dg[idx] = d[x]; else dg[idx - n] = d[x] ? q - d[x] : 0;
                    }
                    ntt_fwd_poly(T, dg);
                    const uint64_t* k0 = key + ((j * 2 + 0) * L + i) * n;
                    const uint64_t* k1 = key + ((j * 2 + 1) * L + i) * n;
                    for (size_t x = 0; x < n; ++x) {
                        uint64_t s0 = acc0[x] + mulmod_barrett(dg[x], k0[x], T); acc0[x] = s0 - ((s0 >= q) ? q : 0);
                        uint64_t s1 = acc1[x] + mulmod_barrett(dg[x], k1[x], T); acc1[x] = s1 - ((s1 >= q) ? q : 0);
                    }
                }
                ntt_inv_poly(T, acc0); ntt_inv_poly(T, acc1);
            }
            for (int comp = 0; comp < 2; ++comp)
                for (size_t i = 0; i < Ld; ++i) {
                    const uint64_t q = c->limb[i].q;
                    const uint64_t inv = powmod(P % q, q - 2, q);
                    uint64_t* o = out2 + (((size_t)it * 2 + comp) * Ld + i) * n;
                    for (size_t x = 0; x < n; ++x) {
                        const uint64_t tl = (t[(comp * L + Ld) * n + x] + h) % P;
                        const uint64_t a = (uint64_t)(((u128)t[(comp * L + i) * n + x] + h % q) % q);
                        o[x] = mulmod_slow((a + q - tl % q) % q, inv, q);
                    }
                    if (comp == 0)                                                           /* + sigma_g(c0) */
                        for (size_t x = 0; x < n; ++x) {
                            const size_t idx = (x * (size_t)g) & (2 * n - 1);
                            const uint64_t v = in2[i * n + x];
                            const uint64_t sv = idx < n ? v : (v ? q - v : 0);
                            uint64_t* dst = &o[idx & (n - 1)];
                            uint64_t r = *dst + sv; *dst = r - ((r >= q) ? q : 0);
                        }
                }
        }
        free(t);
    }
}

/* N3, round 3 ("double hoisting", Bossuat et al. 2021): the hoisted rotations BEFORE the division by P, left in the NTT domain
 * over the extended basis.  Definition form (automorphism in the coefficient domain, then the transform - the GPU permutes in the
 * NTT domain instead):  block 0 = (P c0, P c1);  block 1 + r =
 *   ( sum_j NTT(sigma_g lift([c1]_{q_j})) (.) key_{g,j,0} + P NTT(sigma_g c0),  sum_j NTT(sigma_g lift([c1]_{q_j})) (.) key_{g,j,1} ),
 * the P-multiples vanishing on the special limb.  c = EXTENDED context; in2: [2][Ld][N] (one item, coefficient domain);
 * keys: [k][Ld][2][L][N]; out: [1 + k][2][L][N]. */
void orc_rotate_hoisted_qp(const orc_ctx* c, uint64_t* out, const uint64_t* in2, const uint32_t* elts, const uint64_t* keys, size_t k_rot, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs, Ld = L - 1;
    const uint64_t P = c->limb[Ld].q;
    threads = clamp_threads(threads);
#pragma omp parallel num_threads(threads)
    {
        uint64_t* d = malloc(2 * n * sizeof(uint64_t));
        uint64_t* dg = d + n;
#pragma omp for schedule(static)
        for (long long it = -1; it < (long long)k_rot; ++it) {
            const uint32_t g = it < 0 ? 1u : elts[it];
            const uint64_t* key = it < 0 ? NULL : keys + (size_t)it * Ld * 2 * L * n;
            uint64_t* o = out + (size_t)(it + 1) * 2 * L * n;
            for (size_t i = 0; i < L; ++i) {
                const orc_limb* T = &c->limb[i];
                const uint64_t q = T->q, pm = P % q;
                uint64_t *acc0 = o + (0 * L + i) * n, *acc1 = o + (1 * L + i) * n;
                memset(acc0, 0, n * sizeof(uint64_t)); memset(acc1, 0, n * sizeof(uint64_t));
                for (int comp = 0; comp < 2 && i < Ld; ++comp) {          /* P sigma_g(c_comp): comp 1 only for the identity block */
                    if (comp == 1 && it >= 0) break;
                    for (size_t x = 0; x < n; ++x) {
                        const size_t idx = (x * (size_t)g) & (2 * n - 1);
                        const uint64_t v = in2[(comp * Ld + i) * n + x];
                        if (idx < n) dg[idx] = v; else dg[idx - n] = v ? q - v : 0;
                    }
                    ntt_fwd_poly(T, dg);
                    uint64_t* a = comp ? acc1 : acc0;
                    for (size_t x = 0; x < n; ++x) a[x] = mulmod_slow(dg[x], pm, q);
                }
                if (it < 0) continue;
                for (size_t j = 0; j < Ld; ++j) {
                    for (size_t x = 0; x < n; ++x) d[x] = in2[(Ld + j) * n + x] % q;          /* lift */
                    for (size_t x = 0; x < n; ++x) {                                          /* then rotate, mod q_i */
                        const size_t idx = (x * (size_t)g) & (2 * n - 1);
                        if (idx < n) dg[idx] = d[x]; else dg[idx - n] = d[x] ? q - d[x] : 0;
                    }
                    ntt_fwd_poly(T, dg);
                    const uint64_t* k0 = key + ((j * 2 + 0) * L + i) * n;
                    const uint64_t* k1 = key + ((j * 2 + 1) * L + i) * n;
                    for (size_t x = 0; x < n; ++x) {
                        uint64_t s0 = acc0[x] + mulmod_barrett(dg[x], k0[x], T); acc0[x] = s0 - ((s0 >= q) ? q : 0);
                        uint64_t s1 = acc1[x] + mulmod_barrett(dg[x], k1[x], T); acc1[x] = s1 - ((s1 >= q) ? q : 0);
                    }
                }
            }
        }
        free(d);
    }
}

/* N3, round 3: the key inner product of hybrid key switching alone, left in the NTT domain over Q P (the giant steps' terms):
 * out[b][comp][i] = sum_j NTT_i(lift_i([c1]_{q_j})) (.) key_{j,comp,i}.  in2: [batch][2][Ld][N] coefficient domain, key: [Ld][2][L][N],
 * out: [batch][2][L][N].  Same inner loop as orc_keyswitch_hybrid, without its inverse transforms and its division. */
void orc_switch_key_qp(const orc_ctx* c, uint64_t* out, const uint64_t* in2, const uint64_t* key, size_t batch, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs, Ld = L - 1;
    threads = clamp_threads(threads);
#pragma omp parallel num_threads(threads)
    {
        uint64_t* d = malloc(n * sizeof(uint64_t));
#pragma omp for schedule(static)
        for (long long bi = 0; bi < (long long)batch; ++bi) {
            const uint64_t* digits = in2 + (((size_t)bi * 2 + 1) * Ld) * n;
            for (size_t i = 0; i < L; ++i) {
                const orc_limb* T = &c->limb[i];
                const uint64_t q = T->q;
                uint64_t *acc0 = out + (((size_t)bi * 2 + 0) * L + i) * n, *acc1 = out + (((size_t)bi * 2 + 1) * L + i) * n;
                memset(acc0, 0, n * sizeof(uint64_t)); memset(acc1, 0, n * sizeof(uint64_t));
                for (size_t j = 0; j < Ld; ++j) {
                    for (size_t k = 0; k < n; ++k) d[k] = digits[j * n + k] % q;
                    ntt_fwd_poly(T, d);
                    const uint64_t* k0 = key + ((j * 2 + 0) * L + i) * n;
                    const uint64_t* k1 = key + ((j * 2 + 1) * L + i) * n;
                    for (size_t k = 0; k < n; ++k) {
                        uint64_t s0 = acc0[k] + mulmod_barrett(d[k], k0[k], T); acc0[k] = s0 - ((s0 >= q) ? q : 0);
                        uint64_t s1 = acc1[k] + mulmod_barrett(d[k], k1[k], T); acc1[k] = s1 - ((s1 >= q) ? q : 0);
                    }
                }
            }
        }
        free(d);
    }
}

/* Round 4: exact base extension / scale-and-round between limb ranges of one context (include/dpfhe.h dpfhe_base_extend,
 * dpfhe_scale_round).  Restated with 128-bit arithmetic on the mixed-radix (Garner) digits; oracle/pyoracle.py holds the DEFINITION
 * (Python big integers: CRT, centre, reduce) that pins this function.
 *   mode 0: out_j = X mod p_j;   mode 1: out_j = round(mul * X_all / Qs) mod p_j  where the sources are mul * x_i and the input holds all limbs. */
static uint64_t mulmod_(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((u128)a * b % q); }
static uint64_t powmod_(uint64_t b, uint64_t e, uint64_t q) { uint64_t r = 1; b %= q; while (e) { if (e & 1) r = mulmod_(r, b, q); b = mulmod_(b, b, q); e >>= 1; } return r; }
void orc_base_extend(const orc_ctx* c, int mode, uint64_t* out, size_t out_stride_limbs, const uint64_t* in, size_t in_stride_limbs, uint32_t src0, uint32_t ns,
                     uint32_t dst0, uint32_t nd, uint64_t mul, size_t n_polys) {
    const size_t n = (size_t)1 << c->log2n;
    uint64_t q[10], p[20], inv[10][10], half[10];
    for (uint32_t i = 0; i < ns; ++i) q[i] = c->limb[src0 + i].q;
    for (uint32_t j = 0; j < nd; ++j) p[j] = c->limb[dst0 + j].q;
    for (uint32_t i = 0; i < ns; ++i) for (uint32_t k = i + 1; k < ns; ++k) inv[i][k] = powmod_(q[i] % q[k], q[k] - 2, q[k]);
    { u128 carry = 0; for (int k = (int)ns - 1; k >= 0; --k) { u128 cur = carry * q[k] + (q[k] - 1); half[k] = (uint64_t)(cur / 2); carry = cur & 1; } }
    for (size_t pi = 0; pi < n_polys; ++pi)
        for (size_t w = 0; w < n; ++w) {
            uint64_t v[10];
            for (uint32_t k = 0; k < ns; ++k) {
                uint64_t t = in[(pi * in_stride_limbs + (mode ? src0 : 0) + k) * n + w];
                if (mode) t = mulmod_(t, mul % q[k], q[k]);
                for (uint32_t i = 0; i < k; ++i) t = mulmod_((t + q[k] - v[i] % q[k]) % q[k], inv[i][k], q[k]);
                v[k] = t;
            }
            int neg = 0;
            for (int k = (int)ns - 1; k >= 0; --k) if (v[k] != half[k]) { neg = v[k] > half[k]; break; }
            for (uint32_t j = 0; j < nd; ++j) {
                uint64_t acc = v[ns - 1] % p[j], Qm = 1;
                for (int k = (int)ns - 2; k >= 0; --k) acc = (uint64_t)(((u128)acc * (q[k] % p[j]) + v[k]) % p[j]);
                for (uint32_t i = 0; i < ns; ++i) Qm = mulmod_(Qm, q[i] % p[j], p[j]);
                if (neg) acc = (acc + p[j] - Qm) % p[j];
                if (mode) {
                    const uint64_t x = in[(pi * in_stride_limbs + dst0 + j) * n + w];
                    acc = mulmod_((mulmod_(x, mul % p[j], p[j]) + p[j] - acc) % p[j], powmod_(Qm, p[j] - 2, p[j]), p[j]);
                }
                out[(pi * out_stride_limbs + j) * n + w] = acc;
            }
        }
}

void orc_apply_galois(const orc_ctx* c, uint64_t* out, const uint64_t* in, size_t n_rns_polys, uint32_t g) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    for (size_t p = 0; p < n_rns_polys * L; ++p) {
        const uint64_t q = c->limb[p % L].q;
        for (size_t i = 0; i < n; ++i) {
            const size_t idx = (i * (size_t)g) & (2 * n - 1);
            const uint64_t v = in[p * n + i];
            if (idx < n) out[p * n + idx] = v;
            else out[p * n + idx - n] = v ? q - v : 0;
        }
    }
}

/* key switch after an automorphism: (c0 + sum_j [c1]_{q_j} (.) key_j[0], sum_j [c1]_{q_j} (.) key_j[1]); in2,out2 [batch][2][L][N] */
void orc_switch_key(const orc_ctx* c, uint64_t* out2, const uint64_t* in2, const uint64_t* key, size_t batch, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    threads = clamp_threads(threads);
#pragma omp parallel num_threads(threads)
    {
        uint64_t* w = malloc(3 * n * sizeof(uint64_t));
#pragma omp for schedule(static)
        for (long long it = 0; it < (long long)(batch * L); ++it) {
            const size_t bi = (size_t)it / L, i = (size_t)it % L;
            const orc_limb* T = &c->limb[i];
            const uint64_t q = T->q;
            uint64_t *d = w, *acc0 = w + n, *acc1 = w + 2 * n;
            memset(acc0, 0, 2 * n * sizeof(uint64_t));
            for (size_t j = 0; j < L; ++j) {
                const uint64_t* c1j = in2 + ((bi * 2 + 1) * L + j) * n;
                for (size_t k = 0; k < n; ++k) d[k] = c1j[k] % q;
                ntt_fwd_poly(T, d);
                const uint64_t* k0 = key + ((j * 2 + 0) * L + i) * n;
                const uint64_t* k1 = key + ((j * 2 + 1) * L + i) * n;
                for (size_t k = 0; k < n; ++k) {
                    uint64_t s0 = acc0[k] + mulmod_barrett(d[k], k0[k], T); acc0[k] = s0 - ((s0 >= q) ? q : 0);
                    uint64_t s1 = acc1[k] + mulmod_barrett(d[k], k1[k], T); acc1[k] = s1 - ((s1 >= q) ? q : 0);
                }
            }
            ntt_inv_poly(T, acc0); ntt_inv_poly(T, acc1);
            const uint64_t* c0 = in2 + ((bi * 2 + 0) * L + i) * n;
            uint64_t* o0 = out2 + ((bi * 2 + 0) * L + i) * n;
            uint64_t* o1 = out2 + ((bi * 2 + 1) * L + i) * n;
            for (size_t k = 0; k < n; ++k) {
                uint64_t s0 = acc0[k] + c0[k]; o0[k] = s0 - ((s0 >= q) ? q : 0);
                o1[k] = acc1[k];
            }
        }
        free(w);
    }
}

/* y[rows][comps][L][N] = sum_j W[rows][cols][L][N] (.) x[cols][comps][L][N]   (all NTT domain) */
void orc_matvec_plain(const orc_ctx* c, uint64_t* y, const uint64_t* W, const uint64_t* x, size_t rows, size_t cols, size_t comps, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long it = 0; it < (long long)(rows * comps * L); ++it) {
        const size_t i = (size_t)it / (comps * L), cc = ((size_t)it / L) % comps, l = (size_t)it % L;
        const orc_limb* T = &c->limb[l];
        uint64_t* yo = y + ((i * comps + cc) * L + l) * n;
        for (size_t k = 0; k < n; ++k) {
            uint64_t acc = 0;
            for (size_t j = 0; j < cols; ++j) {
                uint64_t p = mulmod_barrett(W[((i * cols + j) * L + l) * n + k], x[((j * comps + cc) * L + l) * n + k], T);
                acc += p; acc -= (acc >= T->q) ? T->q : 0;
            }
            yo[k] = acc;
        }
    }
}

/* scalar-weight matvec: y[rows][comps][L][N] = sum_j w[rows][cols][L] * x[cols][comps][L][N] */
void orc_matvec_scalar(const orc_ctx* c, uint64_t* y, const uint64_t* w, const uint64_t* x, size_t rows, size_t cols, size_t comps, int threads) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    threads = clamp_threads(threads);
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long it = 0; it < (long long)(rows * comps * L); ++it) {
        const size_t i = (size_t)it / (comps * L), cc = ((size_t)it / L) % comps, l = (size_t)it % L;
        const orc_limb* T = &c->limb[l];
        uint64_t* yo = y + ((i * comps + cc) * L + l) * n;
        for (size_t k = 0; k < n; ++k) {
            uint64_t acc = 0;
            for (size_t j = 0; j < cols; ++j) {
                uint64_t p = mulmod_barrett(w[(i * cols + j) * L + l] % T->q, x[((j * comps + cc) * L + l) * n + k], T);
                acc += p; acc -= (acc >= T->q) ? T->q : 0;
            }
            yo[k] = acc;
        }
    }
}

/* sum of `count` ciphertexts of `comps` components into one (A8 local_reduce) */
void orc_reduce_sum(const orc_ctx* c, uint64_t* out, const uint64_t* in, size_t count, size_t comps) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs, words = comps * L * n;
    for (size_t w = 0; w < words; ++w) {
        const uint64_t q = c->limb[(w / n) % L].q;
        uint64_t acc = 0;
        for (size_t i = 0; i < count; ++i) { acc += in[i * words + w]; acc -= (acc >= q) ? q : 0; }
        out[w] = acc;
    }
}

/* splitmix64 stream of words uniform-ish in [0, q_limb): layout [n_rns_polys][L][N] (SURVEY.md App. B generator) */
void orc_fill_splitmix(const orc_ctx* c, uint64_t* out, size_t n_rns_polys, uint64_t seed) {
    const size_t n = 1ull << c->log2n, L = c->n_limbs;
    uint64_t s = seed;
    for (size_t p = 0; p < n_rns_polys * L; ++p) {
        const uint64_t q = c->limb[p % L].q;
        for (size_t j = 0; j < n; ++j) {
            s += 0x9E3779B97F4A7C15ull;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            out[p * n + j] = z % q;
        }
    }
}
