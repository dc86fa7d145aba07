// tables.h - host-side construction of the per-limb constants and twiddle tables (A0 of SURVEY.md
// section 8a).  Plain C++17, no HIP: shared by the C-ABI (dpfhe_cabi.hip) and tools/emulate.cpp.
#pragma once
#include <vector>

#include "modarith.h"

namespace dpfhe {

typedef unsigned __int128 u128;

inline u64 h_mulmod(u64 a, u64 b, u64 q) { return (u64)((u128)a * b % q); }
inline u64 h_powmod(u64 b, u64 e, u64 q) {
    u64 r = 1;
    b %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = h_mulmod(r, b, q);
        b = h_mulmod(b, b, q);
    }
    return r;
}
// deterministic Miller-Rabin for n < 2^64 (the first twelve primes as bases suffice below 3.3 * 10^24)
inline bool h_is_prime(u64 n) {
    if (n < 2) return false;
    const u64 bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (u64 p : bases) {
        if (n == p) return true;
        if (n % p == 0) return false;
    }
    u64 d = n - 1;
    int r = 0;
    while (!(d & 1)) { d >>= 1; ++r; }
    for (u64 a : bases) {
        u64 x = h_powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int i = 1; i < r && witness; ++i) {
            x = h_mulmod(x, x, n);
            if (x == n - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}
inline u64 h_shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }
inline u32 h_brv(u32 x, int bits) {
    u32 r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

// FoldArith twiddle of w: w and w 2^32 mod q, each split into 30-bit halves (modarith.h TwFold)
inline TwFold h_tw_fold(u64 w, u64 q) {
    const u64 ws = (u64)(((u128)w << 32) % q), m30 = (1ull << 30) - 1;
    return TwFold{(w & m30) | ((w >> 30) << 32), (ws & m30) | ((ws >> 30) << 32)};
}

template <class Tw> inline Tw h_make_tw(u64 w, u64 q);
template <> inline TwFold h_make_tw<TwFold>(u64 w, u64 q) { return h_tw_fold(w, q); }
template <> inline TwShoup h_make_tw<TwShoup>(u64 w, u64 q) { return TwShoup{w, (u64)(((u128)w << 64) / q)}; }
// F64Arith: w and the correctly rounded quotient w / q (both exact or within half an ulp: w < q < 2^47 are exact doubles)
template <> inline TwF64 h_make_tw<TwF64>(u64 w, u64 q) { return TwF64{(double)w, (double)w / (double)q}; }

// ---- per-limb arithmetic classes (round 6): which policy's transform kernels a limb runs on ------------------------------------------
// kClassFold: the pinned shape 2^60 - d; kClassF64: any prime below 2^47; kClassFoldScaled: 2^k - d0, 48 <= k < 60, d0 2^(60-k) < 2^24;
// kClassF64Wide: any other prime below 2^50 (doubles again, with reductions inside the transforms); kClassShoup: everything else (51 ... 59-bit primes far
// from a power of two).
enum LimbClass { kClassShoup = 0, kClassFold = 1, kClassF64 = 2, kClassFoldScaled = 3, kClassF64Wide = 4, kLimbClasses = 5 };
inline int h_bit_length(u64 v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }
// the scaling shift 60 - k of a FoldScaledArith prime, 0 when q is not one
inline int fold_scaled_shift(u64 q) {
    const int k = h_bit_length(q);
    if (k < 48 || k > 59) return 0;
    const u64 d0 = (1ull << k) - q;
    return (d0 << (60 - k)) < (1ull << 24) ? 60 - k : 0;
}
inline bool f64_eligible(u64 q) { return q < (1ull << F64Arith::kMaxBits); }
inline LimbClass limb_class(u64 q) {
    if (q < (1ull << 60) && ((1ull << 60) - q) < (1ull << 24)) return kClassFold;
    if (f64_eligible(q)) return kClassF64;
    if (fold_scaled_shift(q)) return kClassFoldScaled;
    if (q < (1ull << F64WideArith::kMaxBits)) return kClassF64Wide;
    return kClassShoup;
}
// FoldScaledArith twiddle of w (< q): w unscaled, the companion w 2^32 modulo the SCALED modulus q' = q 2^sh (modarith.h)
inline TwFold h_tw_fold_scaled(u64 w, u64 q, int sh) { return h_tw_fold(w, q << sh); }
// the LimbConst a class's transform kernels read (modarith.h: FoldScaledArith / F64Arith reinterpret fields); `lc` is build_limb_tables' record
inline LimbConst limb_const_of_class(const LimbConst& lc, LimbClass cls) {
    LimbConst c = lc;
    if (cls == kClassFoldScaled) {
        const int sh = fold_scaled_shift(lc.q);
        c.d = ((1ull << (60 - sh)) - lc.q) << sh;
        c.pad1 = (u64)sh;
    } else if (cls == kClassF64 || cls == kClassF64Wide) {
        const double qd = (double)lc.q, qi = 1.0 / qd;
        c.d = 0;
        c.ninv = __builtin_bit_cast(u64, qd);
        c.ninv_sh = __builtin_bit_cast(u64, qi);
    } else if (cls == kClassShoup) {
        c.d = 0;
    }
    return c;
}

// Split transform (N > 16384, ntt_top.h): after the first log_n1 radix-2 stages the remaining ones act inside blocks of
// N2 = N >> log_n1 consecutive words, and block r runs an ordinary N2-point merged transform whose twiddles are the
// sub-tree of the big table rooted at node N1 + r:  sub[m + i] = table[(N1 + r) m + i]  (m a power of two < N2, i < m).
inline std::vector<u64> subtree_table(const std::vector<u64>& table, int log2n, int log_n1, size_t r) {
    const size_t n2 = (size_t)1 << (log2n - log_n1), root = ((size_t)1 << log_n1) + r;
    std::vector<u64> sub(n2, 0);
    for (size_t m = 1; m < n2; m <<= 1)
        for (size_t i = 0; i < m; ++i) sub[m + i] = table[root * m + i];
    return sub;
}

// q = 2^60 - d with d < 2^24: eligible for FoldArith
inline bool fold_eligible(u64 q) { return q < (1ull << 60) && ((1ull << 60) - q) < (1ull << 24); }

// Device layout of a twiddle table for kernel geometry (log2n, loge): the sections of the window-0 stages
// (bit positions < perm_stages = Geo::kPermStages) are transposed from [thread][entry] to [entry][thread] so that a
// wave fetches lane-contiguous entries (ntt_core.h tw_index).  T = N >> loge threads.
template <class Tw>
inline void permute_window0(std::vector<Tw>& t, int log2n, int loge, int perm_stages) {
    const size_t n = (size_t)1 << log2n, T = n >> loge;
    std::vector<Tw> src(t);
    for (int pos = 0; pos < perm_stages && pos < loge - 1; ++pos) {
        const size_t m = n >> (pos + 1), cnt = m / T;   // cnt = 2^(loge-1-pos) entries per thread
        for (size_t tid = 0; tid < T; ++tid)
            for (size_t i = 0; i < cnt; ++i) t[m + i * T + tid] = src[m + tid * cnt + i];
    }
}

struct HostLimbTables {
    LimbConst lc;
    std::vector<u64> rp, irp;        // psi^brv(i), psi^-brv(i)           (index m + i of a CT/GS walk)
    std::vector<u64> rp_sh, irp_sh;  // Shoup companions
    u64 w_last, w_last_sh;           // irp[1] * N^-1  (last inverse stage, N^-1 folded in)
};

// returns 0, or a deeppowers::common::ErrorCode number (INVALID_ARGUMENT = 2000)
inline int build_limb_tables(int log2n, u64 q, u64 psi, HostLimbTables& t) {
    const u64 n = 1ull << log2n;
    if (log2n < 1 || log2n > 20) return 2000;
    if (q < 3 || (q >> 60) || (q - 1) % (2 * n) != 0) return 2000;
    if (psi == 0 || psi >= q || h_powmod(psi, n, q) != q - 1) return 2000;  // order exactly 2N
    t.rp.assign(n, 0); t.irp.assign(n, 0); t.rp_sh.assign(n, 0); t.irp_sh.assign(n, 0);
    const u64 ipsi = h_powmod(psi, q - 2, q);
    u64 pw = 1, ipw = 1;
    for (u64 i = 0; i < n; ++i) {
        const u32 r = h_brv((u32)i, log2n);
        t.rp[r] = pw; t.rp_sh[r] = h_shoup(pw, q);
        t.irp[r] = ipw; t.irp_sh[r] = h_shoup(ipw, q);
        pw = h_mulmod(pw, psi, q); ipw = h_mulmod(ipw, ipsi, q);
    }
    LimbConst& c = t.lc;
    c.q = q;
    c.d = fold_eligible(q) ? (1ull << 60) - q : 0;
    c.ninv = h_powmod(n % q, q - 2, q);
    c.ninv_sh = h_shoup(c.ninv, q);
    const u128 ratio = (~(u128)0) / q;
    c.br_hi = (u64)(ratio >> 64); c.br_lo = (u64)ratio;
    c.two64 = (u64)((((u128)1) << 64) % q);
    c.pad1 = 0;
    t.w_last = h_mulmod(t.irp[1], c.ninv, q);
    t.w_last_sh = h_shoup(t.w_last, q);
    return 0;
}

}  // namespace dpfhe
