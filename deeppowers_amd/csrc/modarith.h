// modarith.h - 64-bit modular arithmetic for the RLWE hot path (host + gfx950 device).
//
// No reference counterpart: deeppowers/deeppowers has no modular arithmetic (SURVEY.md section 0;
// its widest integer type is INT32, /root/reference/src/core/hal/hal.hpp:27-33).  Two policies:
//
//   ShoupArith - any prime q < 2^60.  Harvey lazy butterflies with Shoup companions
//                (w' = floor(w 2^64 / q)):  10 32x32 multiplies per butterfly.
//   FoldArith  - primes just below 2^60, q = 2^60 - d with d < 2^24 (every prime of SURVEY.md
//                Appendix A has d < 2^20).  2^60 = d (mod q), so products are folded with small
//                multiplies instead of a quotient estimate, and a 3-instruction partial reduction
//                lets butterflies run without per-stage corrections (static bound plans in
//                ntt_core.h).  Twiddle multiplies use a split companion table (w and w 2^32 mod q in
//                30-bit halves): 7 v_mad_u64_u32 + 2 cheap ops, every addend a natural 64-bit
//                register pair.  Variable x variable products: mul60 (7 multiply-adds + the folds' shifts), or one
//                factor turned into such a twiddle on the fly (prod_tw / mul_ptw_add: the fused multiply's tensor step).
//
// Everything is written on 32-bit halves through mad32(a,b,c) = a*b + c, which is exactly one
// v_mad_u64_u32 on gfx950 (there is no 64x64 vector multiply on CDNA4).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DPF_HD __host__ __device__ __forceinline__
#else
#define DPF_HD inline __attribute__((always_inline))
#endif

namespace dpfhe {

typedef uint64_t u64;
typedef uint32_t u32;

// tools/emulate.cpp defines DPFHE_EMU_CHECK: host-only counters of broken lazy-arithmetic preconditions
#if defined(DPFHE_EMU_CHECK) && !defined(__HIPCC__)
extern long g_emu_overflows;
#define DPFHE_EMU_ASSERT(cond) do { if (!(cond)) ++g_emu_overflows; } while (0)
#else
#define DPFHE_EMU_ASSERT(cond) do { } while (0)
#endif

struct LimbConst {  // one per RNS limb, read through scalar loads (limb index is workgroup-uniform)
    u64 q;
    u64 d;          // 2^60 - q (FoldArith only; 0 when not applicable)
    u64 ninv;       // N^-1 mod q
    u64 ninv_sh;    // floor(ninv 2^64 / q)
    u64 br_hi;      // floor(2^128 / q), high word   (generic Barrett)
    u64 br_lo;      //                   low word
    u64 two64;      // 2^64 mod q
    u64 pad1;       // FoldScaledArith: 60 - k, the scaling shift of a 2^k - d0 limb (0 otherwise)
};

DPF_HD u64 mad32(u32 a, u32 b, u64 c) { return (u64)a * b + c; }

DPF_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// 2a + c in one v_lshl_add_u64 (hipcc emits a 64-bit shift and an add for the C expression).
// c must be WAVE-UNIFORM (a limb constant): it is passed as a scalar register pair.
DPF_HD u64 shl1_add(u64 a, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    u64 r;
    asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(r) : "v"(a), "s"(c));
    return r;
#else
    return (a << 1) + c;
#endif
}

// x >= m ? x - m : x
DPF_HD u64 csub(u64 x, u64 m) {
    u64 t = x - m;
    return x >= m ? t : x;
}

// a + x 2^32 when the sum is known not to carry out of 64 bits: one 32-bit add into the high word
DPF_HD u64 add_hi32(u64 a, u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 hi = (u32)(a >> 32);
    asm("v_add_u32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(x));
    return ((u64)hi << 32) | (u32)a;
#else
    return a + ((u64)x << 32);
#endif
}

// -------------------------------------------------------------------------------------------------
struct alignas(16) TwShoup {
    u64 w, wsh;
};
// FoldArith twiddle.  With w = a + b 2^30 and w 2^32 mod q = a' + b' 2^30 (all four < 2^30):
//   w  = a  | b  << 32,   ws = a' | b' << 32      (tables.h: h_tw_fold)
struct alignas(16) TwFold {
    u64 w, ws;
};

// Policy flags read by ntt_core.h / the launchers:
//   kFold      - the pinned 2^60 - d primes: the fused kernels' lazy forms, exact division by N (FoldArith only);
//   kFoldCore  - fold butterflies and static bound plans inside a transform (FoldArith, FoldScaledArith);
//   kF64       - residues travel as IEEE doubles inside a transform (F64Arith).
// ntt_lc(c) is the LimbConst the transform's butterflies see (FoldScaledArith: the scaled modulus).
struct ShoupArith {
    typedef TwShoup Tw;
    static constexpr bool kFold = false, kFoldCore = false, kF64 = false;
    static DPF_HD const LimbConst& ntt_lc(const LimbConst& c) { return c; }
    // w*y mod q, result in [0, 2q), for ANY y < 2^64 (w < q, wsh = floor(w 2^64/q))
    static DPF_HD u64 mul_tw(u64 y, const Tw& t, const LimbConst& c) {
        u64 hi = mulhi64(y, t.wsh);
        return y * t.w - hi * c.q;
    }
    // a*b mod q, canonical, a,b < 2^64 with a*b < 2^124 (generic 128-bit Barrett, ratio = floor(2^128/q))
    static DPF_HD u64 mul_var(u64 a, u64 b, const LimbConst& c) {
        u64 z0 = a * b, z1 = mulhi64(a, b);
        u64 carry = mulhi64(z0, c.br_lo);
        u64 t1lo = z0 * c.br_hi, t1hi = mulhi64(z0, c.br_hi);
        u64 s = t1lo + carry;
        t1hi += (s < t1lo);
        u64 t2lo = z1 * c.br_lo, t2hi = mulhi64(z1, c.br_lo);
        u64 s2 = t2lo + s;
        t2hi += (s2 < t2lo);
        u64 qhat = z1 * c.br_hi + t1hi + t2hi;
        u64 r = z0 - qhat * c.q;
        r = csub(r, 2 * c.q);
        return csub(r, c.q);
    }
};

struct FoldArith {
    typedef TwFold Tw;
    static constexpr bool kFold = true, kFoldCore = true, kF64 = false;
    static DPF_HD const LimbConst& ntt_lc(const LimbConst& c) { return c; }
    // fold of the 124-bit product P = [p0, n0, r0, r1] (little-endian 32-bit words, r = P >> 64 < 2^60):
    // P = xl + xh 2^60  ==  xl + xh d (mod q), twice.  Result < 2^60 + 2^53 (< 2q), typically < q + 2^45.
    static DPF_HD u64 fold124(u32 p0, u32 n0, u64 r, u32 d) {
        u64 xl = (u64)p0 | ((u64)(n0 & 0x0fffffffu) << 32);
        u64 xh = (r << 4) | (n0 >> 28);                // P >> 60
        u64 A = mad32((u32)xh, d, xl);
        u64 B = mad32((u32)(xh >> 32), d, A >> 32);    // R = xl + xh*d = [A.lo, B.lo, B.hi] < 2^89
        u32 yh = (u32)(B >> 28);                       // R >> 60
        u64 yl = (u64)(u32)A | ((u64)((u32)B & 0x0fffffffu) << 32);
        return mad32(yh, d, yl);
    }
    // y*w mod q for y < 15 * 2^60 (kLimit in ntt_core.h: what a transform may hand to a dyadic product) and w < 2^60,
    // or y < 14 * 2^60 (kLimitPartner) and w only partially reduced, w < 2^60 + 2^29.
    // The middle column y0 w1 + y1 w0 + carry stays below 2^64 under either bound, so the four partial
    // products chain through the 64-bit addend of v_mad_u64_u32 with no carry fix-up, and the product stays < 2^124.
    static DPF_HD u64 mul60(u64 y, u64 w, u32 d) {
        DPFHE_EMU_ASSERT((y < (15ull << 60) && w < (1ull << 60)) || (y < (14ull << 60) && w < (1ull << 60) + (1ull << 29)));
        const u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
        u64 p = mad32(y0, w0, 0);
        u64 m = mad32(y0, w1, p >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(m));  // keep the chain as written: hipcc otherwise re-associates it and adds a 64-bit add
#endif
        u64 n = mad32(y1, w0, m);
        u64 r = mad32(y1, w1, n >> 32);
        return fold124((u32)p, (u32)n, r, d);
    }
    // same for ANY y < 2^64 (one extra carry add)
    static DPF_HD u64 mul60_full(u64 y, u64 w, u32 d) {
        const u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
        u64 p = mad32(y0, w0, 0);
        u64 m = mad32(y0, w1, p >> 32);
        u64 n = mad32(y1, w0, (u32)m);
        u64 r = mad32(y1, w1, (n >> 32) + (m >> 32));
        return fold124((u32)p, (u32)n, r, d);
    }
    // any x < 2^64  ->  x mod q representative < 2^60 + 16 d
    static DPF_HD u64 reduce(u64 x, const LimbConst& c) {
        return mad32((u32)(x >> 60), (u32)c.d, x & 0x0fffffffffffffffull);
    }
    // y*w mod q for ANY 64-bit y; result < 2^60 + 13 d.
    //   y w  ==  y0 w + y1 (w 2^32 mod q)  =  L + H 2^30     with  L = y0 a + y1 a',  H = y0 b + y1 b'  (< 2^63)
    //        ==  L + H.lo 2^30 + H.hi 4d                      (2^62 == 4d)            (< 3 2^62 + 2^57 < 2^64)
    // Seven chained multiply-adds whose addends are all whole 64-bit results (no {hi,0} pair to build, no
    // shifts), then the 3-instruction reduce.  mul60 below needs 8 more ALU ops for the same job.
    static DPF_HD u64 mul_tw(u64 y, const Tw& t, const LimbConst& c) {
        const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
        const u32 a = (u32)t.w, b = (u32)(t.w >> 32), as = (u32)t.ws, bs = (u32)(t.ws >> 32);
        DPFHE_EMU_ASSERT(((a | b | as | bs) >> 30) == 0);
        const u64 H = mad32(y1, bs, mad32(y0, b, 0));
        u32 two30 = 1u << 30;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("s_mov_b32 %0, 0x40000000" : "=s"(two30));  // opaque: hipcc would turn H.lo * 2^30 into shift + zero-extend + add
#endif
        const u64 L = mad32(y1, as, mad32(y0, a, mad32((u32)H, two30, 0)));
        const u64 R = mad32((u32)(H >> 32), 4 * (u32)c.d, L);
        return reduce(R, c);
    }
    // addend + y*w mod q in ONE fold: the first multiply-add of mul_tw's chain has a free 64-bit addend, so the butterfly
    // sum x' = a + y w costs nothing beyond the product and comes out already reduced (< 2^60 + 16 d).
    // Precondition (units of 2^60):  addend + y/4 < 8 - 2^-6   (L = H.lo 2^30 + y0 a + y1 a' + addend must stay < 2^64 - 2^54:
    // H.lo 2^30 < 4, y0 a < 4, y1 a' < y/4);  any 64-bit y allows addend < 2^62 - 2^54.  ntt_core.h make_ctf_plan tracks it.
    static DPF_HD u64 mul_tw_add(u64 y, const Tw& t, const LimbConst& c, u64 addend) {
        const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
        const u32 a = (u32)t.w, b = (u32)(t.w >> 32), as = (u32)t.ws, bs = (u32)(t.ws >> 32);
        DPFHE_EMU_ASSERT(((a | b | as | bs) >> 30) == 0);
        DPFHE_EMU_ASSERT((unsigned __int128)addend + ((unsigned __int128)y0 << 30) + ((unsigned __int128)y1 << 30) + (1ull << 62) + (1ull << 54) < ((unsigned __int128)1 << 64));
        const u64 H = mad32(y1, bs, mad32(y0, b, 0));
        u32 two30 = 1u << 30;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("s_mov_b32 %0, 0x40000000" : "=s"(two30));
#endif
        const u64 L = mad32(y1, as, mad32(y0, a, mad32((u32)H, two30, addend)));
        const u64 R = mad32((u32)(H >> 32), 4 * (u32)c.d, L);
        return reduce(R, c);
    }
    // ---- variable x variable products through the twiddle chain (round 6) ------------------------------------------------------
    // One factor b (reduced: b < 2^60 + 2^29) is turned into a twiddle on the fly, then every product with it costs mul_tw's 9 instructions
    // instead of mul60's ~25 (hipcc builds mul60's {hi, 0} addends and fold124's shifted words with v_mov / v_alignbit), and a sum of two
    // products comes for free through the chain's addend.  8 instructions per factor, shared by all its products:
    //   halves of b:               a = b mod 2^30,  bh = b >> 30 (<= 2^30)
    //   companion b 2^32 mod q:    b 2^32 = (b mod 2^29) 2^32 + (b >> 29) 2^61  ==  (b mod 2^29) 2^32 + (b >> 29) 2d   (< 2^61 + 2^57), halves as', bh' (< 2^31 + 2^27)
    // The chain with these wider halves, for ANY 64-bit y and addend < 2^62 - 2^58:
    //   H = y0 bh + y1 bh' < 2^62 + 2^63 + 2^59;   L = H.lo 2^30 + y0 a + y1 as' + addend < 3 2^62 + addend;   R = L + H.hi 4d < 2^64  (H.hi 4d < 2^58)
    // (mul_ptw_add checks them exactly under the emulator).
    static DPF_HD Tw prod_tw(u64 b, const LimbConst& c) {
        DPFHE_EMU_ASSERT(b < (1ull << 60) + (1ull << 29));
        const u32 lo = (u32)b;
        u32 a = lo & 0x3fffffffu, bh = (u32)(b >> 30), t29 = (u32)(b >> 29), bl = lo & 0x1fffffffu;
#if defined(__HIP_DEVICE_COMPILE__)
        // opaque 32-bit halves: left visible, hipcc reasons about the 64-bit shifts behind them and multiplies by 34-bit values in two steps,
        // takes the companion's low word from a separate v_mul_lo_u32, builds {0, bl} with a move (35 instead of 8 instructions)
        asm("" : "+v"(a), "+v"(bh), "+v"(t29), "+v"(bl));
#endif
        const u64 bs = add_hi32(mad32(t29, 2 * (u32)c.d, 0), bl);     // (b >> 29) 2d + (b mod 2^29) 2^32 < 2^57 + 2^61: no carry out of the high word
        u32 as = (u32)bs & 0x3fffffffu, bsh = (u32)(bs >> 30);
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(as), "+v"(bsh));
#endif
        Tw t;
        t.w = (u64)a | ((u64)bh << 32);
        t.ws = (u64)as | ((u64)bsh << 32);
        return t;
    }
    // (emulator only) the chain below in exact arithmetic: neither H nor R = L + H.hi 4d leaves 64 bits
    static DPF_HD bool ptw_fits(u32 y0, u32 y1, u32 a, u32 b, u32 as, u32 bs, u64 addend, u32 d) {
        typedef unsigned __int128 u128;
        const u128 H = (u128)y0 * b + (u128)y1 * bs;
        const u128 L = (u128)((u32)(u64)H) * (1u << 30) + (u128)y0 * a + (u128)y1 * as + addend;
        return (H >> 64) == 0 && ((L + (u128)(u32)((u64)H >> 32) * (4 * d)) >> 64) == 0;
    }
    // addend + y b mod q for the twiddle prod_tw made of b; result < 2^60 + 16 d
    static DPF_HD u64 mul_ptw_add(u64 y, const Tw& t, const LimbConst& c, u64 addend) {
        const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
        const u32 a = (u32)t.w, b = (u32)(t.w >> 32), as = (u32)t.ws, bs = (u32)(t.ws >> 32);
        DPFHE_EMU_ASSERT(ptw_fits(y0, y1, a, b, as, bs, addend, (u32)c.d));
        const u64 H = mad32(y1, bs, mad32(y0, b, 0));
        u32 two30 = 1u << 30;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("s_mov_b32 %0, 0x40000000" : "=s"(two30));
#endif
        const u64 L = mad32(y1, as, mad32(y0, a, mad32((u32)H, two30, addend)));
        const u64 R = mad32((u32)(H >> 32), 4 * (u32)c.d, L);
        return reduce(R, c);
    }
    static DPF_HD u64 mul_ptw(u64 y, const Tw& t, const LimbConst& c) { return mul_ptw_add(y, t, c, 0); }
    // acc + x e mod q for a key word e < 2^60 + 2^29 (canonical) and ANY 64-bit x (a lazy forward output), acc < 2^62 - 2^58: 17 instructions, result
    // < 2^60 + 16 d - the accumulator of a key inner product stays ONE reduced word (mul60 + lazy sums: ~26 per term and a reduction every 13 terms)
    static DPF_HD u64 mac_var(u64 acc, u64 x, u64 e, const LimbConst& c) { return mul_ptw_add(x, prod_tw(e, c), c, acc); }
    // x N^-1 mod q for N = 2^n by EXACT DIVISION instead of a twiddle product (round 4; 6 instructions against mul_tw's 9):
    // q = 1 (mod 2N), so with m = x mod N the number x - m q is divisible by N, and
    //     (x - m q) / N = (x + m d) / N - m 2^(60-n);      adding q keeps it positive:   y = ((x + m d) >> n) + (q - (m << (60 - n))).
    // (m << (60 - n)) has no bits below 2^32 for n <= 28, so the subtraction is one 32-bit op on the high word.
    // Precondition: x + m d < 2^64 (any x < 2^64 - 2^(n+24)); result < q + 2^(64-n), to be canonicalised by canon_small.
    static DPF_HD u64 mul_ninv(u64 x, const LimbConst& c, int n) {
        const u32 m = (u32)x & ((1u << n) - 1u);
        DPFHE_EMU_ASSERT((unsigned __int128)x + (unsigned __int128)m * c.d < ((unsigned __int128)1 << 64));
        const u64 t = mad32(m, (u32)c.d, x);
        const u32 vhi = (u32)(c.q >> 32) - (m << (28 - n));
        return (t >> n) + (((u64)vhi << 32) | (u32)c.q);
    }
    // r < 2^60 + 2^59  ->  r mod q in [0, q), without compare/select: r >= q  <=>  r + d >= 2^60, and then
    // r - q = (r + d) - 2^60.  4 instructions (add, shift, multiply-add, and) against 5 for csub.
    static DPF_HD u64 canon_small(u64 r, const LimbConst& c) {
        DPFHE_EMU_ASSERT(r < (3ull << 59));
        const u32 k = (u32)((r + c.d) >> 60);
        return mad32(k, (u32)c.d, r) & 0x0fffffffffffffffull;
    }
    static DPF_HD u64 canon(u64 x, const LimbConst& c) { return canon_small(reduce(x, c), c); }
    // a*b mod q, canonical; a < 2^64, b < 2^60
    static DPF_HD u64 mul_var(u64 a, u64 b, const LimbConst& c) { return csub(mul60(a, b, (u32)c.d), c.q); }

    // ---- lazy dot products of CANONICAL residues (the plaintext matrix-vector products, kernels_misc.h) -------------------
    // Both factors are split at bit 30 (a = a0 + a1 2^30, all four halves < 2^30), so every partial product is < 2^60 and
    // the three columns  S0 = sum a0 b0,  S1 = sum (a0 b1 + a1 b0),  S2 = sum a1 b1  are plain 64-bit multiply-add chains:
    // FOUR v_mad_u64_u32 per multiply-accumulate and nothing else (the 128-bit accumulators this replaces cost 21 VALU
    // instructions per term: 4 multiply-adds, 2 multiplies, 7 moves, 4 64-bit adds, compare/select/add3).  After at most
    // kDot30Period terms (S1 < 16 * 2^60) the columns are folded into one reduced word:
    //   S0 + S1 2^30 + S2 2^60  ==  S0 + (S1 mod 2^30) 2^30 + (S2 + (S1 >> 30)) d        (2^60 == d)
    // 18 instructions per fold (4 multiply-adds), i.e. 2.25 per term.
    struct Half30 { u32 lo, hi; };
    struct Dot30 { u64 s0, s1, s2; };
    static constexpr int kDot30Period = 8;
    static DPF_HD Half30 split30(u64 v) {
        DPFHE_EMU_ASSERT(v < (1ull << 60));
        return Half30{(u32)v & 0x3fffffffu, (u32)(v >> 30)};
    }
    static DPF_HD void dot30_mac(Dot30& s, const Half30& a, const Half30& b) {
        DPFHE_EMU_ASSERT(s.s0 < (15ull << 60) && s.s1 < (14ull << 60) && s.s2 < (15ull << 60));
        s.s0 = mad32(a.lo, b.lo, s.s0);
        u64 m = mad32(a.lo, b.hi, s.s1);
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(m));   // keep the chain as written: hipcc otherwise sums the two cross products first and spends a 64-bit add on s1
#endif
        s.s1 = mad32(a.hi, b.lo, m);
        s.s2 = mad32(a.hi, b.hi, s.s2);
    }
    // folds the columns and a running reduced word r (< 2^60 + 2^29) into a new running word < 2^60 + 16 d.  Written on 32-bit halves so
    // that every step is ONE instruction: the shifted addends are built by multiply-adds with 2^30 and by an add into the high word
    // (left as 64-bit shifts hipcc materialises {0, x} register pairs: 23 instructions instead of 16).
    static DPF_HD u64 dot30_fold(const Dot30& s, u64 r, const LimbConst& c) {
        DPFHE_EMU_ASSERT(r < (1ull << 60) + (1ull << 29));
        const u32 d = (u32)c.d;
        u32 two30 = 1u << 30;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("s_mov_b32 %0, 0x40000000" : "=s"(two30));                         // opaque: x * 2^30 + y stays one v_mad_u64_u32
#endif
        const u64 r0 = reduce(s.s0, c) + r;                                     // < 2^61 + 2^30            (r is 0 in the matvec kernels: folded away)
        const u64 U = s.s2 + (s.s1 >> 30);                                      // < 2^63 + 2^34
        const u64 V = mad32((u32)s.s1 & 0x3fffffffu, two30, r0);                // + (S1 mod 2^30) 2^30:  < 3 * 2^60 + 2^30
        const u64 T = mad32((u32)(U >> 32), d, 0);                              // U.hi d < 2^56;  U.hi d 2^32 = (T mod 2^28) 2^32 + (T >> 28) 2^60
        u64 A = mad32((u32)U, d, V);                                            // < 3 * 2^60 + 2^57
        A = mad32((u32)(T >> 28), d, A);                                        // + < 2^52
        A = add_hi32(A, (u32)T & 0x0fffffffu);                                  // + (T mod 2^28) 2^32 < 2^60: A < 2^63, no carry out of the high word
        return reduce(A, c);
    }
    // the same without a running word and WITHOUT the reduction of S0 in front, for columns whose S0 is known to be small: S0 < 12 * 2^60 (a period that
    // started from a folded word < 2^60 + 16 d and added at most 8 products, or up to 11 products from zero).  Then
    //   S0 + (S1 mod 2^30) 2^30 + U.lo d + (T >> 28) d + (T mod 2^28) 2^32  <  12 + 1 + 2^-4 + 2^-9 + 1  <  14.1 * 2^60: no wrap, three instructions less.
    static DPF_HD u64 dot30_fold0(const Dot30& s, const LimbConst& c) {
        DPFHE_EMU_ASSERT(s.s0 < (12ull << 60) && s.s2 < (1ull << 63));
        const u32 d = (u32)c.d;
        u32 two30 = 1u << 30;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("s_mov_b32 %0, 0x40000000" : "=s"(two30));
#endif
        const u64 U = s.s2 + (s.s1 >> 30);
        const u64 T = mad32((u32)(U >> 32), d, 0);
        u64 A = mad32((u32)s.s1 & 0x3fffffffu, two30, s.s0);
        A = mad32((u32)U, d, A);
        A = mad32((u32)(T >> 28), d, A);
        A = add_hi32(A, (u32)T & 0x0fffffffu);
        return reduce(A, c);
    }
};

// -------------------------------------------------------------------------------------------------
// FoldScaledArith - primes q = 2^k - d0 with 48 <= k < 60 and d0 2^(60-k) < 2^24 (round 6).
// With s = 2^(60-k) the residue x is carried as s x modulo q' = s q = 2^60 - d, d = d0 s:  s x mod s q = s (x mod q), so the
// scaled words are an exact image of Z_q, q' has the shape FoldArith wants (2^60 = d mod q', d < 2^24) and every fold
// reduction, bound plan and butterfly of FoldArith applies unchanged with (q', d).  Twiddles stay UNSCALED (w < q; the
// companion w 2^32 is taken mod q': tables.h h_tw_fold_scaled), so (s x) w = s (x w): one shift when a word enters a
// transform, one when it leaves.  N^-1 is a twiddle product (the exact division of FoldArith::mul_ninv needs q' = 1 mod 2N).
// LimbConst of such a limb: q = the TRUE prime, d = d0 s, pad1 = 60 - k; Barrett constants of the true prime.
// At the kernels' level the policy is "generic": canonical words of the true prime between transforms (kFold = false).
struct FoldScaledArith {
    typedef TwFold Tw;
    static constexpr bool kFold = false, kFoldCore = true, kF64 = false;
    static DPF_HD LimbConst ntt_lc(const LimbConst& c) {
        LimbConst r = c;
        r.q = c.q << c.pad1;
        return r;
    }
    static DPF_HD u64 enter(u64 x, const LimbConst& c) { return x << c.pad1; }                       // canonical x < q  ->  s x < q'
    static DPF_HD u64 leave(u64 x, const LimbConst& c) { return x >> c.pad1; }                       // canonical mod q' (a multiple of s)  ->  x
    static DPF_HD u64 mul_var(u64 a, u64 b, const LimbConst& c) {                                    // canonical a, b < q
        const u64 r = FoldArith::mul60(a << c.pad1, b, (u32)c.d);                                    // s a b mod q', < q' + 2^53
        return csub(r, c.q << c.pad1) >> c.pad1;
    }
};

// -------------------------------------------------------------------------------------------------
// F64Arith - primes q < 2^47, F64WideArith - primes q < 2^50 (round 6): inside a transform a residue is an IEEE double holding a (signed) integer, and the
// modular product is the error-free FMA sequence
//     p = y w;  e = fma(y, w, -p);  h = rint(y (w/q));  r = fma(-h, q, p);  t = r + e            ( = y w - h q exactly )
// - 6 full-rate FP64 instructions, no integer multiply.  Exactness (|y| < 2^51, 0 <= w < q < 2^47, wq = fl(w / q)):
//   * y wq differs from y w / q by at most |y| 2^-52 <= 1/2, so |y w / q - h| <= 1 and |t| <= q;
//   * p and h q are integers and |r| = |t - e| <= q + ulp(p)/2 < 2^53, so the second fma is exact; so is r + e.
//   (the same two lines hold for q < 2^50: ulp(p) / 2 <= 2^48.)
// Butterflies are plain signed additions: a forward (Cooley-Tukey) word grows by at most q per stage, 1 + log2 N <= 16 q <= 2^51
// at the last stage - no reduction inside a forward transform at all; the inverse (Gentleman-Sande) sums double per stage and are
// reduced where the static plan (ntt_core.h make_gs_plan, cap 16 q) says so (3 instructions).  The rounding to an integer is the
// magic-constant addition (fma(y, wq, 1.5 2^52) - 1.5 2^52): no dependence on the rate of v_rndne_f64.
// At the kernels' level the policy is "generic": canonical u64 words between transforms.
// LimbConst of such a limb (the array behind DevTables<F64Arith>::lc): q = the prime, ninv / ninv_sh REINTERPRETED as the doubles
// q and fl(1 / q) (N^-1 reaches the transform as a twiddle), d = 0; Barrett constants as usual.
struct alignas(16) TwF64 {
    double w, wq;   // w and fl(w / q)
};
// MAXBITS: the widest prime the instantiation takes.  A product needs |y| < 2^51, i.e. |y| < kCap q with kCap = 2^(51 - MAXBITS) - 16 at 47 bits (no
// reduction inside a forward transform), 2 at 50 bits (F64WideArith: ntt_core.h's static plans reduce most multiplied words first, 3 instructions each:
// about the fold arithmetic's instruction count, on full-rate FP64 instructions, for primes no other fast class takes).
template <int MAXBITS>
struct F64ArithT {
    typedef TwF64 Tw;
    static constexpr bool kFold = false, kFoldCore = false, kF64 = true;
    static constexpr int kMaxBits = MAXBITS;
    static_assert(MAXBITS >= 20 && MAXBITS <= 50, "the error-free product needs q + ulp(y w) / 2 < 2^53 and a quotient estimate within 1");
    static constexpr int kCap = 1 << (51 - MAXBITS);   // |y| < kCap q  =>  |y| < 2^51
    static DPF_HD const LimbConst& ntt_lc(const LimbConst& c) { return c; }
    static DPF_HD double f(u64 x) { return __builtin_bit_cast(double, x); }
    static DPF_HD u64 b(double x) { return __builtin_bit_cast(u64, x); }
    static DPF_HD double qd(const LimbConst& c) { return f(c.ninv); }
    static DPF_HD double qinv(const LimbConst& c) { return f(c.ninv_sh); }
    static constexpr double kTwo52 = 4503599627370496.0, kMagic = 6755399441055744.0;   // 2^52, 1.5 2^52
    // integer x < 2^52 -> double, and back for an integer-valued 0 <= r < 2^52: two instructions each
    static DPF_HD double to_f(u64 x) { return f(x | 0x4330000000000000ull) - kTwo52; }
    static DPF_HD u64 from_f(double r) { return b(r + kTwo52) & 0x000fffffffffffffull; }
    static DPF_HD double rint_mul(double y, double k) { return __builtin_fma(y, k, kMagic) - kMagic; }   // rint(y k), |y k| < 2^51
    static DPF_HD double mulmod(double y, double w, double wq, double q) {
        DPFHE_EMU_ASSERT(y > -2251799813685248.0 && y < 2251799813685248.0);
        const double p = y * w;
        const double e = __builtin_fma(y, w, -p);
        const double h = rint_mul(y, wq);
        const double r = __builtin_fma(-h, q, p);
        return r + e;
    }
    // any |x| < 2^52  ->  x mod q in [-q/2 - 1, q/2 + 1]
    static DPF_HD double reduce(double x, double q, double qi) { return __builtin_fma(-rint_mul(x, qi), q, x); }
    static DPF_HD double canon(double x, double q, double qi) {
        const double r = reduce(x, q, qi);
        return r < 0.0 ? r + q : r;
    }
    static DPF_HD u64 enter(u64 x, const LimbConst&) { return b(to_f(x)); }
    static DPF_HD u64 leave(u64 x, const LimbConst& c) { return from_f(canon(f(x), qd(c), qinv(c))); }
    // a*b mod q, canonical in and out (the dyadic products of the generic fused multiply)
    static DPF_HD u64 mul_var(u64 a, u64 bb, const LimbConst& c) {
        const double q = qd(c), qi = qinv(c), x = to_f(a), y = to_f(bb);
        const double p = x * y;
        const double e = __builtin_fma(x, y, -p);
        const double h = rint_mul(p, qi);               // p / q < 2^47: the estimate is within 2^-5 of the quotient
        const double r = __builtin_fma(-h, q, p) + e;   // in [-q/2 - 1, q/2 + 1]
        return from_f(r < 0.0 ? r + q : r);
    }
};
typedef F64ArithT<47> F64Arith;
typedef F64ArithT<50> F64WideArith;

// canonical add / sub / negate (inputs canonical)
DPF_HD u64 add_mod(u64 a, u64 b, u64 q) { return csub(a + b, q); }
DPF_HD u64 sub_mod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
DPF_HD u64 neg_mod(u64 a, u64 q) { return a ? q - a : 0; }

// two words moved as one 16-byte access
struct __attribute__((aligned(16))) U64x2 {
    u64 a, b;
};

}  // namespace dpfhe
