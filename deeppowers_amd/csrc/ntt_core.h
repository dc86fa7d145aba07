// ntt_core.h - the register/LDS-tiled negacyclic NTT of one residue polynomial (host + device).
//
// SURVEY.md section 8(a) rows A1/A2 (no reference counterpart - section 0).  One workgroup owns one
// residue polynomial.  Each of T = N/E threads holds E = 2^LOGE coefficients in registers and runs
// up to LOGE radix-2 stages per "phase" without touching memory; between phases the polynomial is
// re-tiled through one padded LDS buffer.  N=4096: E=16, 256 threads, phases of 4+4+4 stages, two
// LDS round trips, one HBM read and one HBM write in total.
//
// The forward transform is Cooley-Tukey (natural in -> bit-reversed out), the inverse is
// Gentleman-Sande (bit-reversed in -> natural out, N^-1 folded into the last stage); the inverse
// walks the same phase list backwards, so the forward OUTPUT register mapping equals the inverse
// INPUT mapping and a fused ct x ct multiply never leaves registers between them.
//
// Only the FIRST exchange of the forward transform (last of the inverse) moves words between waves: in every window
// starting at or below lane bit 6 a wave owns the same 64 E consecutive coefficients, so all later exchanges are
// transpositions inside a wave.  Each wave therefore owns a private, padded LDS region (Geo::kWaveStride words) and the
// wave-local exchanges need no workgroup barrier - one s_barrier per transform instead of three or four.
//
// Every function takes `tid` explicitly and LDS as a plain pointer, so tools/emulate.cpp can run
// the very same code thread-by-thread on the CPU and compare with the oracle.
#pragma once
#include "modarith.h"

namespace dpfhe {

// tools/emulate.cpp defines DPFHE_EMU_CHECK to count 64-bit wrap-arounds in the lazy arithmetic
#if defined(DPFHE_EMU_CHECK) && !defined(__HIPCC__)
inline u64 chk_add(u64 a, u64 b) { if ((unsigned __int128)a + b >> 64) ++g_emu_overflows; return a + b; }
inline u64 chk_sub_add(u64 a, u64 b, u64 off) {  // a - b + off must stay in [0, 2^64)
    __int128 v = (__int128)a - (__int128)b + (__int128)off;
    if (v < 0 || (v >> 64)) ++g_emu_overflows;
    return a - b + off;
}
inline u64 chk_shl1_add_sub(u64 a, u64 c, u64 s) {  // 2a + c - s must stay in [0, 2^64)
    __int128 v = 2 * (__int128)a + (__int128)c - (__int128)s;
    if (v < 0 || (v >> 64) || ((unsigned __int128)2 * a + c) >> 64) ++g_emu_overflows;
    return shl1_add(a, c) - s;
}
#else
DPF_HD u64 chk_shl1_add_sub(u64 a, u64 c, u64 s) { return shl1_add(a, c) - s; }
DPF_HD u64 chk_add(u64 a, u64 b) { return a + b; }
DPF_HD u64 chk_sub_add(u64 a, u64 b, u64 off) { return a - b + off; }
#endif

// ------------------------------------------------------------------------------------------------
// geometry
// ------------------------------------------------------------------------------------------------
struct Phase {
    int c;   // window start: the thread-local index k occupies bits [c, c+LOGE) of the coefficient index
    int b;   // butterfly bits of this phase are [b, b+r)  (a sub-range of the window)
    int r;   // number of radix-2 stages
};

// forward phase p of geometry (logn, loge) handles bits [b, b+r) going DOWN from the top; balanced split of logn
constexpr int geo_nph(int logn, int loge) { return (logn + loge - 1) / loge; }
constexpr Phase geo_phase(int logn, int loge, int p) {
    int rem = logn, top = logn, r = 0;
    for (int i = 0; i <= p; ++i) {
        int left = geo_nph(logn, loge) - i;
        r = (rem + left - 1) / left;
        top = rem;
        rem -= r;
    }
    int b = top - r;
    int c = b < (logn - loge) ? b : (logn - loge);
    return Phase{c, b, r};
}
// stages (bit positions [0, n)) of the window-0 phase whose twiddle-table section is stored transposed (tw_index)
constexpr int geo_perm_stages(int logn, int loge) {
    return geo_nph(logn, loge) >= 2 ? geo_phase(logn, loge, geo_nph(logn, loge) - 1).r : 0;
}

template <int LOGN_, int LOGE_>
struct Geo {
    static constexpr int LOGN = LOGN_, LOGE = LOGE_;
    static constexpr int N = 1 << LOGN, E = 1 << LOGE, T = N / E;
    static constexpr int NPH = geo_nph(LOGN, LOGE);
    static_assert(LOGN >= LOGE && LOGE >= 1, "bad geometry");

    static constexpr Phase phase(int p) { return geo_phase(LOGN, LOGE, p); }
    static constexpr int kPermStages = geo_perm_stages(LOGN, LOGE);
    // coefficient index of local element k of thread tid in a phase with window start c
    static DPF_HD int index(int c, int tid, int k) {
        if ((1 << c) >= T) return (k << c) | tid;  // top window: tid < T = 2^c, no split (and no mask for the compiler to fold)
        return ((tid >> c) << (c + LOGE)) | (k << c) | (tid & ((1 << c) - 1));
    }
    // ---- LDS layout: one private region per wave --------------------------------------------------------------
    // In any window c <= 6 the 64 threads of wave w hold coefficients [w 64E, (w+1) 64E), so the exchange between
    // windows clo < chi <= 6 only permutes words inside a wave.
    static constexpr int LOGW = 6 + LOGE, kWaveWords = 1 << LOGW;
    static constexpr int kWaves = T >= 64 ? T / 64 : 1;
    static constexpr bool exch_wave_local(int p) { return T <= 64 || phase(p).c <= 6; }   // exchange between phases p and p + 1
    static constexpr int highest_cross_wave_exchange() {
        int h = -1;
        for (int p = 0; p + 1 < NPH; ++p) if (!exch_wave_local(p)) h = p;
        return h;
    }
    // Padding inside a wave's region.  Bijective for any choice (monotone); chosen so that both access patterns of the
    // N=4096 / N=8192 kernels are bank-conflict free:
    //  * window 0 side: a thread touches E consecutive words with 16-byte accesses; a lane stride of
    //    E+2 words spreads a ds_read/write_b128 lane group over all 16-byte slots;
    //  * window chi side: lanes are 2^chi-word runs at a 2^(chi+LOGE)-word stride; +16 words per
    //    stride puts the two runs of a 32-lane ds_read_b64 group on opposite halves of the bank row.
    static constexpr int wave_extent() {   // largest padded local address over all wave-local exchanges, both directions, + 1
        int mx = kWaveWords;
        const int last = (N < kWaveWords ? N : kWaveWords) - 1;
        for (int p = 0; p + 1 < NPH; ++p) {
            if (!exch_wave_local(p)) continue;
            const int chi = phase(p).c, clo = phase(p + 1).c;
            int a = last;
            if (clo == 0) a += (last >> LOGE) << 1;
            a += (last >> (chi + LOGE)) << 4;    // inverse direction (a superset of the forward padding)
            if (a + 1 > mx) mx = a + 1;
        }
        if ((64 * (E + 2)) > mx) mx = 64 * (E + 2);   // row layout of load_bot_lds / store_bot_lds
        return mx;
    }
    // region stride: a multiple of 32 words (256 bytes = one bank row), so the all-to-all exchange - which only adds the
    // region offset to the plain index - keeps the bank pattern of an unpadded buffer
    static constexpr int kWaveStride = (wave_extent() + 31) & ~31;
    // lds_addr is ADDITIVE over disjoint bit fields of j (shifts, masks and constant multiples only), which is what lets
    // NttBody split every address into a per-thread base and a compile-time offset per local element.
    template <int CLO, int CHI, bool FWD>
    static constexpr int lds_addr(int j) {
        const int w = j >> LOGW, jl = j & (kWaveWords - 1);
        if (CHI <= 6 || T <= 64) {   // wave-local exchange
            int a = jl;
            if (CLO == 0) a += (jl >> LOGE) << 1;
            if (CLO != 0 || !FWD) a += (jl >> (CHI + LOGE)) << 4;
            return w * kWaveStride + a;
        }
        static_assert(CHI <= 6 || CLO != 0, "the all-to-all exchange is never the one into window 0");
        return w * kWaveStride + jl;   // all-to-all (top window -> next): plain index inside each region
    }
    static DPF_HD int lds_row(int tid) { return (tid >> 6) * kWaveStride + (tid & 63) * (E + 2); }   // private row of a thread (kLdsIO)
    static constexpr int lds_words() { return kWaves * kWaveStride; }
};

// twiddle table index of the butterfly whose lower element is local k, at global bit position `pos`
// (distance 2^pos): table[(N >> (pos+1)) + (j >> (pos+1))], split into thread part + constant part.
// In the window-0 phase a thread needs 2^(LOGE-1-pos) consecutive entries per stage, so one wave-wide fetch would
// touch up to 64 different 128-byte lines; the device tables are therefore stored with those stages transposed
// ([entry-of-thread][thread], tables.h permute_window0) and every fetch is lane-contiguous.
// The index is split into a workgroup-uniform part and a per-thread part so that the fetch uses the scalar-base +
// 32-bit-lane-offset addressing form (no per-entry 64-bit address arithmetic, no extra address registers).
template <class G>
DPF_HD int tw_index_uniform(int c, int k, int pos) {
    const int lb = pos - c;
    if (G::kPermStages > 0 && c == 0) return (1 << (G::LOGN - 1 - pos)) + (k >> (lb + 1)) * G::T;
    return (1 << (G::LOGN - 1 - pos)) + (k >> (lb + 1));
}
template <class G>
DPF_HD unsigned tw_index_thread(int c, int th /* = tid >> c */, int pos) {
    const int lb = pos - c;
    if (G::kPermStages > 0 && c == 0) return (unsigned)th;
    return (unsigned)th << (G::LOGE - 1 - lb);
}
template <class G>
DPF_HD int tw_index(int c, int th, int k, int pos) { return tw_index_uniform<G>(c, k, pos) + (int)tw_index_thread<G>(c, th, pos); }
// tid >> c, forced to the constant 0 for the top window (all its twiddles are workgroup-uniform and
// are fetched with scalar loads)
template <class G>
DPF_HD int tid_high(int c, int tid) { return (c + G::LOGE >= G::LOGN) ? 0 : (tid >> c); }

// ------------------------------------------------------------------------------------------------
// static bound plans (FoldArith).  Bounds are in units of q/1024.  A value with bound B is < B q/1024.
// FoldArith::mul_tw takes any 64-bit word, so inside a transform a word only has to fit (kWord); words that
// leave a transform unreduced feed FoldArith::mul60 and stay below 15 q (kLimit).
// ------------------------------------------------------------------------------------------------
constexpr int kUnit = 1024;
constexpr int kMulB = kUnit + 9;    // mul60 output  < q + 2^53           (q > 2^59.99)
constexpr int kRedB = kUnit + 1;    // reduce output < 2^60 + 16d
constexpr int kTwB = kRedB;         // mul_tw output < 2^60 + 13d
constexpr int kWord = 16 * kUnit;   // 16 q < 2^64: a lazy sum must not wrap
constexpr int kLimit = 15 * kUnit;  // < 15 q < 15 * 2^60: the precondition of FoldArith::mul60's first operand
constexpr int kLimitPartner = 14 * kUnit;  // ... when its second operand is only partially reduced (< 2^60 + 2^29)

template <int LOGE>
struct GsPlan {  // one Gentleman-Sande phase on E local elements, up to LOGE stages
    bool red[LOGE][1 << LOGE];   // reduce element k before stage u
    int off[LOGE][1 << LOGE];    // butterfly with lower element k at stage u: y' = (x - y + off q) w
    bool red_end[1 << LOGE];     // reduce element k after the last stage (normalise to <= out bound)
    int out_bound;
};

// lb0 = local bit of the first stage, r stages (local bits ascending), every input < in_bound.
// last_all_mul: the phase ends with the transform's LAST stage, whose sums are multiplied by N^-1 too - by a twiddle product (ninv_shift = 0:
// generic primes, bound kTwB) or by FoldArith::mul_ninv's exact division by 2^ninv_shift (result < q + s / 2^ninv_shift + 1 for a sum s).
template <int LOGE>
constexpr GsPlan<LOGE> make_gs_plan(int lb0, int r, int in_bound, int out_bound, bool last_all_mul, int ninv_shift = 0) {
    GsPlan<LOGE> p{};
    constexpr int E = 1 << LOGE;
    int bnd[E] = {};
    for (int k = 0; k < E; ++k) bnd[k] = in_bound;
    for (int u = 0; u < r; ++u) {
        const int bit = 1 << (lb0 + u);
        for (int k = 0; k < E; ++k) {
            if (k & bit) continue;
            int bx = bnd[k], by = bnd[k | bit];
            int offq = (by + kUnit - 1) / kUnit;
            // (last inverse stage, FoldArith::mul_ninv: the sum x' = a + b is divided by N exactly and needs a little headroom below 2^64)
            const int sum_cap = (last_all_mul && u == r - 1) ? kWord - kUnit / 8 : kWord;
            if (bx + by > sum_cap || bx + offq * kUnit > kWord) {
                if (bx > kRedB) { p.red[u][k] = true; bx = kRedB; }
                if (by > kRedB) { p.red[u][k | bit] = true; by = kRedB; }
                offq = (by + kUnit - 1) / kUnit;
            }
            p.off[u][k] = offq;
            const bool final_stage = last_all_mul && (u == r - 1);
            // last inverse stage multiplies x' by N^-1 too: a twiddle product, or the exact division (x' < q + (a + b) / N + 1)
            bnd[k] = !final_stage ? bx + by : (ninv_shift ? kUnit + ((bx + by) >> ninv_shift) + 2 : kTwB);
            bnd[k | bit] = kTwB;
        }
    }
    for (int k = 0; k < E; ++k) {
        p.red_end[k] = bnd[k] > out_bound;
        if (p.red_end[k]) bnd[k] = kRedB;
    }
    p.out_bound = out_bound;
    return p;
}

// Cooley-Tukey with the FUSED sum (FoldArith::mul_tw_add):  x' = reduce(a + w y) comes out of the product's own
// multiply-add chain already reduced, and y' = a - w y = 2a + 2q - x' costs one v_lshl_add_u64 and one 64-bit subtract:
// 12 VALU per butterfly instead of 13, and no separate reductions of the sums.  Bounds (units of q/1024): x' -> kRedB,
// y' -> 2 A + 2048, so a word is at 1, 4 or 10 q; the fused chain needs  A + Y/4 <= kFuseLimit  (modarith.h), which only
// fails for an `a` at 10 q: such words (a quarter of the butterflies from the third stage of a phase on) are reduced
// first.  Bounds are tracked per local element inside a phase and handed over as one uniform bound at the LDS exchanges
// (the element <-> thread mapping changes there), capped at kCtfMid = 4 q by reducing the few words above it.
constexpr int kFuseLimit = 8 * kUnit - 16;   // addend + y/4 < 8 * 2^60 - 2^54
constexpr int kCtfMid = 2 * kRedB + 2 * kUnit;

template <int LOGE>
struct CtfPlan {
    bool red_a[LOGE][1 << LOGE];   // reduce the upper word a of the butterfly with lower-index element k before stage u
    bool red_end[1 << LOGE];       // reduce element k after the last stage (phase hand-over above out_cap)
    int out[1 << LOGE];            // bound of element k when the phase ends
    int out_bound;                 // max of out[]
};

// stages u = 0 .. r-1 act on local bits lb_top, lb_top - 1, ...; every input < in_bound; words above out_cap are reduced at the end
template <int LOGE>
constexpr CtfPlan<LOGE> make_ctf_plan(int lb_top, int r, int in_bound, int out_cap) {
    CtfPlan<LOGE> p{};
    constexpr int E = 1 << LOGE;
    int bnd[E] = {};
    for (int k = 0; k < E; ++k) bnd[k] = in_bound;
    for (int u = 0; u < r; ++u) {
        const int bit = 1 << (lb_top - u);
        for (int k = 0; k < E; ++k) {
            if (k & bit) continue;
            int A = bnd[k];
            const int Y = bnd[k | bit];
            // reduce a when the fused chain would not fit, or when the phase's last stage would hand over y' above the cap
            if (A + Y / 4 + 1 > kFuseLimit || (u == r - 1 && A > kRedB && 2 * A + 2 * kUnit > out_cap)) { p.red_a[u][k] = true; A = kRedB; }
            bnd[k] = kRedB;
            bnd[k | bit] = 2 * A + 2 * kUnit;
        }
    }
    int mx = 0;
    for (int k = 0; k < E; ++k) {
        if (bnd[k] > out_cap) { p.red_end[k] = true; bnd[k] = kRedB; }
        p.out[k] = bnd[k];
        if (bnd[k] > mx) mx = bnd[k];
    }
    p.out_bound = mx;
    return p;
}

// ------------------------------------------------------------------------------------------------
// static bound plans (F64ArithT).  Bounds are on |x|, in units of q/1024.  A product's multiplied word must stay below cap = kCap q (<= 2^51); sums of
// two words below 4 cap (exact additions: < 2^53).  A reduction leaves |x| <= q/2 + 1, a product |x| <= q.  With kCap = 16 (primes below 2^47) the forward
// plan of every compiled ring degree is empty; with kCap = 2 (primes below 2^50) most multiplied words are reduced first.
// ------------------------------------------------------------------------------------------------
constexpr int kF64Red = kUnit / 2 + 1, kF64Tw = kUnit;
template <int LOGE>
struct F64CtPlan {
    bool red_y[LOGE][1 << LOGE];   // reduce the multiplied word (upper-index element k | bit) of the butterfly with lower-index element k before stage u
    bool red_a[LOGE][1 << LOGE];   // reduce the other word first (its sum with the product would leave the exact range)
    bool red_end[1 << LOGE];
    int out_bound;
};
template <int LOGE>
constexpr F64CtPlan<LOGE> make_f64_ct_plan(int lb_top, int r, int in_bound, int out_cap, int cap) {
    F64CtPlan<LOGE> p{};
    constexpr int E = 1 << LOGE;
    int bnd[E] = {};
    for (int k = 0; k < E; ++k) bnd[k] = in_bound;
    for (int u = 0; u < r; ++u) {
        const int bit = 1 << (lb_top - u);
        for (int k = 0; k < E; ++k) {
            if (k & bit) continue;
            int A = bnd[k], Y = bnd[k | bit];
            if (Y > cap) { p.red_y[u][k] = true; Y = kF64Red; }
            if (A + kF64Tw > 4 * cap) { p.red_a[u][k] = true; A = kF64Red; }
            bnd[k] = bnd[k | bit] = A + kF64Tw;
        }
    }
    int mx = 0;
    for (int k = 0; k < E; ++k) {
        if (bnd[k] > out_cap) { p.red_end[k] = true; bnd[k] = kF64Red; }
        if (bnd[k] > mx) mx = bnd[k];
    }
    p.out_bound = mx;
    return p;
}
template <int LOGE>
struct F64GsPlan {
    bool red[LOGE][1 << LOGE];
    bool red_end[1 << LOGE];
};
template <int LOGE>
constexpr F64GsPlan<LOGE> make_f64_gs_plan(int lb0, int r, int in_bound, int out_bound, bool last_all_mul, int cap) {
    F64GsPlan<LOGE> p{};
    constexpr int E = 1 << LOGE;
    int bnd[E] = {};
    for (int k = 0; k < E; ++k) bnd[k] = in_bound;
    for (int u = 0; u < r; ++u) {
        const int bit = 1 << (lb0 + u);
        for (int k = 0; k < E; ++k) {
            if (k & bit) continue;
            int bx = bnd[k], by = bnd[k | bit];
            if (bx + by > cap) {   // the difference is multiplied (and, in the last stage, the sum too)
                if (bx > kF64Red) { p.red[u][k] = true; bx = kF64Red; }
                if (by > kF64Red) { p.red[u][k | bit] = true; by = kF64Red; }
            }
            bnd[k] = (last_all_mul && u == r - 1) ? kF64Tw : bx + by;
            bnd[k | bit] = kF64Tw;
        }
    }
    for (int k = 0; k < E; ++k) {
        p.red_end[k] = bnd[k] > out_bound;
        if (p.red_end[k]) bnd[k] = kF64Red;
    }
    return p;
}

// ------------------------------------------------------------------------------------------------
// per-thread transform body
// ------------------------------------------------------------------------------------------------
// SUB = 1: the body runs as one HALF of a 2N-point transform whose column stage (the first forward / last inverse radix-2 stage) the caller
// runs in registers (ntt_halves.h): forward inputs arrive lazy (< kCtfMid q/1024, the column stage's differences) instead of canonical, and the
// inverse leaves out N^-1 and hands over words below kSubInvOut q/1024 (the caller's column stage multiplies by (2N)^-1).
constexpr int kSubInvOut = 7 * kUnit;   // lo + hi < 14 q: the column stage's sum fits mul_ninv's precondition, its difference + 7 q a 64-bit word
// FWD_IN: static bound (q/1024) of every word the FORWARD transform is given.  kUnit = canonical residues; kRedB = any residue below 2^60 (a word that is
// canonical for ANOTHER limb of a FoldArith context - the digits of a key switch: the first stage's fused multiply-add reduces it for free).
// RAW_INV: the INVERSE transform's input is already in the policy's internal form (register-resident products of forward outputs in the fused
// multiply: doubles for F64Arith, scaled words for FoldScaledArith) - no Arith::enter in front of its first stage.
template <class Arith, int LOGN, int LOGE, int SUB = 0, int FWD_IN = (SUB ? kCtfMid : kUnit), bool RAW_INV = false>
struct NttBody {
    typedef Geo<LOGN, LOGE> G;
    typedef typename Arith::Tw Tw;
    static constexpr int E = G::E, T = G::T, NPH = G::NPH;
    static_assert(FWD_IN >= kUnit && FWD_IN <= kCtfMid && (!SUB || FWD_IN == kCtfMid), "forward input bound: canonical ... the phase hand-over bound");

    // ---------------- global <-> registers ----------------
    // window-top mapping (forward input / inverse output): word j = k*T + tid, 8 B per lane, coalesced
    // (unsigned index on a workgroup-uniform base: hipcc then uses the SGPR-base + 32-bit VGPR offset form and
    //  spends no VALU on 64-bit address arithmetic)
    // Rows that the 13-bit immediate offset cannot reach get their own SCALAR base (SALU adds, free on a VALU-bound
    // kernel): left to itself hipcc builds 64-bit per-lane addresses with v_add_co / v_addc pairs.
    static DPF_HD unsigned row_base(int k) {   // word offset of the immediate-offset window that holds row k, kept in a scalar register
        constexpr int kReach = 4096 / (T * 8) > 0 ? 4096 / (T * 8) : 1;   // rows per window
        unsigned b = (unsigned)(k / kReach * kReach) * T;
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+s"(b));   // opaque (not volatile: a side-effect asm would stop hipcc from using scalar loads for the uniform twiddles): the access becomes (scalar base + b) + 32-bit lane offset + immediate
#endif
        return b;
    }
    // NT: non-temporal accesses for data that is read once and written once and does not fit the 256 MiB Infinity Cache (the
    // fused multiply's 5 GiB working set: -2.4 %).  The batched NTT kernels keep plain accesses: at BASELINE configs[1]'s
    // 128 MiB + 128 MiB the cache holds the stream and non-temporal accesses measured 6 % slower.
    template <bool NT, class V>
    static DPF_HD V ld_stream(const V* p) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (NT) return __builtin_nontemporal_load(p);
#endif
        return *p;
    }
    template <bool NT, class V>
    static DPF_HD void st_stream(V* p, V v) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (NT) { __builtin_nontemporal_store(v, p); return; }
#endif
        *p = v;
    }
    template <bool NT = false>
    static DPF_HD void load_top(int tid, u64 (&x)[E], const u64* g) {
        constexpr int kReach = 4096 / (T * 8) > 0 ? 4096 / (T * 8) : 1;
#pragma clang loop unroll(full)
        for (int k = 0; k < E; ++k) x[k] = ld_stream<NT>(&(g + row_base(k))[(unsigned)((k % kReach) * T) + (unsigned)tid]);
    }
    template <bool NT = false>
    static DPF_HD void store_top(int tid, const u64 (&x)[E], u64* g) {
        constexpr int kReach = 4096 / (T * 8) > 0 ? 4096 / (T * 8) : 1;
#pragma clang loop unroll(full)
        for (int k = 0; k < E; ++k) st_stream<NT>(&(g + row_base(k))[(unsigned)((k % kReach) * T) + (unsigned)tid], x[k]);
    }
    // window-0 mapping (forward output / inverse input): thread owns words [tid*E, tid*E + E)
    struct alignas(16) V2 {
        u64 a, b;
    };
    static constexpr int PC = E / 2, LP = LOGE - 1, kLow = 6 - LP;   // pieces per thread; lane bits [kLow, 6) <-> piece index
    static constexpr bool kWaveIO = (T % 64 == 0) && LP >= 1 && LP <= 4;   // register transposition available
    static constexpr bool kLdsIO = kWaveIO && PC == 8 && NPH >= 2;         // LDS transposition available (load_bot_lds / store_bot_lds)
    static_assert(!kLdsIO || 64 * (E + 2) <= G::kWaveStride, "the row layout must fit the wave's region");
#if defined(__HIP_DEVICE_COMPILE__)
    // Device path: a thread's E words are 8E contiguous bytes, so plain 16-byte accesses put the 64 lanes of one
    // instruction on 64 different 128-byte lines.  The wave instead moves 1 KiB-contiguous slices (lane L, slice r:
    // piece L >> kLow of thread (r << kLow) | (L & (2^kLow - 1)), a piece = 16 bytes) and transposes pieces <-> lanes
    // in registers: one swap stage per piece-index bit (v_permlane32_swap, v_permlane16_swap, row DPP), an involution.
    template <int LANEBIT>
    static __device__ __forceinline__ void lane_swap(u32& a, u32& b) {  // lanes with LANEBIT clear: b <- partner's a; set: a <- partner's b
        if constexpr (LANEBIT == 5) { auto v = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = v[0]; b = v[1]; }
        else if constexpr (LANEBIT == 4) { auto v = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = v[0]; b = v[1]; }
        else if constexpr (LANEBIT == 3) {
            const u32 na = __builtin_amdgcn_update_dpp(a, b, 0x118 /*row_shr:8*/, 0xf, 0xc, false);
            const u32 nb = __builtin_amdgcn_update_dpp(b, a, 0x108 /*row_shl:8*/, 0xf, 0x3, false);
            a = na; b = nb;
        } else {
            static_assert(LANEBIT == 2, "piece transposition uses lane bits 2..5");
            const u32 na = __builtin_amdgcn_update_dpp(a, b, 0x114 /*row_shr:4*/, 0xf, 0xa, false);
            const u32 nb = __builtin_amdgcn_update_dpp(b, a, 0x104 /*row_shl:4*/, 0xf, 0x5, false);
            a = na; b = nb;
        }
    }
    template <int S>
    static __device__ __forceinline__ void transpose_stage(u64 (&x)[E]) {  // piece-index bit LP-1-S <-> lane bit 5-S
        constexpr int rb = 1 << (LP - 1 - S);
#pragma clang loop unroll(full)
        for (int r = 0; r < PC; ++r) {
            if (r & rb) continue;
#pragma clang loop unroll(full)
            for (int e = 0; e < 2; ++e) {
                u32 alo = (u32)x[2 * r + e], ahi = (u32)(x[2 * r + e] >> 32);
                u32 blo = (u32)x[2 * (r | rb) + e], bhi = (u32)(x[2 * (r | rb) + e] >> 32);
                lane_swap<5 - S>(alo, blo);
                lane_swap<5 - S>(ahi, bhi);
                x[2 * r + e] = (u64)alo | ((u64)ahi << 32);
                x[2 * (r | rb) + e] = (u64)blo | ((u64)bhi << 32);
            }
        }
        if constexpr (S + 1 < LP) transpose_stage<S + 1>(x);
    }
    static __device__ __forceinline__ unsigned slice_piece(int tid, int r) {  // 16-byte piece index of slice r for this lane
        const unsigned lane = (unsigned)tid & 63u, wave_thread0 = (unsigned)tid & ~63u;
        const unsigned t = wave_thread0 + (((unsigned)r << kLow) | (lane & ((1u << kLow) - 1u)));
        return t * PC + (lane >> kLow);
    }
    // The same transposition through LDS instead of registers: no VALU work at all (the kernels are VALU-bound, the LDS
    // pipe is ~10 % busy).  Each thread's E words live in its own padded row (E + 2 words: conflict-free 16-byte accesses);
    // a wave only ever touches the rows of its own 64 threads, which are also exactly what it read in the last forward
    // exchange, so no workgroup barrier is needed around it - only before a LATER all-to-all exchange overwrites the rows.
    // Lane L moves, in slice r, piece (L/8 + L%8) % 8 of thread 8r + L%8: the skew keeps 16 consecutive lanes on 16
    // different 16-byte slots.
    // Both results are a per-thread base (slice 0) plus a compile-time multiple of r - written that way so that ONE address register
    // and immediate offsets serve all eight slices (left as lds_row(thread of slice r) the compiler rebuilds every address with
    // bit-field arithmetic: ~30 VALU instructions per transform).
    static __device__ __forceinline__ void lds_slice(int tid, int r, unsigned& lds_word, unsigned& piece) {
        const unsigned lane = (unsigned)tid & 63u, wave_thread0 = (unsigned)tid & ~63u, i = lane & 7u, pc = ((lane >> 3) + i) & 7u;
        const unsigned base_w = (unsigned)G::lds_row((int)(wave_thread0 + i)) + pc * 2, base_p = (wave_thread0 + i) * PC + pc;   // thread 8 r + i of the wave, r = 0
        lds_word = base_w + (unsigned)r * (8u * (E + 2));
        piece = base_p + (unsigned)r * (8u * PC);
    }
    template <bool NT = false>
    static __device__ __forceinline__ void load_bot_lds(int tid, u64 (&x)[E], const u64* g, u64* lds) {
        const V2* p = reinterpret_cast<const V2*>(g);
        V2 v[PC];
#pragma clang loop unroll(full)
        for (int r = 0; r < PC; ++r) {
            unsigned w, pc; lds_slice(tid, r, w, pc);
            if (NT) { v[r].a = __builtin_nontemporal_load(&p[pc].a); v[r].b = __builtin_nontemporal_load(&p[pc].b); }
            else v[r] = p[pc];
        }
#pragma clang loop unroll(full)
        for (int r = 0; r < PC; ++r) { unsigned w, pc; lds_slice(tid, r, w, pc); *reinterpret_cast<V2*>(lds + w) = v[r]; }
        const V2* row = reinterpret_cast<const V2*>(lds + (unsigned)G::lds_row(tid));   // same wave wrote it: program order + lgkmcnt
#pragma clang loop unroll(full)
        for (int k = 0; k < PC; ++k) { V2 t = row[k]; x[2 * k] = t.a; x[2 * k + 1] = t.b; }
    }
    // The same staging in two halves, for callers that want the global loads in flight while they compute: stage_load requests the
    // wave's 64 E-word region (lane-contiguous 16-byte pieces), stage_rows / stage_gather put it through the wave's LDS rows.
    // stage_gather reads word k of the thread from LDS word addr[k] instead of its own row: the Galois permutation in the NTT domain
    // maps every aligned 2^s-word block ONTO an aligned 2^s-word block (high bits of the bit-reversed index depend on high bits only),
    // so a thread's E source words sit in ONE row and a wave's sources in ONE 64 E-word region - `g` is then the polynomial shifted
    // by (source region - own region) * 64 E words, and no 8-byte global gather is left.  Rows are wave-private: wavefront-scope
    // ordering only (the LDS runs one wave's DS instructions in order); the caller fences before reusing the rows.
    static __device__ __forceinline__ void stage_load(int tid, u64 (&v)[E], const u64* g) {   // plain words: struct arrays carried around a loop stay in scratch
        const V2* p = reinterpret_cast<const V2*>(g);
#pragma clang loop unroll(full)
        for (int r = 0; r < PC; ++r) { unsigned w, pc; lds_slice(tid, r, w, pc); const V2 t = p[pc]; v[2 * r] = t.a; v[2 * r + 1] = t.b; }
    }
    static __device__ __forceinline__ void stage_write(int tid, const u64 (&v)[E], u64* lds) {
#pragma clang loop unroll(full)
        for (int r = 0; r < PC; ++r) { unsigned w, pc; lds_slice(tid, r, w, pc); *reinterpret_cast<V2*>(lds + w) = V2{v[2 * r], v[2 * r + 1]}; }
    }
    static __device__ __forceinline__ void stage_rows(int tid, u64 (&x)[E], const u64 (&v)[E], u64* lds) {
        stage_write(tid, v, lds);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const V2* row = reinterpret_cast<const V2*>(lds + (unsigned)G::lds_row(tid));
#pragma clang loop unroll(full)
        for (int k = 0; k < PC; ++k) { V2 t = row[k]; x[2 * k] = t.a; x[2 * k + 1] = t.b; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    static __device__ __forceinline__ void stage_gather(int tid, u64 (&x)[E], const u64 (&v)[E], u64* lds, const unsigned (&addr)[E]) {
        stage_write(tid, v, lds);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma clang loop unroll(full)
        for (int k = 0; k < E; ++k) x[k] = lds[addr[k]];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // source addressing of stage_gather for the thread's E output positions tid E + k under sigma_g: LDS word addresses of the
    // source words, and the region shift (in words) to add to the polynomial's base
    static __device__ __forceinline__ long gather_plan(int tid, unsigned g, unsigned (&addr)[E]) {
        unsigned src0 = 0;
#pragma clang loop unroll(full)
        for (int k = 0; k < E; ++k) {
            const unsigned p = (unsigned)tid * E + k;
            const unsigned e = 2u * (__brev(p) >> (32 - LOGN)) + 1u;
            const unsigned e2 = (g * e) & (2u * G::N - 1u);
            const unsigned sp = __brev((e2 - 1u) >> 1) >> (32 - LOGN);
            if (k == 0) src0 = sp;
            addr[k] = sp & (E - 1);
        }
        const unsigned row = (unsigned)G::lds_row((int)(((unsigned)tid & ~63u) + ((src0 >> LOGE) & 63u)));
#pragma clang loop unroll(full)
        for (int k = 0; k < E; ++k) addr[k] += row;
        const int region = __builtin_amdgcn_readfirstlane((int)(src0 >> (6 + LOGE)));     // wave-uniform
        return ((long)region - (long)(tid >> 6)) * (long)(64 * E);
    }
    template <bool NT = false>
    static __device__ __forceinline__ void store_bot_lds(int tid, const u64 (&x)[E], u64* g, u64* lds) {
        V2* row = reinterpret_cast<V2*>(lds + (unsigned)G::lds_row(tid));
#pragma clang loop unroll(full)
        for (int k = 0; k < PC; ++k) row[k] = V2{x[2 * k], x[2 * k + 1]};
        V2* p = reinterpret_cast<V2*>(g);
#pragma clang loop unroll(full)
        for (int r = 0; r < PC; ++r) {
            unsigned w, pc; lds_slice(tid, r, w, pc);
            const V2 t = *reinterpret_cast<const V2*>(lds + w);
            if (NT) { __builtin_nontemporal_store(t.a, &p[pc].a); __builtin_nontemporal_store(t.b, &p[pc].b); }
            else p[pc] = t;
        }
    }
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
    static void stage_load(int, u64 (&)[E], const u64*) {}
    static void stage_rows(int, u64 (&)[E], const u64 (&)[E], u64*) {}
    static void stage_gather(int, u64 (&)[E], const u64 (&)[E], u64*, const unsigned (&)[E]) {}
    static long gather_plan(int, unsigned, unsigned (&)[E]) { return 0; }
    template <bool NT = false> static DPF_HD void load_bot_lds(int, u64 (&)[E], const u64*, u64*) {}   // device-only paths: never called on the host
    template <bool NT = false> static DPF_HD void store_bot_lds(int, const u64 (&)[E], u64*, u64*) {}
#endif
    static DPF_HD void load_bot(int tid, u64 (&x)[E], const u64* g) {
        const V2* p = reinterpret_cast<const V2*>(g);
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (kWaveIO) {
#pragma clang loop unroll(full)
            for (int r = 0; r < PC; ++r) { V2 v = p[slice_piece(tid, r)]; x[2 * r] = v.a; x[2 * r + 1] = v.b; }
            transpose_stage<0>(x);
            return;
        }
#endif
#pragma clang loop unroll(full)
        for (int k = 0; k < E / 2; ++k) { V2 v = p[(unsigned)tid * (E / 2) + k]; x[2 * k] = v.a; x[2 * k + 1] = v.b; }
    }
    static DPF_HD void store_bot(int tid, const u64 (&x)[E], u64* g) {
        V2* p = reinterpret_cast<V2*>(g);
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (kWaveIO) {
            u64 y[E];
#pragma clang loop unroll(full)
            for (int k = 0; k < E; ++k) y[k] = x[k];
            transpose_stage<0>(y);
#pragma clang loop unroll(full)
            for (int r = 0; r < PC; ++r) p[slice_piece(tid, r)] = V2{y[2 * r], y[2 * r + 1]};
            return;
        }
#endif
#pragma clang loop unroll(full)
        for (int k = 0; k < E / 2; ++k) p[(unsigned)tid * (E / 2) + k] = V2{x[2 * k], x[2 * k + 1]};
    }

    // ---------------- LDS exchange between forward phases P and P+1 ----------------
    // side = the phase whose register mapping is used for this access
    template <int P, int SIDE, bool FWD>
    static DPF_HD int xaddr(int tid, int k) {
        constexpr int chi = G::phase(P).c, clo = G::phase(P + 1).c;
        constexpr int c = (SIDE == P) ? chi : clo;
        // index(c, tid, k) = index(c, tid, 0) | (k << c) with disjoint bits: base(tid) + offset(k), the offset a constant
        // after unrolling, so one address register and immediate offsets serve all E accesses
        return G::template lds_addr<clo, chi, FWD>(G::index(c, tid, 0)) + G::template lds_addr<clo, chi, FWD>(k << c);
    }
    template <int P, int SIDE, bool FWD>
    static DPF_HD void lds_write(int tid, const u64 (&x)[E], u64* lds) {
        constexpr int c = G::phase(SIDE).c;
        if (c == 0) {
            V2* p = reinterpret_cast<V2*>(lds + xaddr<P, SIDE, FWD>(tid, 0));
#pragma clang loop unroll(full)
            for (int k = 0; k < E / 2; ++k) p[k] = V2{x[2 * k], x[2 * k + 1]};
        } else {
#pragma clang loop unroll(full)
            for (int k = 0; k < E; ++k) lds[(unsigned)xaddr<P, SIDE, FWD>(tid, k)] = x[k];
        }
    }
    template <int P, int SIDE, bool FWD>
    static DPF_HD void lds_read(int tid, u64 (&x)[E], const u64* lds) {
        constexpr int c = G::phase(SIDE).c;
        if (c == 0) {
            const V2* p = reinterpret_cast<const V2*>(lds + xaddr<P, SIDE, FWD>(tid, 0));
#pragma clang loop unroll(full)
            for (int k = 0; k < E / 2; ++k) { V2 v = p[k]; x[2 * k] = v.a; x[2 * k + 1] = v.b; }
        } else {
#pragma clang loop unroll(full)
            for (int k = 0; k < E; ++k) x[k] = lds[(unsigned)xaddr<P, SIDE, FWD>(tid, k)];
        }
    }

    // ---------------- forward (Cooley-Tukey) phase ----------------
    // Per-thread twiddles of one phase, fetched ahead of use: twr[u][i] is the i-th distinct twiddle of
    // in-phase stage u (2^(LOGE-1-lb) of them).  Only the entries a phase touches are ever live.
    typedef Tw TwRegs[LOGE][E / 2];

    template <int P, bool FWD>
    static DPF_HD void load_tw(int tid, const Tw* tw, TwRegs& twr) {
        constexpr Phase ph = G::phase(P);
        const int th = tid_high<G>(ph.c, tid);
#pragma clang loop unroll(full)
        for (int u = 0; u < ph.r; ++u) {
            const int pos = FWD ? (ph.b + ph.r - 1 - u) : (ph.b + u);
            const int lb = pos - ph.c;
#pragma clang loop unroll(full)
            for (int i = 0; i < E / 2; ++i)  // constant trip count (the bound below folds after unrolling u)
                if (i < (1 << (LOGE - 1 - lb))) twr[u][i] = (tw + tw_index_uniform<G>(ph.c, i << (lb + 1), pos))[tw_index_thread<G>(ph.c, th, pos)];
        }
    }

    // F64ArithT: kF64Cap = the product's precondition in plan units; phase hand-over at 2 cap (additions stay exact below 4 cap); the last forward phase
    // hands over at most cap, what the lazy products take (prod below) and what leave() converts
    template <class A = Arith> static constexpr int f64_cap_of() { if constexpr (A::kF64) return A::kCap * kUnit; else return kWord; }
    static constexpr int kF64Cap = f64_cap_of<>();
    template <int P>
    static constexpr F64CtPlan<LOGE> f64_ct_plan() {
        constexpr Phase ph = G::phase(P);
        int in = kUnit;
        for (int i = 0; i < P; ++i) {
            const Phase pi = G::phase(i);
            in = make_f64_ct_plan<LOGE>(pi.b - pi.c + pi.r - 1, pi.r, in, 2 * kF64Cap, kF64Cap).out_bound;
        }
        return make_f64_ct_plan<LOGE>(ph.b - ph.c + ph.r - 1, ph.r, in, (P == NPH - 1) ? kF64Cap : 2 * kF64Cap, kF64Cap);
    }
    template <int P, int IN>
    static constexpr F64GsPlan<LOGE> f64_gs_plan() {
        constexpr Phase ph = G::phase(P);
        constexpr int mid = kF64Cap / 2 > 2 * kUnit ? 2 * kUnit : kF64Cap / 2;   // two such words may meet in the next phase's first butterfly
        return make_f64_gs_plan<LOGE>(ph.b - ph.c, ph.r, (P == NPH - 1) ? IN : mid, (P == 0) ? kF64Cap : mid, P == 0, kF64Cap);
    }
    // bound plan of forward phase P (FoldArith): canonical input to phase 0, kCtfMid at every exchange, nothing capped after the last
    template <int P>
    static constexpr CtfPlan<LOGE> ctf_plan() {
        constexpr Phase ph = G::phase(P);
        int in = FWD_IN;
        for (int i = 0; i < P; ++i) {   // hand-over bound of the previous phase
            const Phase pi = G::phase(i);
            in = make_ctf_plan<LOGE>(pi.b - pi.c + pi.r - 1, pi.r, in, kCtfMid).out_bound;
        }
        return make_ctf_plan<LOGE>(ph.b - ph.c + ph.r - 1, ph.r, in, (P == NPH - 1) ? kWord : kCtfMid);
    }
    // bound of every word a forward transform hands to a dyadic product when its output is left lazy
    static constexpr int kFwdOutBound = Arith::kFoldCore ? ctf_plan<NPH - 1>().out_bound : Arith::kF64 ? f64_ct_plan<NPH - 1>().out_bound : 4 * kUnit;
    static_assert(!Arith::kFoldCore || kFwdOutBound <= kLimitPartner, "lazy forward outputs must satisfy mul60's bound");

    // generic policies (FoldScaledArith, F64Arith) convert a canonical word when it enters a transform and back when it leaves (Arith::enter /
    // Arith::leave); the pinned-prime and Harvey policies work on the words as they are
    static constexpr bool kConverts = Arith::kF64 || (Arith::kFoldCore && !Arith::kFold);
    static_assert(!kConverts || !SUB, "the halves form is FoldArith's");

    static DPF_HD void enter(u64 (&x)[E], const LimbConst& lc) {
        if constexpr (kConverts) {
#pragma clang loop unroll(full)
            for (int k = 0; k < E; ++k) x[k] = Arith::enter(x[k], lc);
        }
    }
    template <int P>
    static DPF_HD void fwd_phase_r(u64 (&x)[E], const TwRegs& twr, const LimbConst& lc_in) {
        constexpr Phase ph = G::phase(P);
        const auto& lc = Arith::ntt_lc(lc_in);   // FoldScaledArith: the scaled modulus q' = 2^60 - d
        if constexpr (P == 0) enter(x, lc_in);
        const u64 two_q = 2 * lc.q;
        if constexpr (Arith::kF64) {
            constexpr F64CtPlan<LOGE> plan = f64_ct_plan<P>();
            const double q = Arith::qd(lc), qi = Arith::qinv(lc);
#pragma clang loop unroll(full)
            for (int u = 0; u < ph.r; ++u) {
                const int lb = ph.b + ph.r - 1 - u - ph.c;
#pragma clang loop unroll(full)
                for (int k = 0; k < E; ++k) {
                    if (k & (1 << lb)) continue;
                    const int kk = k | (1 << lb);
                    const Tw& w = twr[u][k >> (lb + 1)];
                    double a = Arith::f(x[k]), y = Arith::f(x[kk]);
                    if (plan.red_a[u][k]) a = Arith::reduce(a, q, qi);
                    if (plan.red_y[u][k]) y = Arith::reduce(y, q, qi);
                    const double t = Arith::mulmod(y, w.w, w.wq, q);   // |t| <= q: a word grows by at most q per stage
                    x[k] = Arith::b(a + t);
                    x[kk] = Arith::b(a - t);
                }
            }
#pragma clang loop unroll(full)
            for (int k = 0; k < E; ++k)
                if (plan.red_end[k]) x[k] = Arith::b(Arith::reduce(Arith::f(x[k]), q, qi));
        } else if constexpr (Arith::kFoldCore) {
            constexpr CtfPlan<LOGE> plan = ctf_plan<P>();
#pragma clang loop unroll(full)
            for (int u = 0; u < ph.r; ++u) {
                const int lb = ph.b + ph.r - 1 - u - ph.c;
#pragma clang loop unroll(full)
                for (int k = 0; k < E; ++k) {
                    if (k & (1 << lb)) continue;
                    const int kk = k | (1 << lb);
                    u64 a = x[k];
                    if (plan.red_a[u][k]) a = FoldArith::reduce(a, lc);
                    const u64 s = FoldArith::mul_tw_add(x[kk], twr[u][k >> (lb + 1)], lc, a);
                    x[k] = s;                                   // a + w y, < 2^60 + 16 d
                    x[kk] = chk_shl1_add_sub(a, two_q, s);      // a - w y = 2a + 2q - x'
                }
            }
#pragma clang loop unroll(full)
            for (int k = 0; k < E; ++k)
                if (plan.red_end[k]) x[k] = FoldArith::reduce(x[k], lc);
        } else {
#pragma clang loop unroll(full)
            for (int u = 0; u < ph.r; ++u) {
                const int lb = ph.b + ph.r - 1 - u - ph.c;
#pragma clang loop unroll(full)
                for (int k = 0; k < E; ++k) {
                    if (k & (1 << lb)) continue;
                    const u64 a = csub(x[k], two_q);            // Harvey: [0,4q) -> [0,2q)
                    const u64 t = Arith::mul_tw(x[k | (1 << lb)], twr[u][k >> (lb + 1)], lc);
                    x[k] = a + t;
                    x[k | (1 << lb)] = a - t + two_q;
                }
            }
        }
    }
    template <int P>
    static DPF_HD void fwd_phase(int tid, u64 (&x)[E], const Tw* tw, const LimbConst& lc) {
        TwRegs twr;
        load_tw<P, true>(tid, tw, twr);
        fwd_phase_r<P>(x, twr, lc);
    }
    // forward output -> canonical residues
    static DPF_HD void fwd_canon(u64 (&x)[E], const LimbConst& lc_in) {
        const auto& lc = Arith::ntt_lc(lc_in);
        if constexpr (Arith::kF64) {
#pragma clang loop unroll(full)
            for (int k = 0; k < E; ++k) x[k] = Arith::leave(x[k], lc);
        } else if constexpr (Arith::kFoldCore) {
            constexpr CtfPlan<LOGE> plan = ctf_plan<NPH - 1>();   // sums leave the last stage reduced: 4 instructions instead of 7
#pragma clang loop unroll(full)
            for (int k = 0; k < E; ++k) {
                x[k] = plan.out[k] <= kRedB ? FoldArith::canon_small(x[k], lc) : FoldArith::canon(x[k], lc);
                if constexpr (kConverts) x[k] = Arith::leave(x[k], lc_in);
            }
        } else {
#pragma clang loop unroll(full)
            for (int k = 0; k < E; ++k) x[k] = csub(csub(x[k], 2 * lc.q), lc.q);
        }
    }

    // forward output left lazy for a dyadic product, but every word < 2^60 + 2^29 (mul60's bound on its SECOND operand):
    // only the differences (y') need the 3-instruction reduction, the sums are reduced already
    static DPF_HD void fwd_reduce_partner(u64 (&x)[E], const LimbConst& lc_in) {
        static_assert(Arith::kFoldCore, "fold policies only");
        const auto& lc = Arith::ntt_lc(lc_in);
        constexpr CtfPlan<LOGE> plan = ctf_plan<NPH - 1>();
#pragma clang loop unroll(full)
        for (int k = 0; k < E; ++k)
            if (plan.out[k] > kRedB) x[k] = FoldArith::reduce(x[k], lc);
    }

    // ---------------- lazy products of forward outputs (the fused multiply's tensor step, in registers) ----------------
    // prod(a, b): a = a forward output as the transform left it, b = one that went through prod_partner.  Sums of two products (prod_add) feed the
    // inverse transform of a RAW_INV body with IN = kProdInvIn.  FoldScaledArith: a product of two scaled words carries the scale twice; the inverse
    // then runs on last-stage twiddles with s^-1 folded in (DevTables::last2).
    static constexpr bool kLazyProducts = Arith::kFoldCore || Arith::kF64;
    static constexpr int kProdInvIn = Arith::kF64 ? 2 * kUnit : kRedB;   // (fold: what `tensor` leaves; prod / prod_add sums stay below it only through tensor)
    static_assert(!Arith::kF64 || kFwdOutBound <= kF64Cap, "F64ArithT: |a b / q| <= |a| / 2 must stay below 2^50 for the quotient estimate of a lazy product");
    static DPF_HD void prod_partner(u64 (&y)[E], const LimbConst& lc) {
        if constexpr (Arith::kF64) {
            const double q = Arith::qd(lc), qi = Arith::qinv(lc);
#pragma clang loop unroll(full)
            for (int k = 0; k < E; ++k) y[k] = Arith::b(Arith::reduce(Arith::f(y[k]), q, qi));   // |y| <= q / 2 + 1
        } else fwd_reduce_partner(y, lc);
    }
    static DPF_HD u64 prod(u64 a, u64 b, const LimbConst& lc) {
        if constexpr (Arith::kF64) {
            // |a| < kCap q <= 2^51, |b| <= q / 2 + 1: the quotient estimate is within 0.375 of a b / q, so |result| < 0.875 q + 1; both fma exact (modarith.h)
            const double x = Arith::f(a), y = Arith::f(b), q = Arith::qd(lc);
            const double p = x * y;
            const double e = __builtin_fma(x, y, -p);
            const double h = Arith::rint_mul(p, Arith::qinv(lc));
            return Arith::b(__builtin_fma(-h, q, p) + e);
        } else return FoldArith::mul60(a, b, (u32)lc.d);
    }
    static DPF_HD u64 prod_add(u64 p, u64 r) {
        if constexpr (Arith::kF64) return Arith::b(Arith::f(p) + Arith::f(r));
        else return chk_add(p, r);
    }
    // the tensor step of one coefficient: (c0, c1, c2) = (a0 b0, a0 b1 + a1 b0, a1 b1); b0, b1 went through prod_partner.  Fold policies (round 6): each b
    // becomes a twiddle on the fly (FoldArith::prod_tw, 8 instructions), the four products run through the twiddle chain (9 each, the sum of c1 through its
    // addend): ~53 instructions per coefficient against ~103 with four mul60; every output is ONE reduced word (kProdInvIn = kRedB).
    static DPF_HD void tensor(u64 a0, u64 a1, u64 b0, u64 b1, u64& c0, u64& c1, u64& c2, const LimbConst& lc) {
        if constexpr (Arith::kFoldCore) {
            const Tw t0 = FoldArith::prod_tw(b0, lc), t1 = FoldArith::prod_tw(b1, lc);
            c0 = FoldArith::mul_ptw(a0, t0, lc);
            c1 = FoldArith::mul_ptw_add(a1, t0, lc, FoldArith::mul_ptw(a0, t1, lc));
            c2 = FoldArith::mul_ptw(a1, t1, lc);
        } else {
            c0 = prod(a0, b0, lc);
            c1 = prod_add(prod(a0, b1, lc), prod(a1, b0, lc));
            c2 = prod(a1, b1, lc);
        }
    }

    // ---------------- inverse (Gentleman-Sande) phase; forward phase list walked backwards -------
    // IN = static bound of every input word of the whole inverse transform (units of q/1024)
    static constexpr int kGsMid = 2 * kMulB;  // uniform bound re-established at every phase boundary
    template <int P, int IN>
    static constexpr GsPlan<LOGE> gs_plan() {
        constexpr Phase ph = G::phase(P);
        if (SUB) return make_gs_plan<LOGE>(ph.b - ph.c, ph.r, (P == NPH - 1) ? IN : kGsMid, (P == 0) ? kSubInvOut : kGsMid, false);
        return make_gs_plan<LOGE>(ph.b - ph.c, ph.r, (P == NPH - 1) ? IN : kGsMid, (P == 0) ? kLimit : kGsMid, P == 0, Arith::kFold ? LOGN : 0);
    }

    template <int P, int IN>
    static DPF_HD void inv_phase_r(u64 (&x)[E], const TwRegs& twr, const Tw& w_last, const Tw& w_ninv, const LimbConst& lc_in) {
        constexpr Phase ph = G::phase(P);
        const auto& lc = Arith::ntt_lc(lc_in);
        if constexpr (P == NPH - 1 && !RAW_INV) enter(x, lc_in);
        if constexpr (Arith::kF64) {
            constexpr F64GsPlan<LOGE> plan = f64_gs_plan<P, IN>();
            static_assert(IN <= 2 * kF64Cap, "F64ArithT: the sum of the first butterfly's operands must be an exact addition");
            const double q = Arith::qd(lc), qi = Arith::qinv(lc);
#pragma clang loop unroll(full)
            for (int u = 0; u < ph.r; ++u) {
                const int pos = ph.b + u;
                const int lb = pos - ph.c;
                const bool last = (pos == LOGN - 1);
#pragma clang loop unroll(full)
                for (int k = 0; k < E; ++k) {
                    if (k & (1 << lb)) continue;
                    const int kk = k | (1 << lb);
                    double a = Arith::f(x[k]), b = Arith::f(x[kk]);
                    if (plan.red[u][k]) a = Arith::reduce(a, q, qi);
                    if (plan.red[u][kk]) b = Arith::reduce(b, q, qi);
                    const double s = a + b, dlt = a - b;
                    if (last) {
                        x[k] = Arith::b(Arith::mulmod(s, w_ninv.w, w_ninv.wq, q));
                        x[kk] = Arith::b(Arith::mulmod(dlt, w_last.w, w_last.wq, q));
                    } else {
                        const Tw& w = twr[u][k >> (lb + 1)];
                        x[k] = Arith::b(s);
                        x[kk] = Arith::b(Arith::mulmod(dlt, w.w, w.wq, q));
                    }
                }
            }
#pragma clang loop unroll(full)
            for (int k = 0; k < E; ++k)
                if (plan.red_end[k]) x[k] = Arith::b(Arith::reduce(Arith::f(x[k]), q, qi));
        } else {
            constexpr GsPlan<LOGE> plan = gs_plan<P, IN>();
            const u64 q = lc.q, two_q = 2 * lc.q;
#pragma clang loop unroll(full)
            for (int u = 0; u < ph.r; ++u) {
                const int pos = ph.b + u;
                const int lb = pos - ph.c;
                const bool last = (pos == LOGN - 1) && !SUB;   // SUB: the caller's column stage is the last one
#pragma clang loop unroll(full)
                for (int k = 0; k < E; ++k) {
                    if (k & (1 << lb)) continue;
                    const int kk = k | (1 << lb);
                    u64 a = x[k], b = x[kk];
                    u64 s, dlt;
                    if constexpr (Arith::kFoldCore) {
                        if (plan.red[u][k]) a = FoldArith::reduce(a, lc);
                        if (plan.red[u][kk]) b = FoldArith::reduce(b, lc);
                        s = chk_add(a, b);
                        dlt = chk_sub_add(a, b, (u64)plan.off[u][k] * q);
                    } else {  // Harvey: inputs in [0,2q)
                        s = csub(a + b, two_q);
                        dlt = a - b + two_q;
                    }
                    if (last) {  // N^-1 folded into the last stage: x' = (a+b) N^-1, y' = (a-b) w N^-1
                        if constexpr (Arith::kFold) x[k] = FoldArith::mul_ninv(s, lc, LOGN);   // exact division by N: 6 instructions instead of 9
                        else x[k] = tw_mul(s, w_ninv, lc);                                     // (scaled limbs: q' is not 1 mod 2N)
                        x[kk] = tw_mul(dlt, w_last, lc);
                    } else {
                        x[k] = s;
                        x[kk] = tw_mul(dlt, twr[u][k >> (lb + 1)], lc);
                    }
                }
            }
            if constexpr (Arith::kFoldCore) {
#pragma clang loop unroll(full)
                for (int k = 0; k < E; ++k)
                    if (plan.red_end[k]) x[k] = FoldArith::reduce(x[k], lc);
            }
        }
    }
    // twiddle product of the integer policies (FoldScaledArith runs FoldArith's on the scaled modulus)
    static DPF_HD u64 tw_mul(u64 y, const Tw& t, const LimbConst& lc) {
        if constexpr (Arith::kFoldCore) return FoldArith::mul_tw(y, t, lc);
        else return Arith::mul_tw(y, t, lc);
    }
    template <int P, int IN>
    static DPF_HD void inv_phase(int tid, u64 (&x)[E], const Tw* tw, const Tw& w_last, const Tw& w_ninv, const LimbConst& lc) {
        TwRegs twr;
        load_tw<P, false>(tid, tw, twr);
        inv_phase_r<P, IN>(x, twr, w_last, w_ninv, lc);
    }
    // inverse output (all words are outputs of the last stage: twiddle products < 2^60 + 13 d and exact divisions < q + 2^(64 - LOGN) + 1, both
    // inside canon_small's precondition r < 1.5 * 2^60 - the static plan carries the real bound, checked here) -> canonical
    static_assert(!Arith::kFold || SUB || (kWord >> LOGN) + 2 <= kUnit / 2, "the exact division's output (make_gs_plan: kUnit + (sum >> LOGN) + 2) must fit canon_small");
    static DPF_HD void inv_canon(u64 (&x)[E], const LimbConst& lc_in) {
        const auto& lc = Arith::ntt_lc(lc_in);
#pragma clang loop unroll(full)
        for (int k = 0; k < E; ++k) {
            if constexpr (Arith::kF64) x[k] = Arith::leave(x[k], lc);
            else if constexpr (Arith::kFoldCore) {
                x[k] = FoldArith::canon_small(x[k], lc);
                if constexpr (kConverts) x[k] = Arith::leave(x[k], lc_in);
            } else x[k] = csub(x[k], lc.q);
        }
    }
};

}  // namespace dpfhe
