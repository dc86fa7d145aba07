// emulate.cpp - CPU emulation of the HIP NTT kernels (TEST INFRASTRUCTURE).
//
// Runs the exact per-thread code of deeppowers_amd/csrc/ntt_core.h for every thread id of one
// workgroup, step by step between barriers, with LDS as a plain array.  tests/test_emulated_kernels.py
// compares the result with the oracle, so index/twiddle/padding/bound-plan logic is proven on the CPU
// before any GPU minute is spent.  Built by tests (g++), never shipped, never on the product path.
#define DPFHE_EMU_CHECK 1
#include <cstring>
#include <vector>

#include "../deeppowers_amd/csrc/ntt_core.h"
#include "../deeppowers_amd/csrc/ntt_halves.h"
#include "../deeppowers_amd/csrc/ntt_quarters.h"
#include "../deeppowers_amd/csrc/ntt_top.h"
#include "../deeppowers_amd/csrc/tables.h"

namespace dpfhe { long g_emu_overflows = 0; }
using namespace dpfhe;
extern "C" long emu_overflows() { return g_emu_overflows; }

template <class Arith> struct TwTab;
template <> struct TwTab<ShoupArith> {
    static std::vector<TwShoup> make(const std::vector<u64>& w, const std::vector<u64>& sh, u64) {
        std::vector<TwShoup> v(w.size());
        for (size_t i = 0; i < w.size(); ++i) v[i] = TwShoup{w[i], sh[i]};
        return v;
    }
    static TwShoup one(u64 w, u64 sh, u64) { return TwShoup{w, sh}; }
};
template <> struct TwTab<FoldArith> {
    static std::vector<TwFold> make(const std::vector<u64>& w, const std::vector<u64>&, u64 q) {
        std::vector<TwFold> v(w.size());
        for (size_t i = 0; i < w.size(); ++i) v[i] = h_tw_fold(w[i], q);
        return v;
    }
    static TwFold one(u64 w, u64, u64 q) { return h_tw_fold(w, q); }
};

// round 6: the per-limb classes' policies (tables.h limb_class).  class_lc() = the LimbConst their transform kernels read.
template <> struct TwTab<F64Arith> {
    static std::vector<TwF64> make(const std::vector<u64>& w, const std::vector<u64>&, u64 q) {
        std::vector<TwF64> v(w.size());
        for (size_t i = 0; i < w.size(); ++i) v[i] = h_make_tw<TwF64>(w[i], q);
        return v;
    }
    static TwF64 one(u64 w, u64, u64 q) { return h_make_tw<TwF64>(w, q); }
};
template <> struct TwTab<F64WideArith> : TwTab<F64Arith> {};
template <> struct TwTab<FoldScaledArith> {
    static std::vector<TwFold> make(const std::vector<u64>& w, const std::vector<u64>&, u64 q) {
        std::vector<TwFold> v(w.size());
        for (size_t i = 0; i < w.size(); ++i) v[i] = h_tw_fold_scaled(w[i], q, fold_scaled_shift(q));
        return v;
    }
    static TwFold one(u64 w, u64, u64 q) { return h_tw_fold_scaled(w, q, fold_scaled_shift(q)); }
};
template <class Arith> static bool class_ok(u64 q) {
    if (Arith::kFold) return fold_eligible(q);
    if constexpr (Arith::kF64) return q < (1ull << Arith::kMaxBits);
    if (Arith::kFoldCore) return fold_scaled_shift(q) != 0;
    return true;
}
template <class Arith> static LimbConst class_lc(const LimbConst& lc) {
    return limb_const_of_class(lc, Arith::kFold ? kClassFold : Arith::kF64 ? kClassF64 : Arith::kFoldCore ? kClassFoldScaled : kClassShoup);   // (F64Wide reads F64's record)
}

template <class B> static constexpr int E_of() { return B::E; }
// Exchanges are run the way the kernels synchronise them: the all-to-all exchange as "all threads write, barrier, all
// threads read"; a wave-local exchange (Geo::exch_wave_local) one WAVE at a time - write then read - in DESCENDING wave
// order, with no barrier, so a word that had to cross waves, or a region that another wave's exchange clobbers, shows up
// as a mismatch against the oracle (emu_check_lds_regions below proves the address sets disjoint as well).
template <class B, int P, int SIDE_W, int SIDE_R, bool FWD>
static void emu_exchange(std::vector<u64>& regs, std::vector<u64>& lds) {
    constexpr int E = B::E, T = B::T;
    auto X = [&](int tid) -> u64(&)[E] { return *reinterpret_cast<u64(*)[E]>(&regs[(size_t)tid * E]); };
    if (!B::G::exch_wave_local(P)) {
        for (int tid = 0; tid < T; ++tid) B::template lds_write<P, SIDE_W, FWD>(tid, X(tid), lds.data());
        for (int tid = 0; tid < T; ++tid) B::template lds_read<P, SIDE_R, FWD>(tid, X(tid), lds.data());
        return;
    }
    for (int w = (T + 63) / 64 - 1; w >= 0; --w) {
        const int t0 = w * 64, t1 = (t0 + 64 < T) ? t0 + 64 : T;
        for (int tid = t0; tid < t1; ++tid) B::template lds_write<P, SIDE_W, FWD>(tid, X(tid), lds.data());
        for (int tid = t0; tid < t1; ++tid) B::template lds_read<P, SIDE_R, FWD>(tid, X(tid), lds.data());
    }
}

template <class B, int P>
struct FwdSteps {
    static void run(std::vector<u64>& regs, std::vector<u64>& lds, const typename B::Tw* tw, const LimbConst& lc) {
        constexpr int E = B::E, T = B::T;
        auto X = [&](int tid) -> u64(&)[E] { return *reinterpret_cast<u64(*)[E]>(&regs[(size_t)tid * E]); };
        for (int tid = 0; tid < T; ++tid) B::template fwd_phase<P>(tid, X(tid), tw, lc);
        if constexpr (P + 1 < B::NPH) {
            emu_exchange<B, P, P, P + 1, true>(regs, lds);
            FwdSteps<B, P + 1>::run(regs, lds, tw, lc);
        }
    }
};

template <class B, int P, int IN>
struct InvSteps {
    static void run(std::vector<u64>& regs, std::vector<u64>& lds, const typename B::Tw* tw, const typename B::Tw& wl,
                    const typename B::Tw& wn, const LimbConst& lc) {
        constexpr int E = B::E, T = B::T;
        auto X = [&](int tid) -> u64(&)[E] { return *reinterpret_cast<u64(*)[E]>(&regs[(size_t)tid * E]); };
        for (int tid = 0; tid < T; ++tid) B::template inv_phase<P, IN>(tid, X(tid), tw, wl, wn, lc);
        if constexpr (P > 0) {
            emu_exchange<B, P - 1, P, P - 1, false>(regs, lds);
            InvSteps<B, P - 1, IN>::run(regs, lds, tw, wl, wn, lc);
        }
    }
};

// The barrier-free protocol rests on address-set facts; check them exhaustively for one geometry.  For every exchange X
// (forward and inverse direction) let W(X, w) / R(X, w) be the LDS words wave w writes / reads.  Returns 0 when
//   (1) wave-local X:  R(X, w) is a subset of W(X, w), and W(X, w) lies inside wave w's region [w S, (w+1) S)  (S = kWaveStride);
//   (2) the all-to-all exchange, forward direction: every wave READS only its own region (so later wave-local writes need no
//       barrier), inverse direction: every wave WRITES only its own region (so no barrier is needed before it);
//   (3) every address map is injective and inside the buffer.
// A negative return value names the failed check.
template <class B, int P>
static int check_exchanges() {
    constexpr int E = B::E, T = B::T, S = B::G::kWaveStride, W = B::G::kWaves;
    const int words = B::G::lds_words();
    for (int fwd = 0; fwd < 2; ++fwd) {
        std::vector<int> owner_w(words, -1), seen(words, 0);
        // write side / read side register mappings of exchange P in this direction
        for (int pass = 0; pass < 2; ++pass) {   // 0: writes, 1: reads
            std::fill(seen.begin(), seen.end(), 0);
            for (int tid = 0; tid < T; ++tid) {
                const int w = tid / 64;
                for (int k = 0; k < E; ++k) {
                    int a;
                    if (fwd) a = pass == 0 ? B::template xaddr<P, P, true>(tid, k) : B::template xaddr<P, P + 1, true>(tid, k);
                    else a = pass == 0 ? B::template xaddr<P, P + 1, false>(tid, k) : B::template xaddr<P, P, false>(tid, k);
                    if (a < 0 || a >= words) return -3;
                    if (seen[a]++) return -3;                       // injective
                    const bool own = a >= w * S && a < (w + 1) * S;
                    if (B::G::exch_wave_local(P)) {
                        if (!own) return -1;
                        if (pass == 0) owner_w[a] = w; else if (owner_w[a] != w) return -1;
                    } else {
                        // forward: the LAST cross-wave exchange is read inside the own region (the wave-local exchanges that follow
                        // need no barrier); inverse: the FIRST one met (highest index) is written inside the own region
                        if (fwd && pass == 1 && !own && P == B::G::highest_cross_wave_exchange()) return -2;
                        if (!fwd && pass == 0 && !own && P == B::G::highest_cross_wave_exchange()) return -2;
                        if (pass == 0) owner_w[a] = w; else if (owner_w[a] < 0) return -3;   // every word read was written
                    }
                }
            }
        }
    }
    (void)W;
    if constexpr (P + 2 < B::NPH) return check_exchanges<B, P + 1>();
    return 0;
}
template <int LOGN, int LOGE>
static int check_geo() {
    typedef NttBody<FoldArith, LOGN, LOGE> B;
    if constexpr (B::NPH >= 2) {
        // cross-wave exchanges come first (forward order): everything after the last of them stays inside a wave
        for (int p = 0; p + 1 < B::NPH; ++p) if (!B::G::exch_wave_local(p) && p > B::G::highest_cross_wave_exchange()) return -4;
        // the kLdsIO rows of a wave lie in its region
        for (int tid = 0; tid < B::T; ++tid) {
            const int r = B::G::lds_row(tid), w = tid / 64;
            if (r < w * B::G::kWaveStride || r + E_of<B>() + 2 > (w + 1) * B::G::kWaveStride) return -5;
        }
        return check_exchanges<B, 0>();
    }
    return 0;
}

template <class Arith, int LOGN, int LOGE>
static int emu(int inverse, u64 q, u64 psi, const u64* in, u64* out) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    HostLimbTables t;
    int rc = build_limb_tables(LOGN, q, psi, t);
    if (rc) return rc;
    if (!class_ok<Arith>(q)) return 2000;
    const u64 ninv = t.lc.ninv;     // (F64Arith's LimbConst reuses the field)
    t.lc = class_lc<Arith>(t.lc);
    constexpr int E = B::E, T = B::T;
    std::vector<u64> regs((size_t)T * E), lds(B::G::lds_words(), 0xDEADBEEFDEADBEEFull);
    auto X = [&](int tid) -> u64(&)[E] { return *reinterpret_cast<u64(*)[E]>(&regs[(size_t)tid * E]); };
    if (!inverse) {
        auto tw = TwTab<Arith>::make(t.rp, t.rp_sh, q);
        permute_window0(tw, LOGN, LOGE, B::G::kPermStages);
        for (int tid = 0; tid < T; ++tid) B::load_top(tid, X(tid), in);
        FwdSteps<B, 0>::run(regs, lds, tw.data(), t.lc);
        for (int tid = 0; tid < T; ++tid) { B::fwd_canon(X(tid), t.lc); B::store_bot(tid, X(tid), out); }
    } else {
        auto tw = TwTab<Arith>::make(t.irp, t.irp_sh, q);
        permute_window0(tw, LOGN, LOGE, B::G::kPermStages);
        auto wl = TwTab<Arith>::one(t.w_last, t.w_last_sh, q), wn = TwTab<Arith>::one(ninv, h_shoup(ninv, q), q);
        for (int tid = 0; tid < T; ++tid) B::load_bot(tid, X(tid), in);
        InvSteps<B, B::NPH - 1, kUnit>::run(regs, lds, tw.data(), wl, wn, t.lc);
        for (int tid = 0; tid < T; ++tid) { B::inv_canon(X(tid), t.lc); B::store_top(tid, X(tid), out); }
    }
    return 0;
}

// arith: 0 Shoup, 1 Fold, 2 F64, 3 FoldScaled, 4 F64Wide (tables.h LimbClass).  `in`/`out` must be 16-byte aligned.  returns 0, 2000 bad args, -1 unsupported geometry
extern "C" int emu_ntt(int arith, int log2n, int loge, int inverse, u64 q, u64 psi, const u64* in, u64* out) {
#define CASE(LN, LE)                                                                     \
    if (log2n == LN && loge == LE) {                                                     \
        if (arith == 2) return emu<F64Arith, LN, LE>(inverse, q, psi, in, out);          \
        if (arith == 4) return emu<F64WideArith, LN, LE>(inverse, q, psi, in, out);      \
        if (arith == 3) return emu<FoldScaledArith, LN, LE>(inverse, q, psi, in, out);   \
        return arith ? emu<FoldArith, LN, LE>(inverse, q, psi, in, out) : emu<ShoupArith, LN, LE>(inverse, q, psi, in, out); \
    }
    CASE(8, 4) CASE(10, 4) CASE(11, 4) CASE(12, 4) CASE(13, 5) CASE(14, 5) CASE(14, 4) CASE(12, 3) CASE(12, 5) CASE(13, 4) CASE(6, 3)
#undef CASE
    return -1;
}

// The GENERIC fused multiply's data path (kernels.h ct_mul_kernel, policies without lazy products): four forward transforms to canonical
// words, Arith::mul_var products, three inverse transforms - for the round-6 policies, whose conversions (Arith::enter / leave) sit inside.
template <class Arith, int LOGN, int LOGE>
static int emu_ct_mul_generic(u64 q, u64 psi, const u64* a0, const u64* a1, const u64* b0, const u64* b1, u64* out3) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    HostLimbTables t;
    int rc = build_limb_tables(LOGN, q, psi, t);
    if (rc) return rc;
    if (!class_ok<Arith>(q)) return 2000;
    const u64 ninv = t.lc.ninv;
    const LimbConst lc = class_lc<Arith>(t.lc);
    constexpr int E = B::E, T = B::T, N = B::G::N;
    auto twf = TwTab<Arith>::make(t.rp, t.rp_sh, q), twi = TwTab<Arith>::make(t.irp, t.irp_sh, q);
    permute_window0(twf, LOGN, LOGE, B::G::kPermStages);
    permute_window0(twi, LOGN, LOGE, B::G::kPermStages);
    const auto wl = TwTab<Arith>::one(t.w_last, t.w_last_sh, q), wn = TwTab<Arith>::one(ninv, h_shoup(ninv, q), q);
    std::vector<u64> lds(B::G::lds_words());
    auto fwd = [&](const u64* src) {
        std::vector<u64> regs((size_t)T * E);
        auto X = [&](int tid) -> u64(&)[E] { return *reinterpret_cast<u64(*)[E]>(&regs[(size_t)tid * E]); };
        for (int tid = 0; tid < T; ++tid) B::load_top(tid, X(tid), src);
        FwdSteps<B, 0>::run(regs, lds, twf.data(), lc);
        for (int tid = 0; tid < T; ++tid) B::fwd_canon(X(tid), lc);
        return regs;
    };
    auto inv = [&](std::vector<u64> regs, u64* dst) {
        auto X = [&](int tid) -> u64(&)[E] { return *reinterpret_cast<u64(*)[E]>(&regs[(size_t)tid * E]); };
        InvSteps<B, B::NPH - 1, kUnit>::run(regs, lds, twi.data(), wl, wn, lc);
        for (int tid = 0; tid < T; ++tid) { B::inv_canon(X(tid), lc); B::store_top(tid, X(tid), dst); }
    };
    std::vector<u64> S0 = fwd(a0), S1 = fwd(b0), S2 = fwd(b1), S3 = fwd(a1), x((size_t)N);
    for (int i = 0; i < N; ++i) x[i] = Arith::mul_var(S0[i], S1[i], lc);
    inv(x, out3);
    for (int i = 0; i < N; ++i) x[i] = add_mod(Arith::mul_var(S0[i], S2[i], lc), Arith::mul_var(S3[i], S1[i], lc), lc.q);
    inv(x, out3 + N);
    for (int i = 0; i < N; ++i) x[i] = Arith::mul_var(S3[i], S2[i], lc);
    inv(x, out3 + 2 * N);
    return 0;
}
extern "C" int emu_ct_mul_class(int arith, int log2n, u64 q, u64 psi, const u64* a0, const u64* a1, const u64* b0, const u64* b1, u64* out3) {
#define CASE(LN)                                                                                                              \
    if (log2n == LN) {                                                                                                        \
        if (arith == 2) return emu_ct_mul_generic<F64Arith, LN, 4>(q, psi, a0, a1, b0, b1, out3);                             \
        if (arith == 4) return emu_ct_mul_generic<F64WideArith, LN, 4>(q, psi, a0, a1, b0, b1, out3);                         \
        if (arith == 3) return emu_ct_mul_generic<FoldScaledArith, LN, 4>(q, psi, a0, a1, b0, b1, out3);                      \
        if (arith == 0) return emu_ct_mul_generic<ShoupArith, LN, 4>(q, psi, a0, a1, b0, b1, out3);                           \
        return -1;                                                                                                            \
    }
    CASE(8) CASE(10) CASE(12) CASE(13)
#undef CASE
    return -1;
}

// Split transform (N = 2^15, 2^16): ntt_top.h column stages + N1 emulated 4096-point kernels on sub-tree tables, the way
// launch_impl.h launch_ntt_split and dpfhe_ctx_create arrange them.
template <class Arith, int LOG_N1>
static int emu_split(int inverse, u64 q, u64 psi, const u64* in, u64* out) {
    constexpr int LN2 = 12, LE = 4, N1 = 1 << LOG_N1, LOGN = LN2 + LOG_N1;
    typedef NttBody<Arith, LN2, LE> B;
    typedef typename Arith::Tw Tw;
    HostLimbTables t;
    int rc = build_limb_tables(LOGN, q, psi, t);
    if (rc) return rc;
    if (Arith::kFold && !fold_eligible(q)) return 2000;
    constexpr int E = B::E, T = B::T, N2 = B::G::N;
    const size_t n = (size_t)1 << LOGN;
    std::vector<Tw> top_f(N1), top_i(N1);
    for (int i = 1; i < N1; ++i) { top_f[i] = h_make_tw<Tw>(t.rp[i], q); top_i[i] = h_make_tw<Tw>(t.irp[i], q); }
    // (dpfhe_cabi.hip: FoldArith sub-transforms divide by their own length in their last stage, so the column stage carries N1^-1 only)
    const u64 up = Arith::kFold ? ((u64)N2 % q) : 1;
    const InvLast<Tw> top_last{h_make_tw<Tw>(h_mulmod(t.w_last, up, q), q), h_make_tw<Tw>(h_mulmod(t.lc.ninv, up, q), q)};
    std::vector<u64> buf(in, in + n), lds(B::G::lds_words());
    auto columns = [&](bool fwd) {
        for (size_t c = 0; c < (size_t)N2; ++c) {
            u64 x[N1];
            for (int r = 0; r < N1; ++r) x[r] = buf[(size_t)r * N2 + c];
            if (fwd) top_forward<Arith, LOG_N1>(x, top_f.data(), t.lc); else top_inverse<Arith, LOG_N1>(x, top_i.data(), top_last, t.lc);
            for (int r = 0; r < N1; ++r) buf[(size_t)r * N2 + c] = x[r];
        }
    };
    auto blocks = [&](bool fwd) {
        for (size_t r = 0; r < (size_t)N1; ++r) {
            const std::vector<u64> words = subtree_table(fwd ? t.rp : t.irp, LOGN, LOG_N1, r);
            std::vector<Tw> tw(words.size());
            for (size_t i = 0; i < words.size(); ++i) tw[i] = h_make_tw<Tw>(words[i], q);
            permute_window0(tw, LN2, LE, B::G::kPermStages);
            std::vector<u64> regs((size_t)T * E), blk(buf.begin() + r * N2, buf.begin() + (r + 1) * N2), res(N2);
            auto X = [&](int tid) -> u64(&)[E] { return *reinterpret_cast<u64(*)[E]>(&regs[(size_t)tid * E]); };
            if (fwd) {
                for (int tid = 0; tid < T; ++tid) B::load_top(tid, X(tid), blk.data());
                FwdSteps<B, 0>::run(regs, lds, tw.data(), t.lc);
                for (int tid = 0; tid < T; ++tid) { B::fwd_canon(X(tid), t.lc); B::store_bot(tid, X(tid), res.data()); }
            } else {
                // generic primes: no N^-1 inside a block; FoldArith: sums are divided by N2 exactly, differences carry N2^-1 in their twiddle (dpfhe_cabi.hip)
                const u64 n2inv = Arith::kFold ? h_powmod((u64)N2 % q, q - 2, q) : 1;
                const Tw wl = h_make_tw<Tw>(h_mulmod(words[1], n2inv, q), q), wn = h_make_tw<Tw>(1, q);
                for (int tid = 0; tid < T; ++tid) B::load_bot(tid, X(tid), blk.data());
                InvSteps<B, B::NPH - 1, kUnit>::run(regs, lds, tw.data(), wl, wn, t.lc);
                for (int tid = 0; tid < T; ++tid) { B::inv_canon(X(tid), t.lc); B::store_top(tid, X(tid), res.data()); }
            }
            std::copy(res.begin(), res.end(), buf.begin() + r * N2);
        }
    };
    if (!inverse) { columns(true); blocks(true); } else { blocks(false); columns(false); }
    std::copy(buf.begin(), buf.end(), out);
    return 0;
}

extern "C" int emu_ntt_split(int arith, int log2n, int inverse, u64 q, u64 psi, const u64* in, u64* out) {
    if (log2n == 15) return arith ? emu_split<FoldArith, 3>(inverse, q, psi, in, out) : emu_split<ShoupArith, 3>(inverse, q, psi, in, out);
    if (log2n == 16) return arith ? emu_split<FoldArith, 4>(inverse, q, psi, in, out) : emu_split<ShoupArith, 4>(inverse, q, psi, in, out);
    return -1;
}

// Forward transform of words that are only known to be below 2^60 (the digits of a key switch are canonical for ANOTHER limb): NttBody's
// FWD_IN = kRedB plans, which relin_kernel / relin_shared_kernel rely on to skip the canonicalisation.  FoldArith only.
template <int LOGN, int LOGE>
static int emu_fwd_any60(u64 q, u64 psi, const u64* in, u64* out) {
    typedef NttBody<FoldArith, LOGN, LOGE, 0, kRedB> B;
    HostLimbTables t;
    int rc = build_limb_tables(LOGN, q, psi, t);
    if (rc) return rc;
    if (!fold_eligible(q)) return 2000;
    constexpr int E = B::E, T = B::T;
    std::vector<u64> regs((size_t)T * E), lds(B::G::lds_words(), 0xDEADBEEFDEADBEEFull);
    auto X = [&](int tid) -> u64(&)[E] { return *reinterpret_cast<u64(*)[E]>(&regs[(size_t)tid * E]); };
    auto tw = TwTab<FoldArith>::make(t.rp, t.rp_sh, q);
    permute_window0(tw, LOGN, LOGE, B::G::kPermStages);
    for (int tid = 0; tid < T; ++tid) B::load_top(tid, X(tid), in);
    FwdSteps<B, 0>::run(regs, lds, tw.data(), t.lc);
    for (int tid = 0; tid < T; ++tid) { B::fwd_canon(X(tid), t.lc); B::store_bot(tid, X(tid), out); }
    return 0;
}
extern "C" int emu_ntt_fwd_any60(int log2n, u64 q, u64 psi, const u64* in, u64* out) {
    if (log2n == 8) return emu_fwd_any60<8, 4>(q, psi, in, out);
    if (log2n == 10) return emu_fwd_any60<10, 4>(q, psi, in, out);
    if (log2n == 12) return emu_fwd_any60<12, 4>(q, psi, in, out);
    if (log2n == 13) return emu_fwd_any60<13, 4>(q, psi, in, out);
    return -1;
}

// N = 8192 as a column stage in registers + two 4096-point sub-transforms through ONE LDS buffer (ntt_halves.h; kernels_halves.h runs
// exactly these steps on the device): 256 emulated threads hold lo / hi, the sub-transforms run on the tables dpfhe_ctx_create builds
// for them (tables.h subtree_table, roots 2 and 3 of the N = 8192 table).
template <class Arith>
static int emu_halves(int inverse, u64 q, u64 psi, const u64* in, u64* out) {
    typedef Halves13<Arith> H;
    typedef typename H::B B;
    typedef typename Arith::Tw Tw;
    HostLimbTables t;
    int rc = build_limb_tables(H::LOGN, q, psi, t);
    if (rc) return rc;
    if (Arith::kFold && !fold_eligible(q)) return 2000;
    constexpr int E = B::E, T = B::T, N2 = H::N2;
    std::vector<u64> lo((size_t)T * E), hi((size_t)T * E), lds(B::G::lds_words(), 0xDEADBEEFDEADBEEFull);
    auto X = [&](std::vector<u64>& r, int tid) -> u64(&)[E] { return *reinterpret_cast<u64(*)[E]>(&r[(size_t)tid * E]); };
    std::vector<Tw> tw[2];
    for (size_t r = 0; r < 2; ++r) {
        const std::vector<u64> words = subtree_table(inverse ? t.irp : t.rp, H::LOGN, 1, r);
        tw[r].resize(words.size());
        for (size_t i = 0; i < words.size(); ++i) tw[r][i] = h_make_tw<Tw>(words[i], q);
        permute_window0(tw[r], H::LOGN2, H::LOGE, B::G::kPermStages);
    }
    if (!inverse) {
        const Tw wtop = h_make_tw<Tw>(t.rp[1], q);
        for (int tid = 0; tid < T; ++tid) { B::load_top(tid, X(lo, tid), in); B::load_top(tid, X(hi, tid), in + N2); H::fwd_column(X(lo, tid), X(hi, tid), wtop, t.lc); }
        FwdSteps<B, 0>::run(lo, lds, tw[0].data(), t.lc);
        for (int tid = 0; tid < T; ++tid) { B::fwd_canon(X(lo, tid), t.lc); B::store_bot(tid, X(lo, tid), out); }
        FwdSteps<B, 0>::run(hi, lds, tw[1].data(), t.lc);
        for (int tid = 0; tid < T; ++tid) { B::fwd_canon(X(hi, tid), t.lc); B::store_bot(tid, X(hi, tid), out + N2); }
    } else {
        const InvLast<Tw> last{h_make_tw<Tw>(t.w_last, q), h_make_tw<Tw>(t.lc.ninv, q)};
        const Tw unused = h_make_tw<Tw>(1, q);
        for (int tid = 0; tid < T; ++tid) B::load_bot(tid, X(lo, tid), in);
        InvSteps<B, B::NPH - 1, kUnit>::run(lo, lds, tw[0].data(), unused, unused, t.lc);
        for (int tid = 0; tid < T; ++tid) B::load_bot(tid, X(hi, tid), in + N2);
        InvSteps<B, B::NPH - 1, kUnit>::run(hi, lds, tw[1].data(), unused, unused, t.lc);
        for (int tid = 0; tid < T; ++tid) {
            H::inv_column(X(lo, tid), X(hi, tid), last, t.lc);
            B::inv_canon(X(lo, tid), t.lc); B::inv_canon(X(hi, tid), t.lc);
            B::store_top(tid, X(lo, tid), out); B::store_top(tid, X(hi, tid), out + N2);
        }
    }
    return 0;
}
// N = 16384 as two column stages in registers + four 4096-point sub-transforms through ONE LDS buffer (ntt_quarters.h; kernels_quarters.h runs exactly these
// steps on the device): 256 emulated threads hold q0..q3, the sub-transforms run on the sub-tree tables rooted at nodes 4..7.  FoldArith.
extern "C" int emu_ntt_quarters(int inverse, u64 q, u64 psi, const u64* in, u64* out) {
    typedef Quarters14 Q;
    typedef Q::B B;
    HostLimbTables t;
    int rc = build_limb_tables(Q::LOGN, q, psi, t);
    if (rc) return rc;
    if (!fold_eligible(q)) return 2000;
    constexpr int E = B::E, T = B::T, N2 = Q::N2;
    std::vector<u64> r[4], lds(B::G::lds_words(), 0xDEADBEEFDEADBEEFull);
    for (auto& v : r) v.assign((size_t)T * E, 0);
    auto X = [&](std::vector<u64>& v, int tid) -> u64(&)[E] { return *reinterpret_cast<u64(*)[E]>(&v[(size_t)tid * E]); };
    std::vector<TwFold> tw[4];
    for (size_t i = 0; i < 4; ++i) {
        const std::vector<u64> words = subtree_table(inverse ? t.irp : t.rp, Q::LOGN, 2, i);
        tw[i].resize(words.size());
        for (size_t j = 0; j < words.size(); ++j) tw[i][j] = h_tw_fold(words[j], q);
        permute_window0(tw[i], Q::LOGN2, Q::LOGE, B::G::kPermStages);
    }
    if (!inverse) {
        const QuartersTop top{h_tw_fold(t.rp[1], q), h_tw_fold(t.rp[2], q), h_tw_fold(t.rp[3], q)};
        for (int tid = 0; tid < T; ++tid) {
            for (int i = 0; i < 4; ++i) B::load_top(tid, X(r[i], tid), in + (size_t)i * N2);
            Q::fwd_columns(X(r[0], tid), X(r[1], tid), X(r[2], tid), X(r[3], tid), top, t.lc);
        }
        for (int i = 0; i < 4; ++i) {
            FwdSteps<B, 0>::run(r[i], lds, tw[i].data(), t.lc);
            for (int tid = 0; tid < T; ++tid) { B::fwd_canon(X(r[i], tid), t.lc); B::store_bot(tid, X(r[i], tid), out + (size_t)i * N2); }
        }
    } else {
        const InvLast<TwFold> last{h_tw_fold(t.w_last, q), h_tw_fold(t.lc.ninv, q)};
        const TwFold wi2 = h_tw_fold(t.irp[2], q), wi3 = h_tw_fold(t.irp[3], q), unused = h_tw_fold(1, q);
        for (int i = 0; i < 4; ++i) {
            for (int tid = 0; tid < T; ++tid) B::load_bot(tid, X(r[i], tid), in + (size_t)i * N2);
            InvSteps<B, B::NPH - 1, kUnit>::run(r[i], lds, tw[i].data(), unused, unused, t.lc);
        }
        for (int tid = 0; tid < T; ++tid) {
            Q::inv_columns(X(r[0], tid), X(r[1], tid), X(r[2], tid), X(r[3], tid), wi2, wi3, last, t.lc);
            for (int i = 0; i < 4; ++i) { B::inv_canon(X(r[i], tid), t.lc); B::store_top(tid, X(r[i], tid), out + (size_t)i * N2); }
        }
    }
    return 0;
}

extern "C" int emu_ntt_halves(int arith, int inverse, u64 q, u64 psi, const u64* in, u64* out) {
    return arith ? emu_halves<FoldArith>(inverse, q, psi, in, out) : emu_halves<ShoupArith>(inverse, q, psi, in, out);
}

// The fused ct x ct kernels' lazy FoldArith data path (kernels.h ct_mul_kernel / ct_mul_dual_kernel - the paired kernel computes
// the same values from the same operands, only two transforms at a time -, coefficient domain in and out), run with
// the same per-thread transform code and the same dyadic sequence, so that the bound plans and the relaxed mul60
// precondition (lazy forward outputs < 14 q times partially reduced b-side factors) are checked on the CPU with the
// wrap-around / precondition counters armed.  out3 = (c0, c1, c2) of one limb.
template <class Arith, int LOGN, int LOGE>
static int emu_ct_mul_lazy(u64 q, u64 psi, const u64* a0, const u64* a1, const u64* b0, const u64* b1, u64* out3) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    typedef NttBody<Arith, LOGN, LOGE, 0, kUnit, true> BI;   // inverse of register-resident products (kernels.h ct_mul_quad_kernel / ct_mul_dual_kernel)
    HostLimbTables t;
    int rc = build_limb_tables(LOGN, q, psi, t);
    if (rc) return rc;
    if (!class_ok<Arith>(q)) return 2000;
    constexpr int E = B::E, T = B::T, N = B::G::N;
    static_assert(!Arith::kFoldCore || B::kFwdOutBound <= kLimitPartner, "lazy forward outputs must satisfy mul60's bound");
    auto twf = TwTab<Arith>::make(t.rp, t.rp_sh, q), twi = TwTab<Arith>::make(t.irp, t.irp_sh, q);
    permute_window0(twf, LOGN, LOGE, B::G::kPermStages);
    permute_window0(twi, LOGN, LOGE, B::G::kPermStages);
    // FoldScaledArith: the products carry the scale twice; the inverse's last stage folds one s^-1 in (DevTables::last2)
    const u64 sinv = (Arith::kFoldCore && !Arith::kFold) ? h_powmod((1ull << fold_scaled_shift(q)) % q, q - 2, q) : 1;
    const u64 wlv = h_mulmod(t.w_last, sinv, q), wnv = h_mulmod(t.lc.ninv, sinv, q);
    const auto wl = TwTab<Arith>::one(wlv, h_shoup(wlv, q), q), wn = TwTab<Arith>::one(wnv, h_shoup(wnv, q), q);
    const LimbConst lc = class_lc<Arith>(t.lc);
    std::vector<u64> lds(B::G::lds_words());
    auto fwd = [&](const u64* src, bool partner) {
        std::vector<u64> regs((size_t)T * E);
        auto X = [&](int tid) -> u64(&)[E] { return *reinterpret_cast<u64(*)[E]>(&regs[(size_t)tid * E]); };
        for (int tid = 0; tid < T; ++tid) B::load_top(tid, X(tid), src);
        FwdSteps<B, 0>::run(regs, lds, twf.data(), lc);
        if (partner) for (int tid = 0; tid < T; ++tid) B::prod_partner(X(tid), lc);
        return regs;
    };
    auto inv = [&](std::vector<u64> regs, u64* dst) {
        auto X = [&](int tid) -> u64(&)[E] { return *reinterpret_cast<u64(*)[E]>(&regs[(size_t)tid * E]); };
        InvSteps<BI, B::NPH - 1, B::kProdInvIn>::run(regs, lds, twi.data(), wl, wn, lc);
        for (int tid = 0; tid < T; ++tid) { B::inv_canon(X(tid), lc); B::store_top(tid, X(tid), dst); }
    };
    // the tensor step as ct_mul_quad_kernel / ct_mul_dual_kernel run it (NttBody::tensor: fold policies turn b0, b1 into twiddles on the fly)
    std::vector<u64> S0 = fwd(a0, false), S1 = fwd(b0, true), S2 = fwd(b1, true), S3 = fwd(a1, false), c0((size_t)N), c1((size_t)N), c2((size_t)N);
    for (int i = 0; i < N; ++i) B::tensor(S0[i], S3[i], S1[i], S2[i], c0[i], c1[i], c2[i], lc);
    inv(c0, out3);
    inv(c1, out3 + N);
    inv(c2, out3 + 2 * N);
    return 0;
}
template <int LOGN, int LOGE>
static int emu_ct_mul_fold(u64 q, u64 psi, const u64* a0, const u64* a1, const u64* b0, const u64* b1, u64* out3) {
    return emu_ct_mul_lazy<FoldArith, LOGN, LOGE>(q, psi, a0, a1, b0, b1, out3);
}
// the lazy-product path of the round-6 classes (arith 2 F64, 3 FoldScaled; 1 Fold)
extern "C" int emu_ct_mul_lazy_class(int arith, int log2n, u64 q, u64 psi, const u64* a0, const u64* a1, const u64* b0, const u64* b1, u64* out3) {
#define CASE(LN)                                                                                                           \
    if (log2n == LN) {                                                                                                     \
        if (arith == 1) return emu_ct_mul_lazy<FoldArith, LN, 4>(q, psi, a0, a1, b0, b1, out3);                            \
        if (arith == 2) return emu_ct_mul_lazy<F64Arith, LN, 4>(q, psi, a0, a1, b0, b1, out3);                             \
        if (arith == 4) return emu_ct_mul_lazy<F64WideArith, LN, 4>(q, psi, a0, a1, b0, b1, out3);                         \
        if (arith == 3) return emu_ct_mul_lazy<FoldScaledArith, LN, 4>(q, psi, a0, a1, b0, b1, out3);                      \
        return -1;                                                                                                         \
    }
    CASE(8) CASE(10) CASE(12) CASE(13)
#undef CASE
    return -1;
}

extern "C" int emu_ct_mul(int log2n, u64 q, u64 psi, const u64* a0, const u64* a1, const u64* b0, const u64* b1, u64* out3) {
    if (log2n == 8) return emu_ct_mul_fold<8, 4>(q, psi, a0, a1, b0, b1, out3);
    if (log2n == 10) return emu_ct_mul_fold<10, 4>(q, psi, a0, a1, b0, b1, out3);
    if (log2n == 12) return emu_ct_mul_fold<12, 4>(q, psi, a0, a1, b0, b1, out3);
    if (log2n == 13) return emu_ct_mul_fold<13, 4>(q, psi, a0, a1, b0, b1, out3);
    return -1;
}

extern "C" int emu_check_lds_regions(int log2n, int loge) {
#define CASE(LN, LE) if (log2n == LN && loge == LE) return check_geo<LN, LE>();
    CASE(8, 4) CASE(9, 4) CASE(10, 4) CASE(11, 4) CASE(12, 4) CASE(13, 4) CASE(14, 4) CASE(13, 5) CASE(14, 5) CASE(12, 3) CASE(12, 5)
#undef CASE
    return -100;
}

extern "C" int emu_lds_words(int log2n, int loge) {
    if (log2n == 12 && loge == 4) return Geo<12, 4>::lds_words();
    if (log2n == 13 && loge == 5) return Geo<13, 5>::lds_words();
    if (log2n == 10 && loge == 4) return Geo<10, 4>::lds_words();
    return -1;
}

// FoldArith::prod_tw / mul_ptw_add (variable x variable products through the twiddle chain): addend + y b mod q through the very code the
// kernels run, canonicalised; d is passed so that the scaled-fold moduli (q' = 2^60 - d, not prime) can be checked as well.
extern "C" u64 emu_fold_ptw(u64 d, u64 y, u64 b, u64 addend) {
    LimbConst lc{};
    lc.q = (1ull << 60) - d; lc.d = d;
    return FoldArith::canon(FoldArith::mul_ptw_add(y, FoldArith::prod_tw(b, lc), lc, addend), lc);
}

// FoldArith::dot30_* (the plaintext matvec's column accumulators): dot product of canonical residues through the very code
// the kernel runs (fold every kDot30Period terms, the running word riding in column 0), against 128-bit arithmetic.
extern "C" u64 emu_dot30(u64 q, const u64* a, const u64* b, size_t n) {
    LimbConst lc{};
    lc.q = q; lc.d = (1ull << 60) - q;
    FoldArith::Dot30 acc{0, 0, 0};
    int since = 0;
    for (size_t i = 0; i < n; ++i) {
        FoldArith::dot30_mac(acc, FoldArith::split30(a[i]), FoldArith::split30(b[i]));
        if (++since == FoldArith::kDot30Period) { acc = FoldArith::Dot30{FoldArith::dot30_fold0(acc, lc), 0, 0}; since = 0; }
    }
    return FoldArith::canon_small(since ? FoldArith::dot30_fold0(acc, lc) : acc.s0, lc);
}
