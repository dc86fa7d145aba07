// ntt_quarters.h - an N = 16384 transform as TWO radix-2 column stages in registers + FOUR 4096-point sub-transforms (host + device; round 6).
//
// SURVEY.md section 8(a) rows A1/A2 at the ring degree of the parameter set with a security margin (N = 16384; no reference counterpart - section 0).
// The one-kernel geometry Geo<14, 4> needs the whole 128 KiB polynomial in LDS: ONE 1024-thread workgroup per CU, whose load, exchange and store phases
// nothing overlaps (34-37 % of HBM peak against 57 % at N = 4096).  Here the merged Cooley-Tukey transform is cut after its first two stages - the
// cut ntt_halves.h makes after one - so that a 256-thread workgroup holds q_r[k] = a[r 4096 + k 256 + tid] (r = 0..3, 64 words per thread), runs the two
// column stages on its 16 quadruples and sends the four independent 4096-point sub-transforms (sub-trees of the twiddle table rooted at nodes 4..7,
// tables.h subtree_table) through NttBody<FoldArith, 12, 4, SUB = 1> one after the other THROUGH ONE 38 KiB LDS BUFFER: two independent workgroups per
// CU that de-phase, where the 1024-thread kernel is alone on its CU.  FoldArith only (the register budget is the fold arithmetic's).
// Results are the words of the single-kernel transform: same merged butterfly network, same twiddles, canonical outputs.
#pragma once
#include "devtables.h"
#include "ntt_core.h"

namespace dpfhe {

struct QuartersTop {   // per limb: the column stages' twiddles (entries 1, 2, 3 of the full table) in FoldArith's format
    TwFold w1, w2, w3;
};

struct Quarters14 {
    static constexpr int LOGN = 14, LOGN2 = 12, LOGE = 4, N = 1 << LOGN, N2 = 1 << LOGN2;
    typedef NttBody<FoldArith, LOGN2, LOGE, 1> B;
    typedef TwFold Tw;
    static constexpr int E = B::E, T = B::T;
    static_assert(T == 256 && E == 16, "256 threads x (4 x 16) words");

    // one fused Cooley-Tukey butterfly (ntt_core.h fwd_phase_r): lo' = lo + w hi reduced (< 2^60 + 16 d), hi' = lo - w hi = 2 lo + 2 q - lo' (< 2 LO + 2 q)
    static DPF_HD void ct(u64& lo, u64& hi, const Tw& w, const LimbConst& lc) {
        const u64 a = lo;
        const u64 s = FoldArith::mul_tw_add(hi, w, lc, a);
        lo = s;
        hi = chk_shl1_add_sub(a, 2 * lc.q, s);
    }
    // forward column stages on canonical inputs.  Stage 1 (distance 8192, twiddle w1): (q0, q2), (q1, q3); stage 2 (distance 4096): (q0, q1) under w2,
    // (q2, q3) under w3.  Bounds (units of q): after stage 1 q0, q1 < ~1, q2, q3 < 4; stage 2 leaves q0, q2 < ~1 and q1 < 2 + 2 = 4; q2 is reduced before
    // its butterfly (3 instructions) so that q3 < 4 as well: every output is below kCtfMid, the input bound of NttBody's SUB mode.
    static DPF_HD void fwd_columns(u64 (&q0)[E], u64 (&q1)[E], u64 (&q2)[E], u64 (&q3)[E], const QuartersTop& t, const LimbConst& lc) {
#pragma clang loop unroll(full)
        for (int k = 0; k < E; ++k) {
            ct(q0[k], q2[k], t.w1, lc);
            ct(q1[k], q3[k], t.w1, lc);
            ct(q0[k], q1[k], t.w2, lc);
            q2[k] = FoldArith::reduce(q2[k], lc);
            ct(q2[k], q3[k], t.w3, lc);
        }
    }
    // inverse column stages on the sub-transforms' outputs (< kSubInvOut q/1024 = 7 q each, no N^-1 yet).  Stage A (distance 4096): sums reduced (they meet
    // another sum in the last stage), differences times psi^-brv(2) / psi^-brv(3); stage B (distance 8192, the transform's last): sums divided by N exactly,
    // differences times psi^-brv(1) N^-1.
    static DPF_HD void inv_columns(u64 (&q0)[E], u64 (&q1)[E], u64 (&q2)[E], u64 (&q3)[E], const Tw& wi2, const Tw& wi3, const InvLast<Tw>& last,
                                   const LimbConst& lc) {
        const u64 off7 = (u64)(kSubInvOut / kUnit) * lc.q, two_q = 2 * lc.q;
#pragma clang loop unroll(full)
        for (int k = 0; k < E; ++k) {
            const u64 a0 = q0[k], b0 = q1[k], a1 = q2[k], b1 = q3[k];
            const u64 s0 = FoldArith::reduce(chk_add(a0, b0), lc), d0 = FoldArith::mul_tw(chk_sub_add(a0, b0, off7), wi2, lc);   // < 2^60 + 16 d each
            const u64 s1 = FoldArith::reduce(chk_add(a1, b1), lc), d1 = FoldArith::mul_tw(chk_sub_add(a1, b1, off7), wi3, lc);
            q0[k] = FoldArith::mul_ninv(chk_add(s0, s1), lc, LOGN);
            q2[k] = FoldArith::mul_tw(chk_sub_add(s0, s1, two_q), last.w_last, lc);
            q1[k] = FoldArith::mul_ninv(chk_add(d0, d1), lc, LOGN);
            q3[k] = FoldArith::mul_tw(chk_sub_add(d0, d1, two_q), last.w_last, lc);
        }
    }
};

}  // namespace dpfhe
