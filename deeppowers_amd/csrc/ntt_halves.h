// ntt_halves.h - an N = 8192 transform as ONE radix-2 column stage in registers + TWO 4096-point sub-transforms (host + device).
//
// SURVEY.md section 8(a) rows A1/A2 at BASELINE configs[4]'s ring degree (no reference counterpart - section 0).  The merged
// Cooley-Tukey transform's first stage pairs word j with word j + N/2 under ONE twiddle (psi^brv(1)); after it the two halves
// are independent 4096-point transforms on the sub-trees of the twiddle table rooted at nodes 2 and 3 (tables.h subtree_table) -
// the same cut ntt_top.h makes for N > 16384, but here both parts live in one 256-thread workgroup: a thread holds
// lo[k] = a[k 256 + tid] and hi[k] = a[4096 + k 256 + tid], runs the column stage on its 16 pairs, and the two sub-transforms go
// through NttBody<Arith, 12, 4, SUB = 1> one after the other THROUGH ONE 38 KiB LDS BUFFER.  Against the 512-thread geometry
// (Geo<13, 4>: four phases, 80 KiB of LDS, an 8-wave barrier) a CU holds three to four independent 4-wave workgroups that
// de-phase like the N = 4096 kernels do, and a word makes two LDS round trips instead of three.
// The inverse runs the sub-transforms first (no N^-1 inside: NttBody SUB mode) and the column stage last, N^-1 = 2^-13 folded into
// it (FoldArith: exact division of the sums, modarith.h mul_ninv).
// Results are the words of the single-kernel transform: same merged butterfly network, same twiddles, canonical outputs.
#pragma once
#include "devtables.h"
#include "ntt_core.h"

namespace dpfhe {

template <class Arith>
struct Halves13 {
    static constexpr int LOGN = 13, LOGN2 = 12, LOGE = 4, N = 1 << LOGN, N2 = 1 << LOGN2;
    typedef NttBody<Arith, LOGN2, LOGE, 1> B;
    typedef typename Arith::Tw Tw;
    static constexpr int E = B::E, T = B::T;
    static_assert(T == 256 && E == 16, "256 threads x (16 + 16) words");

    // forward column stage on canonical inputs: lo' = lo + w hi (reduced), hi' = lo - w hi (< 4 q: NttBody's SUB input bound)
    static DPF_HD void fwd_column(u64 (&lo)[E], u64 (&hi)[E], const Tw& w, const LimbConst& lc) {
        const u64 two_q = 2 * lc.q;
#pragma clang loop unroll(full)
        for (int k = 0; k < E; ++k) {
            if constexpr (Arith::kFold) {
                const u64 a = lo[k];
                const u64 s = FoldArith::mul_tw_add(hi[k], w, lc, a);
                lo[k] = s;
                hi[k] = chk_shl1_add_sub(a, two_q, s);
            } else {
                const u64 a = lo[k];                             // canonical: inside Harvey's [0, 2q)
                const u64 t = Arith::mul_tw(hi[k], w, lc);
                lo[k] = a + t;
                hi[k] = a - t + two_q;
            }
        }
    }
    // only ONE of the two outputs (a workgroup that owns one half of the NTT domain: kernels_halves.h relin_half_kernel)
    template <int HALF>
    static DPF_HD void fwd_column_half(u64 (&x)[E], const u64 (&lo)[E], const u64 (&hi)[E], const Tw& w, const LimbConst& lc) {
        const u64 two_q = 2 * lc.q;
#pragma clang loop unroll(full)
        for (int k = 0; k < E; ++k) {
            if constexpr (Arith::kFold) {
                const u64 a = lo[k];
                const u64 s = FoldArith::mul_tw_add(hi[k], w, lc, a);
                x[k] = HALF ? chk_shl1_add_sub(a, two_q, s) : s;
            } else {
                const u64 a = lo[k];
                const u64 t = Arith::mul_tw(hi[k], w, lc);
                x[k] = HALF ? a - t + two_q : a + t;
            }
        }
    }
    // inverse column stage on the sub-transforms' outputs (FoldArith: < kSubInvOut q/1024; generic primes: [0, 2q)), N^-1 folded in
    static DPF_HD void inv_column(u64 (&lo)[E], u64 (&hi)[E], const InvLast<Tw>& last, const LimbConst& lc) {
        const u64 two_q = 2 * lc.q;
#pragma clang loop unroll(full)
        for (int k = 0; k < E; ++k) {
            const u64 a = lo[k], b = hi[k];
            if constexpr (Arith::kFold) {
                const u64 s = chk_add(a, b);
                const u64 dlt = chk_sub_add(a, b, (u64)(kSubInvOut / kUnit) * lc.q);
                lo[k] = FoldArith::mul_ninv(s, lc, LOGN);          // (a + b) / N exactly: < q + 14 q / 8192 + 1
                hi[k] = FoldArith::mul_tw(dlt, last.w_last, lc);   // (a - b) psi^-brv(1) N^-1
            } else {
                const u64 s = csub(a + b, two_q);
                const u64 dlt = a - b + two_q;
                lo[k] = Arith::mul_tw(s, last.w_ninv, lc);
                hi[k] = Arith::mul_tw(dlt, last.w_last, lc);
            }
        }
    }
};

}  // namespace dpfhe
