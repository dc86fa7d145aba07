// ntt_top.h - the first (forward) / last (inverse) log2(N1) radix-2 stages of a split negacyclic transform (host + device).
//
// Ring degrees above 16384 do not fit one workgroup's LDS.  The merged Cooley-Tukey transform is cut after its first
// LOG_N1 stages: those pair words N2 = N / N1 or more apart and use twiddles that depend on the stage and group only, so one
// thread can run them on the N1 words of a column {c, c + N2, c + 2 N2, ...} with workgroup-uniform twiddles, streaming
// the polynomial once.  The remaining stages are N1 independent N2-point transforms (kernels.h, sub-tree tables).
// The inverse runs the two parts in the opposite order, N^-1 folded into the very last stage.
#pragma once
#include "devtables.h"
#include "ntt_core.h"

namespace dpfhe {

// natural order in, the in-place Cooley-Tukey positions out; twt[m + i] = psi^brv(m + i) of the FULL table
template <class Arith, int LOG_N1>
DPF_HD void top_forward(u64 (&x)[1 << LOG_N1], const typename Arith::Tw* twt, const LimbConst& lc) {
    constexpr int N1 = 1 << LOG_N1;
    const u64 two_q = 2 * lc.q;
#pragma clang loop unroll(full)
    for (int s = 0; s < LOG_N1; ++s) {
        const int half = N1 >> (s + 1);
#pragma clang loop unroll(full)
        for (int r = 0; r < N1; ++r) {
            if (r & half) continue;
            const typename Arith::Tw w = twt[(1 << s) + (r >> (LOG_N1 - s))];
            u64 a = x[r];
            if (!Arith::kFold) a = csub(a, two_q);
            const u64 t = Arith::mul_tw(x[r | half], w, lc);
            x[r] = chk_add(a, t);                       // FoldArith: grows by 2 q per stage, 1 + 2 LOG_N1 <= 9 q
            x[r | half] = chk_sub_add(a, t, two_q);
        }
    }
#pragma clang loop unroll(full)
    for (int r = 0; r < N1; ++r) x[r] = Arith::kFold ? FoldArith::canon(x[r], lc) : csub(csub(x[r], two_q), lc.q);
}

// the N2-point inverse transforms have run (canonical words, no N^-1 yet); twt[m + i] = psi^-brv(m + i)
template <class Arith, int LOG_N1>
DPF_HD void top_inverse(u64 (&x)[1 << LOG_N1], const typename Arith::Tw* twt, const InvLast<typename Arith::Tw>& last, const LimbConst& lc) {
    constexpr int N1 = 1 << LOG_N1;
    const u64 q = lc.q, two_q = 2 * lc.q;
#pragma clang loop unroll(full)
    for (int s = LOG_N1 - 1; s >= 0; --s) {
        const int half = N1 >> (s + 1);
        // FoldArith bounds (units of q): sums double per stage from 1; the subtrahend of stage s is below 2^(LOG_N1-1-s)
        const u64 off = (u64)(1 << (LOG_N1 - 1 - s)) * q;
#pragma clang loop unroll(full)
        for (int r = 0; r < N1; ++r) {
            if (r & half) continue;
            const u64 a = x[r], b = x[r | half];
            u64 sum, dlt;
            if (Arith::kFold) { sum = chk_add(a, b); dlt = chk_sub_add(a, b, off); }
            else { sum = csub(a + b, two_q); dlt = a - b + two_q; }
            if (s == 0) {
                x[r] = Arith::mul_tw(sum, last.w_ninv, lc);
                x[r | half] = Arith::mul_tw(dlt, last.w_last, lc);
            } else {
                x[r] = sum;
                x[r | half] = Arith::mul_tw(dlt, twt[(1 << s) + (r >> (LOG_N1 - s))], lc);
            }
        }
    }
#pragma clang loop unroll(full)
    for (int r = 0; r < N1; ++r) x[r] = Arith::kFold ? FoldArith::canon_small(x[r], lc) : csub(x[r], q);
}

#if defined(__HIPCC__)
// one thread per column; grid = residue polynomials x (N2 / 256)
template <class Arith, int LOG_N1, bool FWD>
__global__ __launch_bounds__(256) void ntt_top_kernel(u64* __restrict__ out, const u64* __restrict__ in, DevTables<Arith> tb, int log2_n2) {
    constexpr int N1 = 1 << LOG_N1;
    const unsigned blocks_per_poly = (1u << log2_n2) / 256u;
    const size_t p = blockIdx.x / blocks_per_poly;
    const unsigned c = (blockIdx.x % blocks_per_poly) * 256u + threadIdx.x;
    const int limb = (int)(p % (size_t)tb.n_limbs);
    const LimbConst lc = tb.lc[limb];
    const size_t base = (p << (log2_n2 + LOG_N1)) + c;
    u64 x[N1];
#pragma clang loop unroll(full)
    for (int r = 0; r < N1; ++r) x[r] = in[base + ((size_t)r << log2_n2)];
    if (FWD) top_forward<Arith, LOG_N1>(x, tb.top_fwd + (size_t)limb * N1, lc);
    else top_inverse<Arith, LOG_N1>(x, tb.top_inv + (size_t)limb * N1, tb.top_last[limb], lc);
#pragma clang loop unroll(full)
    for (int r = 0; r < N1; ++r) out[base + ((size_t)r << log2_n2)] = x[r];
}
#endif

}  // namespace dpfhe
