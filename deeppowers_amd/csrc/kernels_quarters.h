// kernels_quarters.h - the N = 16384 batched transforms on the N = 4096 body (device code; included by launch_impl.h; round 6).
//
// SURVEY.md section 8(a) A1/A2 at N = 16384, the ring of the parameter set with a security margin, where every operation is composed from the batched
// transforms (kernels_large.h).  ntt_quarters.h has the arithmetic: two radix-2 column stages in registers, then the four independent 4096-point
// sub-transforms one after the other through ONE 38 KiB LDS buffer, 256 threads (4 waves) per workgroup, two workgroups per CU (64 words per thread stay in
// registers) - where Geo<14, 4>'s 1024-thread workgroup owns its CU alone and nothing overlaps its load, exchange and store phases.  FoldArith contexts.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "ntt_quarters.h"

namespace dpfhe {

constexpr int kQuartersOcc = 2;   // workgroups per CU the register budget is sized for (64 data words + one sub-transform's working set per thread)

template <bool NT = false>
__global__ __launch_bounds__(256, kQuartersOcc) void ntt_fwd_quarters_kernel(u64* __restrict__ out, const u64* __restrict__ in, DevTables<FoldArith> tb) {
    typedef Quarters14 Q;
    typedef Q::B B;
    constexpr int E = Q::E, N = Q::N, N2 = Q::N2;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    int tid = threadIdx.x;
    const size_t p = blockIdx.x;
    const int limb = (int)(p % (size_t)tb.n_limbs);
    const LimbConst lc = tb.lc[limb];
    const TwFold* tw = tb.qfwd + (size_t)limb * N;     // [limb][quarter][N2]
    const QuartersTop top = tb.qtop_fwd[limb];
    u64 q0[E], q1[E], q2[E], q3[E];
    B::template load_top<NT>(tid, q0, in + p * N);
    B::template load_top<NT>(tid, q1, in + p * N + N2);
    B::template load_top<NT>(tid, q2, in + p * N + 2 * N2);
    B::template load_top<NT>(tid, q3, in + p * N + 3 * N2);
    Q::fwd_columns(q0, q1, q2, q3, top, lc);
    FwdChain<B, 0>::template run<false>(tid, q0, lds, tw, lc);
    B::fwd_canon(q0, lc);
    B::template store_bot_lds<NT>(tid, q0, out + p * N, lds);
    asm volatile("" : "+v"(tid));   // every chain fetches its own twiddles where it uses them (kernels.h ct_mul_kernel)
    lds_barrier();                  // the rows above are read by their own wave only; the next chain's first exchange writes every region
    FwdChain<B, 0>::template run<false>(tid, q1, lds, tw + N2, lc);
    B::fwd_canon(q1, lc);
    B::template store_bot_lds<NT>(tid, q1, out + p * N + N2, lds);
    asm volatile("" : "+v"(tid));
    lds_barrier();
    FwdChain<B, 0>::template run<false>(tid, q2, lds, tw + 2 * N2, lc);
    B::fwd_canon(q2, lc);
    B::template store_bot_lds<NT>(tid, q2, out + p * N + 2 * N2, lds);
    asm volatile("" : "+v"(tid));
    lds_barrier();
    FwdChain<B, 0>::template run<false>(tid, q3, lds, tw + 3 * N2, lc);
    B::fwd_canon(q3, lc);
    B::template store_bot_lds<NT>(tid, q3, out + p * N + 3 * N2, lds);
}

template <bool NT = false>
__global__ __launch_bounds__(256, kQuartersOcc) void ntt_inv_quarters_kernel(u64* __restrict__ out, const u64* __restrict__ in, DevTables<FoldArith> tb) {
    typedef Quarters14 Q;
    typedef Q::B B;
    constexpr int E = Q::E, N = Q::N, N2 = Q::N2;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    int tid = threadIdx.x;
    const size_t p = blockIdx.x;
    const int limb = (int)(p % (size_t)tb.n_limbs);
    const LimbConst lc = tb.lc[limb];
    const TwFold* tw = tb.qinv + (size_t)limb * N;
    const InvLast<TwFold> last = tb.qtop_last[limb];
    const TwFold wi2 = tb.qtop_inv[2 * limb], wi3 = tb.qtop_inv[2 * limb + 1];
    u64 q0[E], q1[E], q2[E], q3[E];
    B::template load_bot_lds<NT>(tid, q0, in + p * N, lds);            // the staged rows are the wave's own
    InvChain<B, B::NPH - 1, kUnit>::run(tid, q0, lds, tw, last, lc);
    asm volatile("" : "+v"(tid));
    lds_barrier();              // the chain's last exchange is read across waves; the rows below are written inside each wave's region
    B::template load_bot_lds<NT>(tid, q1, in + p * N + N2, lds);
    InvChain<B, B::NPH - 1, kUnit>::run(tid, q1, lds, tw + N2, last, lc);
    asm volatile("" : "+v"(tid));
    lds_barrier();
    B::template load_bot_lds<NT>(tid, q2, in + p * N + 2 * N2, lds);
    InvChain<B, B::NPH - 1, kUnit>::run(tid, q2, lds, tw + 2 * N2, last, lc);
    asm volatile("" : "+v"(tid));
    lds_barrier();
    B::template load_bot_lds<NT>(tid, q3, in + p * N + 3 * N2, lds);
    InvChain<B, B::NPH - 1, kUnit>::run(tid, q3, lds, tw + 3 * N2, last, lc);
    Q::inv_columns(q0, q1, q2, q3, wi2, wi3, last, lc);
    B::inv_canon(q0, lc);
    B::template store_top<NT>(tid, q0, out + p * N);
    B::inv_canon(q1, lc);
    B::template store_top<NT>(tid, q1, out + p * N + N2);
    B::inv_canon(q2, lc);
    B::template store_top<NT>(tid, q2, out + p * N + 2 * N2);
    B::inv_canon(q3, lc);
    B::template store_top<NT>(tid, q3, out + p * N + 3 * N2);
}

}  // namespace dpfhe
