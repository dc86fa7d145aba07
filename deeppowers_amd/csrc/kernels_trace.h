// kernels_trace.h - DIAGNOSTIC builds only (-DDPFHE_DIAGNOSTICS; tools/ab_variant.sh diag -DDPFHE_DIAGNOSTICS; read back by tools/ntt_trace.py): the batched forward transform
// with per-workgroup s_memrealtime stamps (100 MHz), for the workgroup timelines of DESIGN.md section 5.  Thread 0 of every workgroup writes 8 words:
//   0 start, 1 first operand word in registers, 2 all operand words arrived, 3 transform + canonicalisation done, 4 stores issued, 5 stores drained,
//   6 HW_ID | XCC_ID << 32, 7 unused.   Same words out as the untraced kernels; not part of the library.
#pragma once
#include "kernels.h"
#include "kernels_halves.h"

namespace dpfhe {

inline u64* g_ntt_trace = nullptr;
inline unsigned g_ntt_trace_blocks = 0;

__device__ __forceinline__ void ntt_trace_write(u64* trace, u64 t0, u64 t1, u64 t2, u64 t3, u64 t4) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u64 t5 = trace_stamp<true>((u64)threadIdx.x);
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        u64* t = trace + (size_t)blockIdx.x * 8;
        t[0] = t0; t[1] = t1; t[2] = t2; t[3] = t3; t[4] = t4; t[5] = t5; t[6] = (u64)hw | ((u64)xcc << 32); t[7] = 0;
    }
}

template <class Arith, int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void ntt_fwd_trace_kernel(u64* __restrict__ out, const u64* __restrict__ in, DevTables<Arith> tb, u64* __restrict__ trace) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    const int tid = threadIdx.x;
    const u64 t0 = trace_stamp<true>((u64)tid);
    const size_t p = blockIdx.x;
    const int limb = (int)(p % (size_t)tb.n_limbs);
    const LimbConst lc = tb.lc[limb];
    const typename B::Tw* tw = tb.fwd + (size_t)limb * B::G::N;
    u64 x[B::E];
    B::load_top(tid, x, in + p * B::G::N);
    const u64 t1 = trace_stamp<true>(x[0]);
    u64 dep = 0;
#pragma unroll
    for (int k = 0; k < B::E; ++k) dep |= x[k];
    const u64 t2 = trace_stamp<true>(dep);
    FwdChain<B, 0>::run(tid, x, lds, tw, lc);
    B::fwd_canon(x, lc);
    dep = 0;
#pragma unroll
    for (int k = 0; k < B::E; ++k) dep |= x[k];
    const u64 t3 = trace_stamp<true>(dep);
    if constexpr (B::kLdsIO) B::store_bot_lds(tid, x, out + p * B::G::N, lds);
    else B::store_bot(tid, x, out + p * B::G::N);
    const u64 t4 = trace_stamp<true>((u64)tid);
    ntt_trace_write(trace, t0, t1, t2, t3, t4);
}

template <class Arith>
__global__ __launch_bounds__(256, kHalvesOcc) void ntt_fwd_halves_trace_kernel(u64* __restrict__ out, const u64* __restrict__ in, DevTables<Arith> tb, u64* __restrict__ trace) {
    typedef Halves13<Arith> H;
    typedef typename H::B B;
    constexpr int E = H::E, N = H::N, N2 = H::N2;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    int tid = threadIdx.x;
    const u64 t0 = trace_stamp<true>((u64)tid);
    const size_t p = blockIdx.x;
    const int limb = (int)(p % (size_t)tb.n_limbs);
    const LimbConst lc = tb.lc[limb];
    const typename B::Tw* tw = tb.hfwd + (size_t)limb * N;
    const typename B::Tw wtop = tb.htop_fwd[limb];
    u64 lo[E], hi[E];
    B::load_top(tid, lo, in + p * N);
    B::load_top(tid, hi, in + p * N + N2);
    const u64 t1 = trace_stamp<true>(lo[0]);
    u64 dep = 0;
#pragma unroll
    for (int k = 0; k < E; ++k) dep |= lo[k] | hi[k];
    const u64 t2 = trace_stamp<true>(dep);
    H::fwd_column(lo, hi, wtop, lc);
    FwdChain<B, 0>::template run<kHalvesOcc <= 3>(tid, lo, lds, tw, lc);
    B::fwd_canon(lo, lc);
    B::store_bot_lds(tid, lo, out + p * N, lds);
    asm volatile("" : "+v"(tid));
    lds_barrier();
    FwdChain<B, 0>::template run<kHalvesOcc <= 3>(tid, hi, lds, tw + N2, lc);
    B::fwd_canon(hi, lc);
    dep = 0;
#pragma unroll
    for (int k = 0; k < E; ++k) dep |= hi[k];
    const u64 t3 = trace_stamp<true>(dep);
    B::store_bot_lds(tid, hi, out + p * N + N2, lds);
    const u64 t4 = trace_stamp<true>((u64)tid);
    ntt_trace_write(trace, t0, t1, t2, t3, t4);
}

}  // namespace dpfhe
