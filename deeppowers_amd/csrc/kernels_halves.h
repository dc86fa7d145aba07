// kernels_halves.h - the N = 8192 kernels on the N = 4096 body (device code; included by launch_impl.h).
//
// SURVEY.md section 8(a) A1/A2 (batched transforms: in the library for batches of >= kHalvesMinPolys residue polynomials, launch.h) at BASELINE configs[4]'s ring degree; no reference counterpart (section 0).  ntt_halves.h has the arithmetic: one radix-2 column stage in
// registers, then the two independent 4096-point sub-transforms one after the other through ONE 38 KiB LDS buffer, 256 threads
// (4 waves) per workgroup - so that three to four workgroups share a CU and de-phase, where Geo<13, 4>'s 512-thread workgroups sit
// two (batched) or one (fused) to a CU behind 8-wave barriers.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "ntt_halves.h"

namespace dpfhe {

constexpr int kHalvesOcc = 3;     // workgroups (= waves per SIMD) the register budget is sized for (4 spills and is slower: profiles/r05_halves_ntt_ab.txt)

template <class Arith, bool NT = false>
__global__ __launch_bounds__(256, kHalvesOcc) void ntt_fwd_halves_kernel(u64* __restrict__ out, const u64* __restrict__ in, DevTables<Arith> tb) {
    typedef Halves13<Arith> H;
    typedef typename H::B B;
    constexpr int E = H::E, N = H::N, N2 = H::N2;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    int tid = threadIdx.x;
    const size_t p = blockIdx.x;
    const int limb = (int)(p % (size_t)tb.n_limbs);
    const LimbConst lc = tb.lc[limb];
    const typename B::Tw* tw = tb.hfwd + (size_t)limb * N;     // [limb][half][N2]
    const typename B::Tw wtop = tb.htop_fwd[limb];
    u64 lo[E], hi[E];
    B::template load_top<NT>(tid, lo, in + p * N);
    B::template load_top<NT>(tid, hi, in + p * N + N2);
    H::fwd_column(lo, hi, wtop, lc);
    FwdChain<B, 0>::template run<kHalvesOcc <= 3>(tid, lo, lds, tw, lc);       // (phase 1 twiddles requested with the data where the register budget allows)
    B::fwd_canon(lo, lc);
    B::template store_bot_lds<NT>(tid, lo, out + p * N, lds);
    asm volatile("" : "+v"(tid));   // the second chain fetches its own twiddles where it uses them (kernels.h ct_mul_kernel)
    lds_barrier();                  // the rows above are read by their own wave only; the next chain's first exchange writes every region
    FwdChain<B, 0>::template run<kHalvesOcc <= 3>(tid, hi, lds, tw + N2, lc);
    B::fwd_canon(hi, lc);
    B::template store_bot_lds<NT>(tid, hi, out + p * N + N2, lds);
}

template <class Arith, bool NT = false>
__global__ __launch_bounds__(256, kHalvesOcc) void ntt_inv_halves_kernel(u64* __restrict__ out, const u64* __restrict__ in, DevTables<Arith> tb) {
    typedef Halves13<Arith> H;
    typedef typename H::B B;
    constexpr int E = H::E, N = H::N, N2 = H::N2;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    int tid = threadIdx.x;
    const size_t p = blockIdx.x;
    const int limb = (int)(p % (size_t)tb.n_limbs);
    const LimbConst lc = tb.lc[limb];
    const typename B::Tw* tw = tb.hinv + (size_t)limb * N;
    const InvLast<typename B::Tw> last = tb.htop_last[limb];
    u64 lo[E], hi[E];
    {
        typename B::TwRegs tw_first;
        B::template load_tw<B::NPH - 1, false>(tid, tw, tw_first);
        u64 v[E];
        B::stage_load(tid, v, in + p * N + N2);               // the second half's words: in flight during the first half's transform
        B::template load_bot_lds<NT>(tid, lo, in + p * N, lds);
        InvChain<B, B::NPH - 1, kUnit>::run_with(tid, lo, lds, tw, last, lc, tw_first);
        asm volatile("" : "+v"(tid));
        lds_barrier();              // the chain's last exchange is read across waves; the rows below are written inside each wave's region
        B::stage_rows(tid, hi, v, lds);
    }
    InvChain<B, B::NPH - 1, kUnit>::run(tid, hi, lds, tw + N2, last, lc);
    H::inv_column(lo, hi, last, lc);
    B::inv_canon(lo, lc);
    B::template store_top<NT>(tid, lo, out + p * N);
    B::inv_canon(hi, lc);
    B::template store_top<NT>(tid, hi, out + p * N + N2);
}

// (round 5's relin_half_kernel - the giant-step key inner products as one 256-thread workgroup per (item, limb, half of the NTT domain) - measured 3 % slower than
//  relin_kernel<..., 13, 4, MODE 4> (profiles/r05_halves_relin_ab.txt) and was removed in round 6.)

}  // namespace dpfhe
