// kernels_halves.h - the N = 8192 kernels on the N = 4096 body (device code; included by launch_impl.h).
//
// SURVEY.md section 8(a) A1/A2 (batched transforms: in the library for batches of >= kHalvesMinPolys residue polynomials, launch.h) and N1/N3 (the key-switch
// inner products of the packed layers: an A/B build) at BASELINE configs[4]'s ring degree; no reference counterpart (section 0).  ntt_halves.h has the arithmetic: one radix-2 column stage in
// registers, then the two independent 4096-point sub-transforms one after the other through ONE 38 KiB LDS buffer, 256 threads
// (4 waves) per workgroup - so that three to four workgroups share a CU and de-phase, where Geo<13, 4>'s 512-thread workgroups sit
// two (batched) or one (fused) to a CU behind 8-wave barriers.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "ntt_halves.h"

namespace dpfhe {

#ifndef DPFHE_HALVES_OCC
#define DPFHE_HALVES_OCC 3     // workgroups (= waves per SIMD) the register budget is sized for
#endif
#ifndef DPFHE_RELIN_HALF_OCC
#define DPFHE_RELIN_HALF_OCC 2  // relin_half_kernel: 192 VGPRs as written; 3 would need <= 168
#endif
#ifndef DPFHE_HALVES_INV_EARLY
#define DPFHE_HALVES_INV_EARLY 1   // request the second half's words before the first half's transform
#endif

template <class Arith, bool NT = false>
__global__ __launch_bounds__(256, DPFHE_HALVES_OCC) void ntt_fwd_halves_kernel(u64* __restrict__ out, const u64* __restrict__ in, DevTables<Arith> tb) {
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
    FwdChain<B, 0>::template run<DPFHE_HALVES_OCC <= 3>(tid, lo, lds, tw, lc);       // (phase 1 twiddles requested with the data where the register budget allows)
    B::fwd_canon(lo, lc);
    B::template store_bot_lds<NT>(tid, lo, out + p * N, lds);
    asm volatile("" : "+v"(tid));   // the second chain fetches its own twiddles where it uses them (kernels.h ct_mul_kernel)
    lds_barrier();                  // the rows above are read by their own wave only; the next chain's first exchange writes every region
    FwdChain<B, 0>::template run<DPFHE_HALVES_OCC <= 3>(tid, hi, lds, tw + N2, lc);
    B::fwd_canon(hi, lc);
    B::template store_bot_lds<NT>(tid, hi, out + p * N + N2, lds);
}

template <class Arith, bool NT = false>
__global__ __launch_bounds__(256, DPFHE_HALVES_OCC) void ntt_inv_halves_kernel(u64* __restrict__ out, const u64* __restrict__ in, DevTables<Arith> tb) {
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
#if DPFHE_HALVES_INV_EARLY
        u64 v[E];
        B::stage_load(tid, v, in + p * N + N2);               // in flight during the first half's transform
#endif
        B::template load_bot_lds<NT>(tid, lo, in + p * N, lds);
        InvChain<B, B::NPH - 1, kUnit>::run_with(tid, lo, lds, tw, last, lc, tw_first);
        asm volatile("" : "+v"(tid));
        lds_barrier();              // the chain's last exchange is read across waves; the rows below are written inside each wave's region
#if DPFHE_HALVES_INV_EARLY
        B::stage_rows(tid, hi, v, lds);
#else
        B::template load_bot_lds<NT>(tid, hi, in + p * N + N2, lds);
#endif
    }
    InvChain<B, B::NPH - 1, kUnit>::run(tid, hi, lds, tw + N2, last, lc);
    H::inv_column(lo, hi, last, lc);
    B::inv_canon(lo, lc);
    B::template store_top<NT>(tid, lo, out + p * N);
    B::inv_canon(hi, lc);
    B::template store_top<NT>(tid, hi, out + p * N + N2);
}

#if DPFHE_RELIN13_HALVES   // A/B builds only: measured 3 % slower than relin_kernel<..., 13, 4, MODE 4> (profiles/r05_halves_relin_ab.txt)
// ------------------------------------------------------------------------------------------------
// N3: the giant-step key inner products of a packed layer (relin_kernel MODE 4: digits of c1 -> forward transforms -> multiply-accumulate
// with the two key polynomials, result LEFT in the NTT domain over Q P) with one workgroup per (item, limb, HALF of the NTT domain).
// A half of the NTT domain is one 4096-point sub-transform of the column stage's outputs, and nothing downstream couples the halves
// (the products are coefficient-wise, the sums stay in the NTT domain), so the two halves of an (item, limb) are independent workgroups:
// 256 threads, 38 KiB of LDS, two to a CU, where relin_kernel<..., 13, 4, 4> is one 512-thread workgroup per CU with 80 KiB.  Price: each
// workgroup runs the whole column stage (16 butterflies per thread) for its half - 1/13 more butterflies - and reads both halves of
// every digit (L2: the digits are shared by the L limbs' workgroups already).
// Workgroup ids keep relin_kernel's XCD-aware layouts with 2 L "virtual limbs" (limb, half).
// ------------------------------------------------------------------------------------------------
template <class Arith>
__global__ __launch_bounds__(256, DPFHE_RELIN_HALF_OCC) void relin_half_kernel(u64* __restrict__ out2, const u64* __restrict__ in2, const u64* __restrict__ evk, size_t key_stride,
                                                             unsigned key_group, unsigned n_outer, DevTables<Arith> tb) {
    typedef Halves13<Arith> H;
    typedef typename H::B B;
    constexpr int E = H::E, N = H::N, N2 = H::N2;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    int tid = threadIdx.x;
    const int L = tb.n_limbs, VL = 2 * L, Ld = L - 1;
    size_t bi;
    int vl;
    if (n_outer & kRelinRotMajor) {          // all workgroups of a key on one XCD, virtual-limb-major (kernels.h relin_kernel)
        const unsigned n_keys = (n_outer & ~kRelinRotMajor) / (unsigned)VL, per_key = (unsigned)VL * key_group;
        const unsigned q = blockIdx.x >> 3, w = q % per_key, key = (q / per_key) * 8u + (blockIdx.x & 7u);
        if (key >= n_keys) return;
        vl = (int)(w / key_group);
        bi = (size_t)key * key_group + w % key_group;
    } else if (n_outer) {                    // the items of a key group for one virtual limb: same XCD, adjacent in time
        const unsigned q = blockIdx.x >> 3, inner = q % key_group, outer = (q / key_group) * 8u + (blockIdx.x & 7u);
        if (outer >= n_outer) return;
        vl = (int)(outer % (unsigned)VL);
        bi = (size_t)(outer / (unsigned)VL) * key_group + inner;
    } else {
        bi = blockIdx.x / (unsigned)VL;
        vl = (int)(blockIdx.x % (unsigned)VL);
    }
    const int limb = vl >> 1, half = vl & 1;
    const LimbConst lc = tb.lc[limb];
    const typename B::Tw* const tw = tb.hfwd + (size_t)limb * N + (size_t)half * N2;
    const typename B::Tw wtop = tb.htop_fwd[limb];
    const u64* c1 = in2 + ((bi * 2 + 1) * Ld) * N;            // digit j at + j N
    evk += (bi / key_group) * key_stride;
    const u64 two_q = 2 * lc.q;
    u64 acc0[E], acc1[E];
#pragma unroll
    for (int k = 0; k < E; ++k) acc0[k] = acc1[k] = 0;
    int lazy_terms = 0;
#pragma unroll 1
    for (int j = 0; j < Ld; ++j) {
        asm volatile("" : "+v"(tid));
        u64 x[E];
        {
            u64 lo[E], hi[E];
            B::load_top(tid, lo, c1 + (size_t)j * N);
            B::load_top(tid, hi, c1 + (size_t)j * N + N2);
            // [c1]_{q_j} mod q_i.  FoldArith: every residue is < 2^60 and the column stage's fused multiply-add takes any addend < 2^60 and
            // hands over words below 4 q + 2 d (inside NttBody's SUB input bound), so the digit needs no reduction of its own; generic primes
            // may be much smaller than the digit's modulus: canonical first
            if constexpr (!Arith::kFold) {
#pragma unroll
                for (int k = 0; k < E; ++k) { lo[k] = canon_any<Arith>(lo[k], lc); hi[k] = canon_any<Arith>(hi[k], lc); }
            }
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const u64 a = lo[k];
                if constexpr (Arith::kFold) {
                    const u64 s = FoldArith::mul_tw_add(hi[k], wtop, lc, a);
                    x[k] = half ? chk_shl1_add_sub(a, two_q, s) : s;
                } else {
                    const u64 t = Arith::mul_tw(hi[k], wtop, lc);
                    x[k] = half ? a - t + two_q : a + t;
                }
            }
        }
        if (j > 0) lds_barrier();
        FwdChain<B, 0>::template run<false>(tid, x, lds, tw, lc);
        if (Arith::kFold) {
            static_assert(13 * kMulB + kRedB <= kWord, "13 lazily added products + one reduced word must fit a 64-bit word");
            if (lazy_terms == 13) {
#pragma unroll
                for (int k = 0; k < E; ++k) { acc0[k] = FoldArith::reduce(acc0[k], lc); acc1[k] = FoldArith::reduce(acc1[k], lc); }
                lazy_terms = 1;
            }
            ++lazy_terms;
        } else {
            B::fwd_canon(x, lc);
        }
        const u64* k0 = evk + (((size_t)j * 2 + 0) * L + limb) * N + (size_t)half * N2;   // this half of the key polynomials (NTT domain, window-0 mapping)
        const u64* k1 = evk + (((size_t)j * 2 + 1) * L + limb) * N + (size_t)half * N2;
        {
            u64 e[E];
            if constexpr (B::kLdsIO) B::load_bot_lds(tid, e, k0, lds); else B::load_bot(tid, e, k0);
#pragma unroll
            for (int k = 0; k < E; ++k)
                acc0[k] = Arith::kFold ? acc0[k] + FoldArith::mul60(x[k], e[k], (u32)lc.d) : add_mod(acc0[k], ShoupArith::mul_var(x[k], e[k], lc), lc.q);
            asm volatile("" ::: "memory");
            if constexpr (B::kLdsIO) B::load_bot_lds(tid, e, k1, lds); else B::load_bot(tid, e, k1);
#pragma unroll
            for (int k = 0; k < E; ++k)
                acc1[k] = Arith::kFold ? acc1[k] + FoldArith::mul60(x[k], e[k], (u32)lc.d) : add_mod(acc1[k], ShoupArith::mul_var(x[k], e[k], lc), lc.q);
        }
    }
    if (Arith::kFold) {
#pragma unroll
        for (int k = 0; k < E; ++k) { acc0[k] = FoldArith::canon(acc0[k], lc); acc1[k] = FoldArith::canon(acc1[k], lc); }
    }
    u64* o0 = out2 + ((bi * 2 + 0) * L + limb) * N + (size_t)half * N2;
    u64* o1 = out2 + ((bi * 2 + 1) * L + limb) * N + (size_t)half * N2;
    if constexpr (B::kLdsIO && Arith::kFold) {   // through the wave's own LDS rows, non-temporal (kernels.h relin_kernel MODE 4)
        B::template store_bot_lds<true>(tid, acc0, o0, lds);
        wave_sync();
        B::template store_bot_lds<true>(tid, acc1, o1, lds);
    } else {
        B::store_bot(tid, acc0, o0);
        B::store_bot(tid, acc1, o1);
    }
}

#endif

}  // namespace dpfhe
