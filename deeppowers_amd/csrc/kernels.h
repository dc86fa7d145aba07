// kernels.h - the gfx950 kernels of the FHE hot path (device code; included by dpfhe_cabi.hip only).
//
// A1/A2 batched NTT, A3 dyadic ops, A6 fused ct x ct multiply, A7 ct x pt
// matvec, A8 shard-local reduce.  No reference counterpart (section 0).  Integer modular arithmetic:
// no MFMA; the bounds are 64-bit-multiply issue rate (NTT, ct_mul) and HBM bandwidth (dyadic, matvec).
#pragma once
#include <hip/hip_runtime.h>

#include "devtables.h"
#include "launch.h"
#include "ntt_core.h"
#include "ntt_top.h"

namespace dpfhe {

// Workgroup barrier of the ALL-TO-ALL LDS exchange (top window <-> next window), the only one that crosses waves.
// (A barrier that waits for LDS traffic only - s_waitcnt lgkmcnt(0); s_barrier, leaving the prefetched twiddle loads in
// flight - was measured in round 1: equal on ct_mul and the forward NTT, 7 % slower on the inverse NTT, so the plain
// barrier stays.)
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
// Wave-local exchanges (ntt_core.h Geo::exch_wave_local) live in the wave's private LDS region: the LDS executes one
// wave's DS instructions in order, so no hardware synchronisation is needed - only the compiler must not move the
// wave's LDS reads above its LDS writes (it cannot see that OTHER lanes' stores feed this lane's loads).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// Barrier contract of the chains below.  FwdChain: its first exchange writes into EVERY wave's region, so a caller that
// used the LDS buffer before (a previous transform, staged tiles) places one lds_barrier() in front of the chain; the
// chain itself has one barrier (after the all-to-all write).  InvChain: everything before its last exchange stays in
// the wave's own region, the last exchange writes the own region and reads every region after the chain's single
// barrier; a caller running two inverse chains back to back places one lds_barrier() between them (the second chain's
// wave-local writes would otherwise overwrite words other waves are still reading).
// (Geometries with 8 words per thread at N >= 8192 have a second exchange that crosses waves; it is bracketed by barriers.)
template <class G, int X, bool FWD>
__device__ __forceinline__ void exch_sync_before_write() {
    if constexpr (G::exch_wave_local(X)) {
        wave_sync();
    } else {
        // forward: exchange 0 opens the chain (the caller vouches for the buffer); inverse: the first cross-wave exchange met
        // (the highest index) is WRITTEN inside the wave's own region; any further cross-wave exchange must wait for the readers
        constexpr bool first = FWD ? X == 0 : G::highest_cross_wave_exchange() == X;
        if constexpr (!first) lds_barrier();
    }
}
template <class G, int X>
__device__ __forceinline__ void exch_sync_after_write() {
    if constexpr (G::exch_wave_local(X)) wave_sync(); else lds_barrier();
}

// ------------------------------------------------------------------------------------------------
// register-resident transforms shared by the NTT kernels and the fused ct x ct kernel
// ------------------------------------------------------------------------------------------------
// The per-thread twiddles of phase P+1 are requested BEFORE the LDS exchange that follows phase P (the registers of
// phase P's twiddles are dead by then), so their L2 latency overlaps the exchange instead of stalling the next phase.
template <class B, int P>
struct FwdChain {
    typedef typename B::TwRegs TwRegs;
    // entry point (P = 0): the top window's twiddles are workgroup-uniform scalars, so the per-thread twiddles of phase 1
    // are requested right away, together with the caller's data loads, a whole phase ahead of their first use
    // (EARLY = false where the extra 4 (E-1) live registers during phase 0 would spill: the fused kernels)
    template <bool EARLY = true>
    static __device__ __forceinline__ void run(int tid, u64 (&x)[B::E], u64* lds, const typename B::Tw* tw, const LimbConst& lc) {
        TwRegs twr;
        B::template load_tw<P, true>(tid, tw, twr);
        if constexpr (!EARLY) {
            run_with(tid, x, lds, tw, lc, twr);
        } else if constexpr (P + 1 < B::NPH) {
            TwRegs nxt;
            B::template load_tw<P + 1, true>(tid, tw, nxt);
            B::template fwd_phase_r<P>(x, twr, lc);
            exch_sync_before_write<typename B::G, P, true>();
            B::template lds_write<P, P, true>(tid, x, lds);
            exch_sync_after_write<typename B::G, P>();
            B::template lds_read<P, P + 1, true>(tid, x, lds);
            FwdChain<B, P + 1>::run_with(tid, x, lds, tw, lc, nxt);
        } else {
            B::template fwd_phase_r<P>(x, twr, lc);
        }
    }
    static __device__ __forceinline__ void run_with(int tid, u64 (&x)[B::E], u64* lds, const typename B::Tw* tw, const LimbConst& lc,
                                                    const TwRegs& twr) {
        B::template fwd_phase_r<P>(x, twr, lc);
        if constexpr (P + 1 < B::NPH) {
            TwRegs nxt;
            B::template load_tw<P + 1, true>(tid, tw, nxt);
            exch_sync_before_write<typename B::G, P, true>();
            B::template lds_write<P, P, true>(tid, x, lds);
            exch_sync_after_write<typename B::G, P>();
            B::template lds_read<P, P + 1, true>(tid, x, lds);
            FwdChain<B, P + 1>::run_with(tid, x, lds, tw, lc, nxt);
        }
    }
};

// Two transforms OF THE SAME LIMB side by side (the fused multiply's a0|b0 and a1|b1): one set of twiddle fetches and
// twiddle addresses for both, two independent dependency chains per wave (the second polynomial's butterflies issue while
// the first one's LDS exchange is in flight), one barrier for both all-to-all exchanges.  Two LDS buffers.
template <class B, int P>
struct FwdChain2 {
    typedef typename B::TwRegs TwRegs;
    static __device__ __forceinline__ void run(int tid, u64 (&x)[B::E], u64 (&y)[B::E], u64* lds0, u64* lds1, const typename B::Tw* tw,
                                               const LimbConst& lc) {
        TwRegs twr;
        B::template load_tw<P, true>(tid, tw, twr);
        run_with(tid, x, y, lds0, lds1, tw, lc, twr);
    }
    static __device__ __forceinline__ void run_with(int tid, u64 (&x)[B::E], u64 (&y)[B::E], u64* lds0, u64* lds1, const typename B::Tw* tw,
                                                    const LimbConst& lc, const TwRegs& twr) {
        B::template fwd_phase_r<P>(x, twr, lc);
        if constexpr (P + 1 < B::NPH) {
            exch_sync_before_write<typename B::G, P, true>();
            B::template lds_write<P, P, true>(tid, x, lds0);
            B::template fwd_phase_r<P>(y, twr, lc);
            TwRegs nxt;
            B::template load_tw<P + 1, true>(tid, tw, nxt);
            B::template lds_write<P, P, true>(tid, y, lds1);
            exch_sync_after_write<typename B::G, P>();
            B::template lds_read<P, P + 1, true>(tid, x, lds0);
            B::template lds_read<P, P + 1, true>(tid, y, lds1);
            FwdChain2<B, P + 1>::run_with(tid, x, y, lds0, lds1, tw, lc, nxt);
        } else {
            B::template fwd_phase_r<P>(y, twr, lc);
        }
    }
};

// FOUR transforms of one limb on TWO LDS buffers: per phase the pair (x, y) goes through the exchange first, then (z, w) reuses
// the buffers - all four share one set of twiddle fetches.  The buffers are re-written only after every reader of the previous
// pair is done: in program order for wave-local exchanges, behind a workgroup barrier for the all-to-all one.
template <class G, int X>
__device__ __forceinline__ void exch_sync_reuse() {
    if constexpr (G::exch_wave_local(X)) wave_sync(); else lds_barrier();
}
template <class B, int P>
struct FwdChain4 {
    typedef typename B::TwRegs TwRegs;
    static __device__ __forceinline__ void run(int tid, u64 (&x)[B::E], u64 (&y)[B::E], u64 (&z)[B::E], u64 (&w)[B::E], u64* lds0, u64* lds1,
                                               const typename B::Tw* tw, const LimbConst& lc) {
        TwRegs twr;
        B::template load_tw<P, true>(tid, tw, twr);
        run_with(tid, x, y, z, w, lds0, lds1, tw, lc, twr);
    }
    static __device__ __forceinline__ void run_with(int tid, u64 (&x)[B::E], u64 (&y)[B::E], u64 (&z)[B::E], u64 (&w)[B::E], u64* lds0, u64* lds1,
                                                    const typename B::Tw* tw, const LimbConst& lc, const TwRegs& twr) {
        B::template fwd_phase_r<P>(x, twr, lc);
        if constexpr (P + 1 < B::NPH) {
            exch_sync_before_write<typename B::G, P, true>();
            B::template lds_write<P, P, true>(tid, x, lds0);
            B::template fwd_phase_r<P>(y, twr, lc);
            B::template lds_write<P, P, true>(tid, y, lds1);
            exch_sync_after_write<typename B::G, P>();
            B::template lds_read<P, P + 1, true>(tid, x, lds0);
            B::template lds_read<P, P + 1, true>(tid, y, lds1);
            B::template fwd_phase_r<P>(z, twr, lc);
            exch_sync_reuse<typename B::G, P>();
            B::template lds_write<P, P, true>(tid, z, lds0);
            B::template fwd_phase_r<P>(w, twr, lc);
            TwRegs nxt;
            B::template load_tw<P + 1, true>(tid, tw, nxt);
            B::template lds_write<P, P, true>(tid, w, lds1);
            exch_sync_after_write<typename B::G, P>();
            B::template lds_read<P, P + 1, true>(tid, z, lds0);
            B::template lds_read<P, P + 1, true>(tid, w, lds1);
            FwdChain4<B, P + 1>::run_with(tid, x, y, z, w, lds0, lds1, tw, lc, nxt);
        } else {
            B::template fwd_phase_r<P>(y, twr, lc);
            B::template fwd_phase_r<P>(z, twr, lc);
            B::template fwd_phase_r<P>(w, twr, lc);
        }
    }
};

template <class B, int P, int IN>
struct InvChain {
    typedef typename B::TwRegs TwRegs;
    static __device__ __forceinline__ void run(int tid, u64 (&x)[B::E], u64* lds, const typename B::Tw* tw,
                                               const InvLast<typename B::Tw>& last, const LimbConst& lc) {
        TwRegs twr;
        B::template load_tw<P, false>(tid, tw, twr);
        run_with(tid, x, lds, tw, last, lc, twr);
    }
    static __device__ __forceinline__ void run_with(int tid, u64 (&x)[B::E], u64* lds, const typename B::Tw* tw,
                                                    const InvLast<typename B::Tw>& last, const LimbConst& lc, const TwRegs& twr) {
        B::template inv_phase_r<P, IN>(x, twr, last.w_last, last.w_ninv, lc);
        if constexpr (P > 0) {
            TwRegs nxt;
            B::template load_tw<P - 1, false>(tid, tw, nxt);
            exch_sync_before_write<typename B::G, P - 1, false>();    // first cross-wave exchange: writes the wave's OWN region
            B::template lds_write<P - 1, P, false>(tid, x, lds);
            exch_sync_after_write<typename B::G, P - 1>();
            B::template lds_read<P - 1, P - 1, false>(tid, x, lds);
            InvChain<B, P - 1, IN>::run_with(tid, x, lds, tw, last, lc, nxt);
        }
    }
};

template <class B, int P, int IN>
struct InvChain2 {
    typedef typename B::TwRegs TwRegs;
    static __device__ __forceinline__ void run(int tid, u64 (&x)[B::E], u64 (&y)[B::E], u64* lds0, u64* lds1, const typename B::Tw* tw,
                                               const InvLast<typename B::Tw>& last, const LimbConst& lc) {
        TwRegs twr;
        B::template load_tw<P, false>(tid, tw, twr);
        run_with(tid, x, y, lds0, lds1, tw, last, lc, twr);
    }
    static __device__ __forceinline__ void run_with(int tid, u64 (&x)[B::E], u64 (&y)[B::E], u64* lds0, u64* lds1, const typename B::Tw* tw,
                                                    const InvLast<typename B::Tw>& last, const LimbConst& lc, const TwRegs& twr) {
        B::template inv_phase_r<P, IN>(x, twr, last.w_last, last.w_ninv, lc);
        if constexpr (P > 0) {
            exch_sync_before_write<typename B::G, P - 1, false>();
            B::template lds_write<P - 1, P, false>(tid, x, lds0);
            B::template inv_phase_r<P, IN>(y, twr, last.w_last, last.w_ninv, lc);
            TwRegs nxt;
            B::template load_tw<P - 1, false>(tid, tw, nxt);
            B::template lds_write<P - 1, P, false>(tid, y, lds1);
            exch_sync_after_write<typename B::G, P - 1>();
            B::template lds_read<P - 1, P - 1, false>(tid, x, lds0);
            B::template lds_read<P - 1, P - 1, false>(tid, y, lds1);
            InvChain2<B, P - 1, IN>::run_with(tid, x, y, lds0, lds1, tw, last, lc, nxt);
        } else {
            B::template inv_phase_r<P, IN>(y, twr, last.w_last, last.w_ninv, lc);
        }
    }
};

// Three inverse transforms of one limb on two LDS buffers (x | y, then z reuses buffer 0), one set of twiddle fetches.
template <class B, int P, int IN>
struct InvChain3 {
    typedef typename B::TwRegs TwRegs;
    static __device__ __forceinline__ void run(int tid, u64 (&x)[B::E], u64 (&y)[B::E], u64 (&z)[B::E], u64* lds0, u64* lds1, const typename B::Tw* tw,
                                               const InvLast<typename B::Tw>& last, const LimbConst& lc) {
        TwRegs twr;
        B::template load_tw<P, false>(tid, tw, twr);
        run_with(tid, x, y, z, lds0, lds1, tw, last, lc, twr);
    }
    static __device__ __forceinline__ void run_with(int tid, u64 (&x)[B::E], u64 (&y)[B::E], u64 (&z)[B::E], u64* lds0, u64* lds1,
                                                    const typename B::Tw* tw, const InvLast<typename B::Tw>& last, const LimbConst& lc, const TwRegs& twr) {
        B::template inv_phase_r<P, IN>(x, twr, last.w_last, last.w_ninv, lc);
        if constexpr (P > 0) {
            exch_sync_before_write<typename B::G, P - 1, false>();
            B::template lds_write<P - 1, P, false>(tid, x, lds0);
            B::template inv_phase_r<P, IN>(y, twr, last.w_last, last.w_ninv, lc);
            B::template lds_write<P - 1, P, false>(tid, y, lds1);
            exch_sync_after_write<typename B::G, P - 1>();
            B::template lds_read<P - 1, P - 1, false>(tid, x, lds0);
            B::template lds_read<P - 1, P - 1, false>(tid, y, lds1);
            B::template inv_phase_r<P, IN>(z, twr, last.w_last, last.w_ninv, lc);
            TwRegs nxt;
            B::template load_tw<P - 1, false>(tid, tw, nxt);
            exch_sync_reuse<typename B::G, P - 1>();
            B::template lds_write<P - 1, P, false>(tid, z, lds0);
            exch_sync_after_write<typename B::G, P - 1>();
            B::template lds_read<P - 1, P - 1, false>(tid, z, lds0);
            InvChain3<B, P - 1, IN>::run_with(tid, x, y, z, lds0, lds1, tw, last, lc, nxt);
        } else {
            B::template inv_phase_r<P, IN>(y, twr, last.w_last, last.w_ninv, lc);
            B::template inv_phase_r<P, IN>(z, twr, last.w_last, last.w_ninv, lc);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// A1 / A2: batched NTT, one workgroup per residue polynomial (4 resident per CU at N=4096; the hardware
// dispatcher keeps the generations de-phased - persistent workgroups with prefetch, up-front twiddle fetch and
// staggered starts were all measured slower, MEASUREMENTS.md section 5).
// ------------------------------------------------------------------------------------------------
// NT: non-temporal global accesses, chosen by the launcher for batches whose input + output exceed the 256 MiB Infinity Cache (a
// stream that cannot stay there gains from not allocating in it - the plain copy kernel goes from 5.7 to 6.25 TB/s -, while
// configs[1]'s 128 MiB, which DO stay, measured 6 % slower with them: MEASUREMENTS.md section 5).
template <class Arith, int LOGN, int LOGE, bool NT = false>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void ntt_fwd_kernel(u64* __restrict__ out, const u64* __restrict__ in,
                                                                      DevTables<Arith> tb) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    const int tid = threadIdx.x;
    size_t p = blockIdx.x;                   // block p transforms words [p N, (p + 1) N): one polynomial, or one of its n_sub blocks
    const size_t sub = p % (size_t)tb.n_sub;
    int limb = (int)((p / (size_t)tb.n_sub) % (size_t)tb.n_limbs);
    if (tb.n_active) {                       // a launch over one arithmetic class of the context's limbs (devtables.h; n_sub = 1)
        size_t item;
        block_item_limb(tb, blockIdx.x, item, limb);
        p = item * (size_t)tb.n_limbs + (size_t)limb;
    }
    const LimbConst lc = tb.lc[limb];
    const typename B::Tw* tw = tb.fwd + ((size_t)limb * tb.n_sub + sub) * B::G::N;
    u64 x[B::E];
    B::template load_top<NT>(tid, x, in + p * B::G::N);
    FwdChain<B, 0>::run(tid, x, lds, tw, lc);
    B::fwd_canon(x, lc);
    if constexpr (B::kLdsIO) B::template store_bot_lds<NT>(tid, x, out + p * B::G::N, lds);   // rows == what this wave read in the last exchange
    else B::store_bot(tid, x, out + p * B::G::N);
}

template <class Arith, int LOGN, int LOGE, bool NT = false>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void ntt_inv_kernel(u64* __restrict__ out, const u64* __restrict__ in,
                                                                      DevTables<Arith> tb) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    const int tid = threadIdx.x;
    size_t p = blockIdx.x;
    const size_t sub = p % (size_t)tb.n_sub;
    int limb = (int)((p / (size_t)tb.n_sub) % (size_t)tb.n_limbs);
    if (tb.n_active) {
        size_t item;
        block_item_limb(tb, blockIdx.x, item, limb);
        p = item * (size_t)tb.n_limbs + (size_t)limb;
    }
    const LimbConst lc = tb.lc[limb];
    const typename B::Tw* tw = tb.inv + ((size_t)limb * tb.n_sub + sub) * B::G::N;
    const InvLast<typename B::Tw> last = tb.last[(size_t)limb * tb.n_sub + sub];
    typename B::TwRegs tw_first;   // requested before the data: load_bot waits for all of its loads (register transposition)
    B::template load_tw<B::NPH - 1, false>(tid, tw, tw_first);
    u64 x[B::E];
    if constexpr (B::kLdsIO) {
        B::template load_bot_lds<NT>(tid, x, in + p * B::G::N, lds);
        InvChain<B, B::NPH - 1, kUnit>::run_with(tid, x, lds, tw, last, lc, tw_first);   // the staged rows are the wave's own
    } else {
        B::load_bot(tid, x, in + p * B::G::N);
        InvChain<B, B::NPH - 1, kUnit>::run_with(tid, x, lds, tw, last, lc, tw_first);
    }
    B::inv_canon(x, lc);
    B::template store_top<NT>(tid, x, out + p * B::G::N);
}

// ------------------------------------------------------------------------------------------------
// Round 6: the batched transforms of a context whose limbs run on DIFFERENT arithmetic classes (dpfhe_cabi.hip dpfhe_ctx::classes) in ONE launch.
// The tables of such a context live in one blob whose per-limb slots are each in their limb's own format (all twiddle types are 16 bytes, so the
// strides agree); the limb's class is workgroup-uniform, so the kernel branches once, on a scalar, into that class's transform - the same per-thread
// code as ntt_fwd_kernel / ntt_inv_kernel.  Four arms - Shoup, fold, f64, scaled fold; a fifth (F64WideArith) spills a few registers at N = 2048 / 4096, so contexts
// with such limbs AND another class launch per class (launch_ntt_classes returns 1).  (One launch per class measured 54.7 / 50.6 % of HBM peak on the 59/50/40/33-bit context at configs[1]'s
// batch, against 69 and 59 % for its two classes alone: two half-size launches each pay their own tail.)
// ------------------------------------------------------------------------------------------------
template <class Arith, int LOGN, int LOGE>
__device__ __forceinline__ void ntt_fwd_block(u64* __restrict__ out, const u64* __restrict__ in, size_t p, int limb, const MixedTables& tb, u64* lds) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    const int tid = threadIdx.x;
    const LimbConst lc = tb.lc[limb];
    const typename B::Tw* tw = reinterpret_cast<const typename B::Tw*>(tb.fwd) + (size_t)limb * B::G::N;
    u64 x[B::E];
    B::load_top(tid, x, in + p * B::G::N);
    FwdChain<B, 0>::run(tid, x, lds, tw, lc);
    B::fwd_canon(x, lc);
    if constexpr (B::kLdsIO) B::store_bot_lds(tid, x, out + p * B::G::N, lds);
    else B::store_bot(tid, x, out + p * B::G::N);
}
template <class Arith, int LOGN, int LOGE>
__device__ __forceinline__ void ntt_inv_block(u64* __restrict__ out, const u64* __restrict__ in, size_t p, int limb, const MixedTables& tb, u64* lds) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    const int tid = threadIdx.x;
    const LimbConst lc = tb.lc[limb];
    const typename B::Tw* tw = reinterpret_cast<const typename B::Tw*>(tb.inv) + (size_t)limb * B::G::N;
    const InvLast<typename B::Tw> last = reinterpret_cast<const InvLast<typename B::Tw>*>(tb.last)[limb];
    typename B::TwRegs tw_first;
    B::template load_tw<B::NPH - 1, false>(tid, tw, tw_first);
    u64 x[B::E];
    if constexpr (B::kLdsIO) B::load_bot_lds(tid, x, in + p * B::G::N, lds);
    else B::load_bot(tid, x, in + p * B::G::N);
    InvChain<B, B::NPH - 1, kUnit>::run_with(tid, x, lds, tw, last, lc, tw_first);
    B::inv_canon(x, lc);
    B::store_top(tid, x, out + p * B::G::N);
}
// (launch bounds: the occupancy the single-class kernels reach - 4 waves per SIMD - is requested explicitly; left alone the merged kernel takes a few
//  registers more than its widest arm and drops to 3)
template <int LOGN, int LOGE, bool FWD>
__global__ __launch_bounds__(1 << (LOGN - LOGE), (1 << (LOGN - LOGE)) >= 1024 ? 1 : (1 << (LOGN - LOGE)) >= 512 ? 2 : 4) void ntt_classes_kernel(u64* __restrict__ out, const u64* __restrict__ in, MixedTables tb) {
    __shared__ __attribute__((aligned(16))) u64 lds[Geo<LOGN, LOGE>::lds_words()];
    const size_t p = blockIdx.x;
    const int limb = (int)(p % (size_t)tb.n_limbs);
    const int cls = (int)((tb.cls_map >> (4 * limb)) & 15);
#define DPFHE_CLS_RUN(A)                                                         \
    do {                                                                         \
        if (FWD) ntt_fwd_block<A, LOGN, LOGE>(out, in, p, limb, tb, lds);        \
        else ntt_inv_block<A, LOGN, LOGE>(out, in, p, limb, tb, lds);            \
    } while (0)
    if (cls == 2) DPFHE_CLS_RUN(F64Arith);
    else if (cls == 3) DPFHE_CLS_RUN(FoldScaledArith);
    else if (cls == 1) DPFHE_CLS_RUN(FoldArith);
    else DPFHE_CLS_RUN(ShoupArith);
#undef DPFHE_CLS_RUN
}

// N3: inverse NTT of sigma_g applied in the NTT domain.  In forward-output order position p carries the evaluation at
// psi^(2 brv(p) + 1) and sigma_g: a(X) -> a(X^g) only permutes evaluation points, NTT(sigma_g a)[p] = NTT(a)[p'] with
// 2 brv(p') + 1 = g (2 brv(p) + 1) mod 2N: the kernel gathers its input words through that permutation (8-byte gathers inside one
// 2^LOGN-word polynomial: L2 hits after the first touch of a line) and runs the ordinary inverse transform, so
// out = sigma_g(INTT(in)) with no separate automorphism pass.  `polys_per_elt` consecutive residue polynomials share element
// elts.v[blockIdx.x / polys_per_elt].  In place is safe: every load of the workgroup precedes the transform's workgroup barrier,
// every store follows it.
struct GaloisElts { unsigned v[kMaxGaloisBatch]; };

template <class Arith, int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void ntt_inv_galois_kernel(u64* __restrict__ out, const u64* __restrict__ in, GaloisElts elts,
                                                                             unsigned polys_per_elt, DevTables<Arith> tb) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    constexpr int E = B::E, N = B::G::N;
    static_assert(B::G::highest_cross_wave_exchange() >= 0 || B::G::T <= 64, "in-place safety relies on the transform's workgroup barrier");
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    const int tid = threadIdx.x;
    // (a launch over one class of a mixed context: blockIdx.x counts (polynomial, class limb) pairs; polys_per_elt is then per class too)
    size_t p = blockIdx.x;
    int limb = (int)(p % (size_t)tb.n_limbs);
    if (tb.n_active) {
        size_t item;
        block_item_limb(tb, blockIdx.x, item, limb);
        p = item * (size_t)tb.n_limbs + (size_t)limb;
    }
    const unsigned g = elts.v[blockIdx.x / polys_per_elt];
    const LimbConst lc = tb.lc[limb];
    const typename B::Tw* tw = tb.inv + (size_t)limb * N;
    const InvLast<typename B::Tw> last = tb.last[limb];
    typename B::TwRegs tw_first;
    B::template load_tw<B::NPH - 1, false>(tid, tw, tw_first);
    const u64* src = in + p * N;
    u64 x[E];
    if constexpr (B::kLdsIO) {   // coalesced 16-byte loads of the wave's source region, permuted through its LDS rows (ntt_core.h stage_gather)
        unsigned addr[E];
        const long shift = B::gather_plan(tid, g, addr);
        u64 v[E];
        B::stage_load(tid, v, src + shift);
        B::stage_gather(tid, x, v, lds, addr);
    } else {
#pragma unroll
        for (int kk = 0; kk < E; ++kk) {
            const unsigned pos = (unsigned)tid * E + kk;
            const unsigned e = 2u * (__brev(pos) >> (32 - LOGN)) + 1u;
            const unsigned e2 = (g * e) & (2u * N - 1u);
            x[kk] = src[__brev((e2 - 1u) >> 1) >> (32 - LOGN)];
        }
    }
    if constexpr (B::G::T <= 64) __syncthreads();   // single-wave geometries have no barrier inside the transform
    InvChain<B, B::NPH - 1, kUnit>::run_with(tid, x, lds, tw, last, lc, tw_first);
    B::inv_canon(x, lc);
    B::store_top(tid, x, out + p * N);
}

// ------------------------------------------------------------------------------------------------
// A6: fused ciphertext x ciphertext multiply.  One workgroup per (ciphertext pair, limb).
//   4 forward NTTs -> register-resident dyadic tensor product -> 3 inverse NTTs, as three rounds of
//   [forward, forward, inverse].  HBM traffic: 4 reads + 3 writes of a residue poly.
// ------------------------------------------------------------------------------------------------
constexpr int kCtMulOcc = 2;   // workgroups per CU the generic fused multiply's register budget is sized for
template <class Arith, int LOGN, int LOGE, bool IN_NTT, bool OUT_NTT>
__global__ __launch_bounds__(1 << (LOGN - LOGE), (LOGE <= 3 && LOGN - LOGE <= 9) ? 4 : kCtMulOcc) void ct_mul_kernel(u64* __restrict__ out3, const u64* __restrict__ a2,
                                                                        const u64* __restrict__ b2, DevTables<Arith> tb) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    static_assert(LOGE == kFusedLoge, "the fused kernels read the fused twiddle layout (DevTables::fwd4 / inv4)");
    constexpr int E = B::E, N = B::G::N;
    __shared__ __attribute__((aligned(16))) u64 lds[(IN_NTT && OUT_NTT) ? 16 : B::G::lds_words()];
    int tid = threadIdx.x;
    const size_t L = (size_t)tb.n_limbs;
    size_t bi;
    int limb;
    block_item_limb(tb, blockIdx.x, bi, limb);
    const LimbConst lc = tb.lc[limb];
    const u64* src_a = a2 + ((bi * 2) * L + limb) * N;  // component c at + c*L*N
    const u64* src_b = b2 + ((bi * 2) * L + limb) * N;
    u64* dst = out3 + ((bi * 3) * L + limb) * N;
    const size_t cstride = L * N;
    const InvLast<typename B::Tw> last = tb.last[limb];
    constexpr bool kLazy = Arith::kFold && !OUT_NTT;   // products feed the inverse NTT unreduced (< 2 kMulB q/1024)
    static_assert(!kLazy || IN_NTT || B::kFwdOutBound <= kLimitPartner, "lazy forward outputs must satisfy mul60's bound");
    constexpr int kInvIn = Arith::kFold ? 2 * kMulB : kUnit;

    // Schedule (F = forward NTT, I = inverse NTT + store), at most four polynomials live in registers:
    //   round 0: S0 = F(a0);  S1 = F(b0);                                 I(S0*S1) -> c0
    //   round 1: S2 = F(b1), S0 = S0*S2;  x = F(a1), S0 += x*S1, S2 = x*S2;   I(S0) -> c1
    //   round 2:                                                           I(S2) -> c2
    // A loop of three rounds holding two inlined instances of F and one of I (44 KiB of code).  (A loop of seven single
    // steps with one instance each was 2.5 % slower: hipcc copied all three kept polynomials at every loop latch.)
    u64 S0[E], S1[E], S2[E];
    const typename B::Tw* const twf = tb.fwd4 + (size_t)limb * N;
    const typename B::Tw* const twi = tb.inv4 + (size_t)limb * N;
    auto forward = [&](u64 (&x)[E], const u64* src, bool reduce_out, bool first) {
        if (IN_NTT) { B::load_bot(tid, x, src); return; }
        B::template load_top<true>(tid, x, src);   // streamed once: non-temporal
        if (!first) lds_barrier();
        FwdChain<B, 0>::template run<false>(tid, x, lds, twf, lc);   // (early twiddle requests spill here)
        if (!Arith::kFold) B::fwd_canon(x, lc);
        else if (reduce_out) {
            if constexpr (kLazy) B::fwd_reduce_partner(x, lc);
            else B::fwd_canon(x, lc);
        }
    };
#pragma unroll 1   // fully unrolled (4 forward + 3 inverse instances, ~80 KiB of code) it overflows the instruction cache: 12 % slower
    for (int round = 0; round < 3; ++round) {
        // an opaque thread id keeps the per-thread twiddle fetches of every transform where they are used (left visible,
        // hipcc hoists and shares them across transforms and the ~60 registers they then pin spill)
        asm volatile("" : "+v"(tid));
        u64 x[E];
        if (round < 2) {
            forward(x, round == 0 ? src_a : src_b + cstride, round == 1, round == 0);      // a0 | b1
            if (round == 0) {
#pragma unroll
                for (int k = 0; k < E; ++k) S0[k] = x[k];
            } else {
#pragma unroll
                for (int k = 0; k < E; ++k) {
                    S2[k] = x[k];
                    S0[k] = kLazy ? FoldArith::mul60(S0[k], x[k], (u32)lc.d) : Arith::mul_var(S0[k], x[k], lc);
                }
            }
            asm volatile("" : "+v"(tid));   // the two forward instances must not share (and keep alive) their twiddle fetches
            forward(x, round == 0 ? src_b : src_a + cstride, round == 0, false);           // b0 | a1
            if (round == 0) {
#pragma unroll
                for (int k = 0; k < E; ++k) S1[k] = x[k];
            } else {
#pragma unroll
                for (int k = 0; k < E; ++k) {
                    if (kLazy) {
                        S0[k] = S0[k] + FoldArith::mul60(x[k], S1[k], (u32)lc.d);
                        S2[k] = FoldArith::mul60(x[k], S2[k], (u32)lc.d);
                    } else {
                        S0[k] = add_mod(S0[k], Arith::mul_var(x[k], S1[k], lc), lc.q);
                        S2[k] = Arith::mul_var(x[k], S2[k], lc);
                    }
                }
            }
        }
        asm volatile("" : "+v"(tid));
        typename B::TwRegs tw_first;
        if (!OUT_NTT) B::template load_tw<B::NPH - 1, false>(tid, twi, tw_first);
        if (round == 0) {
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = kLazy ? FoldArith::mul60(S0[k], S1[k], (u32)lc.d) : Arith::mul_var(S0[k], S1[k], lc);
        } else if (round == 1) {
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = S0[k];
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = S2[k];
        }
        u64* d = dst + (size_t)round * cstride;
        if (OUT_NTT) {
            B::store_bot(tid, x, d);
        } else {
            // after a forward chain the wave's region is its own again; after another inverse chain other waves may still be
            // reading it (the inverse's last exchange is all-to-all)
            if (IN_NTT ? round > 0 : round == 2) lds_barrier();
            InvChain<B, B::NPH - 1, kInvIn>::run_with(tid, x, lds, twi, last, lc, tw_first);
            B::inv_canon(x, lc);
            B::template store_top<true>(tid, x, d);
        }
    }
}


// The same operation with the forward transforms in PAIRS (FwdChain2: a0|b0, then a1|b1) and the three inverse transforms
// alternating between the two LDS buffers (no barrier between them).  Coefficient-domain input only.
template <class Arith, int LOGN, int LOGE, bool OUT_NTT>
__global__ __launch_bounds__(1 << (LOGN - LOGE), 2) void ct_mul_dual_kernel(u64* __restrict__ out3, const u64* __restrict__ a2,
                                                                         const u64* __restrict__ b2, DevTables<Arith> tb) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    static_assert(LOGE == kFusedLoge, "the fused kernels read the fused twiddle layout (DevTables::fwd4 / inv4)");
    constexpr int E = B::E, N = B::G::N, W = B::G::lds_words();
    __shared__ __attribute__((aligned(16))) u64 lds[2 * W];
    int tid = threadIdx.x;
    const size_t L = (size_t)tb.n_limbs;
    size_t bi;
    int limb;
    block_item_limb(tb, blockIdx.x, bi, limb);
    const LimbConst lc = tb.lc[limb];
    const u64* src_a = a2 + ((bi * 2) * L + limb) * N;
    const u64* src_b = b2 + ((bi * 2) * L + limb) * N;
    u64* dst = out3 + ((bi * 3) * L + limb) * N;
    const size_t cstride = L * N;
    // products of forward outputs as the transforms left them (ntt_core.h tensor), straight into the inverse - or, FoldArith with NTT-domain output (round 6),
    // canonicalised and stored
    constexpr bool kLazy = B::kLazyProducts && (!OUT_NTT || Arith::kFold);
    typedef NttBody<Arith, LOGN, LOGE, 0, kUnit, kLazy> BI;  // the inverse transforms' body: no entry conversion in front of lazy products
    constexpr bool kScaledProducts = kLazy && Arith::kFoldCore && !Arith::kFold;   // FoldScaledArith: the products carry the scale twice
    const InvLast<typename B::Tw> last = kScaledProducts ? tb.last2[limb] : tb.last[limb];
    constexpr int kInvIn = kLazy ? B::kProdInvIn : kUnit;
    const typename B::Tw* const twf = tb.fwd4 + (size_t)limb * N;
    const typename B::Tw* const twi = tb.inv4 + (size_t)limb * N;

    u64 D0[E], D1[E], D2[E];
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        asm volatile("" : "+v"(tid));   // keeps the twiddle fetches of the two passes apart (see ct_mul_kernel)
        u64 x[E], y[E];
        B::template load_top<true>(tid, x, src_a + (size_t)r * cstride);   // (requesting pass 1's operands during pass 0 spills 46 registers at N = 8192: 13 % slower)
        B::template load_top<true>(tid, y, src_b + (size_t)r * cstride);
        if (r) lds_barrier();           // pass 0's last exchange was wave-local, but pass 1's first write crosses waves
        FwdChain2<B, 0>::run(tid, x, y, lds, lds + W, twf, lc);
        if constexpr (kLazy) B::prod_partner(y, lc);     // of every product below exactly one factor is reduced: b0, b1
        else { B::fwd_canon(x, lc); B::fwd_canon(y, lc); }
        if (r == 0) {
#pragma unroll
            for (int k = 0; k < E; ++k) { D0[k] = x[k]; D1[k] = y[k]; }      // a0^, b0^
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const u64 a0 = D0[k], b0 = D1[k];
                if constexpr (kLazy) {
                    B::tensor(a0, x[k], b0, y[k], D0[k], D1[k], D2[k], lc);
                } else {
                    D0[k] = Arith::mul_var(a0, b0, lc);
                    D1[k] = add_mod(Arith::mul_var(a0, y[k], lc), Arith::mul_var(x[k], b0, lc), lc.q);
                    D2[k] = Arith::mul_var(x[k], y[k], lc);
                }
            }
        }
    }
    if (OUT_NTT) {
        if constexpr (kLazy) {
#pragma unroll
            for (int k = 0; k < E; ++k) { D0[k] = FoldArith::canon_small(D0[k], lc); D1[k] = FoldArith::canon_small(D1[k], lc); D2[k] = FoldArith::canon_small(D2[k], lc); }
        }
        B::store_bot(tid, D0, dst);
        B::store_bot(tid, D1, dst + cstride);
        B::store_bot(tid, D2, dst + 2 * cstride);
    } else {
        // c0 | c1 side by side, then c2.  Before the single chain: its wave-local writes go to buffer 0, which other waves read
        // in the pair's last exchange
        asm volatile("" : "+v"(tid));
        InvChain2<BI, B::NPH - 1, kInvIn>::run(tid, D0, D1, lds, lds + W, twi, last, lc);
        B::inv_canon(D0, lc);
        B::template store_top<true>(tid, D0, dst);
        B::inv_canon(D1, lc);
        B::template store_top<true>(tid, D1, dst + cstride);
        asm volatile("" : "+v"(tid));
        lds_barrier();
        InvChain<BI, B::NPH - 1, kInvIn>::run(tid, D2, lds, twi, last, lc);
        B::inv_canon(D2, lc);
        B::template store_top<true>(tid, D2, dst + 2 * cstride);
    }
}

// All four forward transforms and all three inverse transforms of the workgroup share their twiddle fetches (FwdChain4 / InvChain3
// on two LDS buffers): 2 x 64 KiB of per-thread twiddle reads per workgroup instead of 4 x 64 KiB in ct_mul_dual_kernel.
// TRACE (diagnostics only, dpfhe_debug_ct_mul_trace): thread 0 of every workgroup stamps s_memrealtime (100 MHz) at the kernel's
// milestones into trace[blockIdx.x * 12 ..]: 0 start, 1 first operand word arrived, 2 forward transforms done, 3 tensor product done,
// 4 inverse transforms done, 5 stores issued, 6 stores drained, 7 = HW_ID | XCC_ID << 32 (where the workgroup ran), 8 prologue done,
// 9 first operand's loads issued, 10 all loads issued, 11 whole first operand arrived.
// The stamps stay in scalar registers until the end (the kernel has no vector register to spare); `dep` is a value the milestone must have
// produced: as an input operand it orders the stamp after it (and makes the compiler wait for it, if it is a load).
constexpr int kTraceWords = 12;
template <bool TRACE>
__device__ __forceinline__ u64 trace_stamp(u64 dep) {
    u64 t = 0;
    if constexpr (TRACE) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
    return t;
}
template <class Arith, int LOGN, int LOGE, bool TRACE = false, bool OUT_NTT = false>
__global__ __launch_bounds__(1 << (LOGN - LOGE), 2) void ct_mul_quad_kernel(u64* __restrict__ out3, const u64* __restrict__ a2,
                                                                         const u64* __restrict__ b2, DevTables<Arith> tb, u64* __restrict__ trace = nullptr) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    static_assert(B::kLazyProducts && LOGE == kFusedLoge, "a policy with lazy products (ntt_core.h prod), fused twiddle layout");
    typedef NttBody<Arith, LOGN, LOGE, 0, kUnit, true> BI;   // the inverse transforms' body: their input is the register-resident products
    constexpr bool kScaledProducts = Arith::kFoldCore && !Arith::kFold;   // FoldScaledArith: the products carry the scale twice
    constexpr int E = B::E, N = B::G::N, W = B::G::lds_words();
    __shared__ __attribute__((aligned(16))) u64 lds[2 * W];
    const int tid = threadIdx.x;
    const u64 ts0 = trace_stamp<TRACE>((u64)tid);
    const size_t L = (size_t)tb.n_limbs;
    size_t bi;
    int limb;
    block_item_limb(tb, blockIdx.x, bi, limb);
    const LimbConst lc = tb.lc[limb];
    const u64* src_a = a2 + ((bi * 2) * L + limb) * N;
    const u64* src_b = b2 + ((bi * 2) * L + limb) * N;
    u64* dst = out3 + ((bi * 3) * L + limb) * N;
    const size_t cstride = L * N;
    const InvLast<typename B::Tw> last = kScaledProducts ? tb.last2[limb] : tb.last[limb];
    constexpr int kInvIn = B::kProdInvIn;
    u64 x[E], y[E], z[E], w[E];
    const u64 ts_p = trace_stamp<TRACE>((u64)(uintptr_t)src_a ^ (u64)lc.q);   // prologue done: kernel arguments and limb constants are in registers
    B::template load_top<true>(tid, x, src_a);
    const u64 ts_x = trace_stamp<TRACE>((u64)tid);                              // the first operand's 16 loads are issued
    B::template load_top<true>(tid, y, src_b);
    B::template load_top<true>(tid, z, src_a + cstride);
    B::template load_top<true>(tid, w, src_b + cstride);
    const u64 ts_i = trace_stamp<TRACE>((u64)tid);                              // all 64 loads are issued
    const u64 ts1 = trace_stamp<TRACE>(x[0]);
    const u64 ts_xl = trace_stamp<TRACE>(x[E - 1]);                             // the whole first operand has arrived
    FwdChain4<B, 0>::run(tid, x, y, z, w, lds, lds + W, tb.fwd4 + (size_t)limb * N, lc);
    B::prod_partner(y, lc);     // of every product below exactly one factor is reduced: b0, b1
    B::prod_partner(w, lc);
    const u64 ts2 = trace_stamp<TRACE>(w[E - 1]);
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const u64 a0 = x[k], b0 = y[k], a1 = z[k], b1 = w[k];
        B::tensor(a0, a1, b0, b1, x[k], y[k], z[k], lc);
    }
    const u64 ts3 = trace_stamp<TRACE>(z[E - 1]);
    if constexpr (OUT_NTT) {
        // round 6: the products stay in the NTT domain (DPFHE_OUT_NTT: a caller that sums them there and transforms the total back - the inverse transform
        // is linear - never runs 3 of the 7 transforms per pair): canonical words in forward-output order
        static_assert(Arith::kFold && !TRACE, "the pinned primes (the products of a scaled-fold limb carry the scale twice)");
#pragma unroll
        for (int k = 0; k < E; ++k) { x[k] = FoldArith::canon_small(x[k], lc); y[k] = FoldArith::canon_small(y[k], lc); z[k] = FoldArith::canon_small(z[k], lc); }
        B::store_bot(tid, x, dst);
        B::store_bot(tid, y, dst + cstride);
        B::store_bot(tid, z, dst + 2 * cstride);
        return;
    }
    constexpr bool kNtStore = true;   // the 3 GiB of products are written once and read by another kernel much later: around the Infinity Cache (-2.4 %)
    InvChain3<BI, B::NPH - 1, kInvIn>::run(tid, x, y, z, lds, lds + W, tb.inv4 + (size_t)limb * N, last, lc);
    const u64 ts4 = trace_stamp<TRACE>(z[E - 1]);
    B::inv_canon(x, lc);
    B::template store_top<kNtStore>(tid, x, dst);
    B::inv_canon(y, lc);
    B::template store_top<kNtStore>(tid, y, dst + cstride);
    B::inv_canon(z, lc);
    B::template store_top<kNtStore>(tid, z, dst + 2 * cstride);
    if constexpr (TRACE) {
        const u64 ts5 = trace_stamp<TRACE>((u64)tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u64 ts6 = trace_stamp<TRACE>((u64)tid);
        if (tid == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            u64* t = trace + (size_t)blockIdx.x * kTraceWords;
            t[0] = ts0; t[1] = ts1; t[2] = ts2; t[3] = ts3; t[4] = ts4; t[5] = ts5; t[6] = ts6;
            t[7] = (u64)hw | ((u64)xcc << 32);
            t[8] = ts_p; t[9] = ts_x; t[10] = ts_i; t[11] = ts_xl;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// N1: relinearisation with RNS-digit evaluation keys (no special prime).
//   c2 = sum_j d_j g_j (mod Q) with d_j = [c2]_{q_j} (digit j = limb j of c2, coefficient domain) and g_j the CRT basis
//   element of limb j, so   (c0', c1') = (c0, c1) + sum_j d_j (.) evk_j,   evk_j = (-(a_j s) + e_j + g_j s^2, a_j).
// One workgroup per (ciphertext, limb i): for every digit j: load d_j, reduce it mod q_i, forward NTT in registers,
// multiply-accumulate with the two key polynomials (NTT domain, L2-resident: 2 L^2 N words in total); then two inverse
// NTTs, add c0_i / c1_i, store.  L + 2 transforms per workgroup; HBM: L + 2 reads and 2 writes of a residue polynomial.
// ------------------------------------------------------------------------------------------------
constexpr unsigned kRelinRotMajor = 0x80000000u;   // flag in relin_kernel's n_outer: key-major workgroup ids (launch_impl.h launch_relin)
template <class Arith>
__device__ __forceinline__ u64 canon_any(u64 v, const LimbConst& lc) {  // any v < 2^64 -> [0, q)
    if (Arith::kFold) return FoldArith::canon(v, lc);
    else return ShoupArith::mul_var(v, 1, lc);
}

// MODE 0: relinearisation - input has 3 components, digits come from c2, both c0 and c1 are added back.
// MODE 1: key switch after a Galois automorphism (N3) - input has 2 components, digits come from c1, only c0 is added:
//         (c0', c1') = (c0 + sum_j d_j b_j, sum_j d_j a_j).
// MODE 4: MODE 3 with the result LEFT IN THE NTT DOMAIN over Q P (no inverse transforms, nothing added): the giant steps of a packed
//         matrix-vector product sum these over the rotations and pay ONE inverse transform + divide-by-P for the whole sum
//         (dpfhe_switch_key_qp).  Ld transforms per workgroup instead of Ld + 2.
// MODE 2 / 3: the inner product of HYBRID key switching (one special prime P = the context's LAST limb).  The data
//         lives on the first Ld = L - 1 limbs; the kernel runs for all L limbs (including P), takes the Ld digits from
//         c2 (MODE 2, 3-component input) or c1 (MODE 3, 2-component input) and writes t = sum_j d_j (.) key_j to a work
//         buffer [batch][2][L][N] with nothing added back; dpfhe's rescale-add pass then divides by P and adds (c0, c1).
template <class Arith, int LOGN, int LOGE, int MODE, bool TRACE = false>
__global__ __launch_bounds__(1 << (LOGN - LOGE), (Arith::kFold ? 2 : 1)) void relin_kernel(u64* __restrict__ out2, const u64* __restrict__ in3,
                                                                       const u64* __restrict__ evk, size_t key_stride, unsigned key_group,
                                                                       unsigned n_outer, DevTables<Arith> tb, u64* trace = nullptr) {
    // the digits are residues of ANOTHER limb: any word below 2^60.  FoldArith's forward transform takes them as they are (NttBody FWD_IN = kRedB: the first
    // stage's fused multiply-add reduces its addend for free - 7 instructions per word less than canonicalising first); generic primes canonicalise.
    typedef NttBody<Arith, LOGN, LOGE, 0, Arith::kFold ? kRedB : kUnit> B;
    static_assert(LOGE == kFusedLoge, "the fused kernels read the fused twiddle layout (DevTables::fwd4 / inv4)");
    constexpr int E = B::E, N = B::G::N;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    int tid = threadIdx.x;
    const int L = tb.n_limbs;
    size_t bi;
    int limb;
    // La = the limbs THIS launch works on: all of them, or one arithmetic class of a mixed context (devtables.h n_active / active_map; `limb` below is then
    // an index into the class and is mapped to the limb's number in the context right after)
    const unsigned La = tb.n_active ? (unsigned)tb.n_active : (unsigned)L;
    if (n_outer & kRelinRotMajor) {
        // round 4: ALL workgroups of a key (L limbs x key_group items) on ONE XCD, limb-major: the items' digits (read by every limb's
        // workgroup) and the key tiles (read by every item's workgroup) both come from HBM once - the layout below fetched the digits L times
        const unsigned n_keys = (n_outer & ~kRelinRotMajor) / La, per_key = La * key_group;
        const unsigned q = blockIdx.x >> 3, w = q % per_key, key = (q / per_key) * 8u + (blockIdx.x & 7u);
        if (key >= n_keys) return;
        limb = (int)(w / key_group);
        bi = (size_t)key * key_group + w % key_group;
    } else if (n_outer) {
        // `key_group` consecutive items share a key (the giant steps of several tokens): their workgroups for one limb get ids that are
        // equal modulo 8 and adjacent above that, i.e. the same XCD at the same time - the key tiles come from HBM once per group
        const unsigned q = blockIdx.x >> 3, inner = q % key_group, outer = (q / key_group) * 8u + (blockIdx.x & 7u);
        if (outer >= n_outer) return;
        limb = (int)(outer % La);
        bi = (size_t)(outer / La) * key_group + inner;
    } else {
        bi = blockIdx.x / La;
        limb = (int)(blockIdx.x % La);
    }
    if (tb.n_active) limb = (int)((tb.active_map >> (4u * (unsigned)limb)) & 15u);
    const LimbConst lc = tb.lc[limb];
    const InvLast<typename B::Tw> last = tb.last[limb];
    constexpr int kInComps = (MODE == 0 || MODE == 2) ? 3 : 2;
    constexpr bool kHybrid = MODE >= 2;
    const int Ld = kHybrid ? L - 1 : L;                                   // limbs of the data (= number of digits)
    const u64* c2 = in3 + ((bi * kInComps + (kInComps - 1)) * Ld) * N;  // digit j at + j*N
    evk += (bi / key_group) * key_stride;   // per-item keys (batched rotations; key_group consecutive items share one); stride 0 = one key
    u64 acc0[E], acc1[E];
#pragma unroll
    for (int k = 0; k < E; ++k) acc0[k] = acc1[k] = 0;
    int lazy_terms = 0;
    // TRACE (diagnostic builds only, tools/relin_trace.py): per workgroup, where the digit loop spends its time - sums over the digits of
    // [digit words arrived, transform done, first key tile arrived, its products done, second key tile arrived, its products done]
    const u64 tr_start = trace_stamp<TRACE>((u64)tid);
    u64 tr_sum[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int j = 0; j < Ld; ++j) {
        asm volatile("" : "+v"(tid));
        u64 x[E];
        const u64 tr0 = trace_stamp<TRACE>((u64)tid);
        B::load_top(tid, x, c2 + (size_t)j * N);
        if constexpr (!Arith::kFold) {
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = canon_any<Arith>(x[k], lc);   // [c2]_{q_j} mod q_i
        }
        u64 tr_dep = 0;
        if constexpr (TRACE) {
#pragma unroll
            for (int k = 0; k < E; ++k) tr_dep |= x[k];
        }
        const u64 tr1 = trace_stamp<TRACE>(tr_dep);
        if (j > 0) lds_barrier();
        FwdChain<B, 0>::template run<false>(tid, x, lds, tb.fwd4 + (size_t)limb * N, lc);
        if constexpr (TRACE) {
            tr_dep = 0;
#pragma unroll
            for (int k = 0; k < E; ++k) tr_dep |= x[k];
        }
        const u64 tr2 = trace_stamp<TRACE>(tr_dep);
        const u64* k0 = evk + (((size_t)j * 2 + 0) * L + limb) * N;      // key polynomials, NTT domain (window-0 mapping)
        const u64* k1 = evk + (((size_t)j * 2 + 1) * L + limb) * N;
        if (Arith::kFold) {
            static_assert(13 * kMulB + kRedB <= kWord, "13 lazily added products + one reduced word must fit a 64-bit word");
            if (lazy_terms == 13) {  // 13 products + one reduced word stay below 15 q
#pragma unroll
                for (int k = 0; k < E; ++k) { acc0[k] = FoldArith::reduce(acc0[k], lc); acc1[k] = FoldArith::reduce(acc1[k], lc); }
                lazy_terms = 1;
            }
            ++lazy_terms;
        } else {
            B::fwd_canon(x, lc);
        }
        {   // one key polynomial at a time keeps the live set at x + acc0 + acc1 + one key (no spills at 2 waves/SIMD)
            u64 e[E];
            // key polynomials are NTT-domain tiles: transposed through LDS (rows private to the wave) where that measured
            // faster (N = 8192: -7 %), in registers otherwise (N = 4096: the LDS path is 5 % slower at 2 waves per SIMD)
            constexpr bool kLdsKeys = B::kLdsIO && LOGN >= 13;
            // (key tiles with the non-temporal hint - so that the L2 would rather keep the digits - measured 6.6 % slower and 3.7 % MORE fabric reads: profiles/r06_giant_traffic.txt)
            if constexpr (kLdsKeys) B::load_bot_lds(tid, e, k0, lds); else B::load_bot(tid, e, k0);
            u64 tr_d = 0;
            if constexpr (TRACE) {
#pragma unroll
                for (int k = 0; k < E; ++k) tr_d |= e[k];
            }
            const u64 tr3 = trace_stamp<TRACE>(tr_d);
#pragma unroll
            for (int k = 0; k < E; ++k)
                acc0[k] = Arith::kFold ? acc0[k] + FoldArith::mul60(x[k], e[k], (u32)lc.d) : add_mod(acc0[k], Arith::mul_var(x[k], e[k], lc), lc.q);
            if constexpr (TRACE) {
                tr_d = 0;
#pragma unroll
                for (int k = 0; k < E; ++k) tr_d |= acc0[k];
            }
            const u64 tr4 = trace_stamp<TRACE>(tr_d);
            asm volatile("" ::: "memory");
            if constexpr (kLdsKeys) B::load_bot_lds(tid, e, k1, lds); else B::load_bot(tid, e, k1);
            if constexpr (TRACE) {
                tr_d = 0;
#pragma unroll
                for (int k = 0; k < E; ++k) tr_d |= e[k];
            }
            const u64 tr5 = trace_stamp<TRACE>(tr_d);
#pragma unroll
            for (int k = 0; k < E; ++k)
                acc1[k] = Arith::kFold ? acc1[k] + FoldArith::mul60(x[k], e[k], (u32)lc.d) : add_mod(acc1[k], Arith::mul_var(x[k], e[k], lc), lc.q);
            if constexpr (TRACE) {
                tr_d = 0;
#pragma unroll
                for (int k = 0; k < E; ++k) tr_d |= acc1[k];
                const u64 tr6 = trace_stamp<TRACE>(tr_d);
                tr_sum[0] += tr1 - tr0; tr_sum[1] += tr2 - tr1; tr_sum[2] += tr3 - tr2; tr_sum[3] += tr4 - tr3; tr_sum[4] += tr5 - tr4; tr_sum[5] += tr6 - tr5;
            }
        }
    }
    if constexpr (TRACE) {
        u64 tr_d = 0;
#pragma unroll
        for (int k = 0; k < E; ++k) tr_d |= acc1[k];
        const u64 tr_end = trace_stamp<TRACE>(tr_d);
        if (tid == 0 && trace) {
            u64* t = trace + (size_t)blockIdx.x * 8;
            t[0] = tr_start; t[1] = tr_end;
#pragma unroll
            for (int i = 0; i < 6; ++i) t[2 + i] = tr_sum[i];
        }
    }
    if (Arith::kFold) {
#pragma unroll
        for (int k = 0; k < E; ++k) { acc0[k] = FoldArith::reduce(acc0[k], lc); acc1[k] = FoldArith::reduce(acc1[k], lc); }
    }
    if constexpr (MODE == 4) {   // the inner products stay in the NTT domain over Q P: canonical words, forward-output order
        if (Arith::kFold) {
#pragma unroll
            for (int k = 0; k < E; ++k) { acc0[k] = FoldArith::canon_small(acc0[k], lc); acc1[k] = FoldArith::canon_small(acc1[k], lc); }
        }
        // through the wave's own LDS rows (what it read in the last forward exchange: no workgroup barrier), non-temporal: the terms are
        // read once by the next launch and must not push the keys and digits out of the XCD's L2 (160 VALU instructions of register
        // transposition less per workgroup, too)
        if constexpr (B::kLdsIO && Arith::kFold) {
            B::template store_bot_lds<true>(tid, acc0, out2 + ((bi * 2 + 0) * L + limb) * N, lds);
            wave_sync();
            B::template store_bot_lds<true>(tid, acc1, out2 + ((bi * 2 + 1) * L + limb) * N, lds);
        } else {
            B::store_bot(tid, acc0, out2 + ((bi * 2 + 0) * L + limb) * N);
            B::store_bot(tid, acc1, out2 + ((bi * 2 + 1) * L + limb) * N);
        }
        return;
    }
    constexpr int kInvIn = Arith::kFold ? 2 * kMulB : kUnit;
#pragma unroll 1
    for (int c = 0; c < 2; ++c) {
        asm volatile("" : "+v"(tid));
        u64 x[E];
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = c == 0 ? acc0[k] : acc1[k];
        const bool add_back = (MODE == 0) || (MODE == 1 && c == 0);
        // the word to add back is fetched ahead of the transform where registers allow (MODE 1: one accumulator is
        // dead by then); MODE 0 with 16-byte FoldArith twiddles would spill, so it fetches afterwards
        constexpr bool kPrefetch = !Arith::kFold;
        u64 orig[E];
        if (add_back && kPrefetch) B::load_top(tid, orig, in3 + ((bi * kInComps + c) * L + limb) * N);
        if (c > 0) lds_barrier();   // the first inverse follows wave-local work only; the second follows an all-to-all read
        InvChain<B, B::NPH - 1, kInvIn>::run(tid, x, lds, tb.inv4 + (size_t)limb * N, last, lc);
        if (add_back && !kPrefetch) B::load_top(tid, orig, in3 + ((bi * kInComps + c) * L + limb) * N);
        B::inv_canon(x, lc);
        if (add_back) {
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = add_mod(x[k], orig[k], lc.q);
        }
        B::store_top(tid, x, out2 + ((bi * 2 + c) * L + limb) * N);
    }
}

// The same key switch with the digit transforms SIDE BY SIDE (round 2, after the fused multiply showed what the per-thread twiddle
// reads cost): the first four digits go through FwdChain4 (one set of twiddle fetches, two reused LDS buffers) while the
// accumulators are not live yet, the remaining one to three through FwdChain2 / FwdChain, the two inverse transforms through
// InvChain2.  Ld + 2 twiddle streams become 3 (Ld <= 4: 2).  FoldArith, 4 <= Ld <= 7.
template <class Arith, int LOGN, int LOGE, int MODE>
__global__ __launch_bounds__(1 << (LOGN - LOGE), 2) void relin_shared_kernel(u64* __restrict__ out2, const u64* __restrict__ in3,
                                                                           const u64* __restrict__ evk, size_t key_stride, unsigned key_group,
                                                                           unsigned n_outer, DevTables<Arith> tb) {
    typedef NttBody<Arith, LOGN, LOGE, 0, kRedB> B;   // digits enter the forward transforms as they are (< 2^60): relin_kernel
    static_assert(Arith::kFold && LOGE == kFusedLoge, "FoldArith, fused twiddle layout");
    constexpr int E = B::E, N = B::G::N, W = B::G::lds_words();
    __shared__ __attribute__((aligned(16))) u64 lds[2 * W];
    int tid = threadIdx.x;
    const int L = tb.n_limbs;
    size_t bi;
    int limb;
    if (n_outer & kRelinRotMajor) {
        // round 4: ALL workgroups of a key (L limbs x key_group items) on ONE XCD, limb-major: the items' digits (read by every limb's
        // workgroup) and the key tiles (read by every item's workgroup) both come from HBM once - the layout below fetched the digits L times
        const unsigned n_keys = (n_outer & ~kRelinRotMajor) / (unsigned)L, per_key = (unsigned)L * key_group;
        const unsigned q = blockIdx.x >> 3, w = q % per_key, key = (q / per_key) * 8u + (blockIdx.x & 7u);
        if (key >= n_keys) return;
        limb = (int)(w / key_group);
        bi = (size_t)key * key_group + w % key_group;
    } else if (n_outer) {
        // `key_group` consecutive items share a key (the giant steps of several tokens): their workgroups for one limb get ids that are
        // equal modulo 8 and adjacent above that, i.e. the same XCD at the same time - the key tiles come from HBM once per group
        const unsigned q = blockIdx.x >> 3, inner = q % key_group, outer = (q / key_group) * 8u + (blockIdx.x & 7u);
        if (outer >= n_outer) return;
        limb = (int)(outer % (unsigned)L);
        bi = (size_t)(outer / (unsigned)L) * key_group + inner;
    } else {
        bi = blockIdx.x / (unsigned)L;
        limb = (int)(blockIdx.x % (unsigned)L);
    }
    const LimbConst lc = tb.lc[limb];
    const InvLast<typename B::Tw> last = tb.last[limb];
    constexpr int kInComps = (MODE == 0 || MODE == 2) ? 3 : 2;
    constexpr bool kHybrid = MODE >= 2;
    const int Ld = kHybrid ? L - 1 : L;
    const u64* c2 = in3 + ((bi * kInComps + (kInComps - 1)) * Ld) * N;
    evk += (bi / key_group) * key_stride;
    const typename B::Tw* const twf = tb.fwd4 + (size_t)limb * N;
    static_assert(7 * kMulB + kRedB <= kWord, "seven lazily added products must fit a 64-bit word");
    u64 acc0[E], acc1[E];
    auto digit = [&](u64 (&x)[E], int j) { B::load_top(tid, x, c2 + (size_t)j * N); };   // [c]_{q_j}: reduced mod q_i by the transform's first stage
    auto mac = [&](const u64 (&d)[E], int j, bool first) {   // one key polynomial at a time
        u64 e[E];
        B::load_bot(tid, e, evk + (((size_t)j * 2 + 0) * L + limb) * N);
#pragma unroll
        for (int k = 0; k < E; ++k) acc0[k] = (first ? 0 : acc0[k]) + FoldArith::mul60(d[k], e[k], (u32)lc.d);
        asm volatile("" ::: "memory");
        B::load_bot(tid, e, evk + (((size_t)j * 2 + 1) * L + limb) * N);
#pragma unroll
        for (int k = 0; k < E; ++k) acc1[k] = (first ? 0 : acc1[k]) + FoldArith::mul60(d[k], e[k], (u32)lc.d);
        asm volatile("" ::: "memory");
    };
    {
        u64 x[E], y[E], z[E], w[E];
        digit(x, 0); digit(y, 1); digit(z, 2); digit(w, 3);
        FwdChain4<B, 0>::run(tid, x, y, z, w, lds, lds + W, twf, lc);
        mac(x, 0, true); mac(y, 1, false); mac(z, 2, false); mac(w, 3, false);
    }
    const int rem = Ld - 4;
    if (rem >= 2) {
        asm volatile("" : "+v"(tid));
        u64 x[E], y[E];
        digit(x, 4); digit(y, 5);
        lds_barrier();
        FwdChain2<B, 0>::run(tid, x, y, lds, lds + W, twf, lc);
        mac(x, 4, false); mac(y, 5, false);
    }
    if (rem & 1) {
        asm volatile("" : "+v"(tid));
        u64 x[E];
        digit(x, Ld - 1);
        lds_barrier();
        FwdChain<B, 0>::template run<false>(tid, x, lds, twf, lc);
        mac(x, Ld - 1, false);
    }
#pragma unroll
    for (int k = 0; k < E; ++k) { acc0[k] = FoldArith::reduce(acc0[k], lc); acc1[k] = FoldArith::reduce(acc1[k], lc); }
    if constexpr (MODE == 4) {
#pragma unroll
        for (int k = 0; k < E; ++k) { acc0[k] = FoldArith::canon_small(acc0[k], lc); acc1[k] = FoldArith::canon_small(acc1[k], lc); }
        B::store_bot(tid, acc0, out2 + ((bi * 2 + 0) * L + limb) * N);
        B::store_bot(tid, acc1, out2 + ((bi * 2 + 1) * L + limb) * N);
        return;
    }
    asm volatile("" : "+v"(tid));
    InvChain2<B, B::NPH - 1, 2 * kMulB>::run(tid, acc0, acc1, lds, lds + W, tb.inv4 + (size_t)limb * N, last, lc);
    B::inv_canon(acc0, lc);
    if (MODE == 0 || MODE == 1) {
        u64 orig[E];
        B::load_top(tid, orig, in3 + ((bi * kInComps + 0) * L + limb) * N);
#pragma unroll
        for (int k = 0; k < E; ++k) acc0[k] = add_mod(acc0[k], orig[k], lc.q);
    }
    B::store_top(tid, acc0, out2 + ((bi * 2 + 0) * L + limb) * N);
    B::inv_canon(acc1, lc);
    if (MODE == 0) {
        u64 orig[E];
        B::load_top(tid, orig, in3 + ((bi * kInComps + 1) * L + limb) * N);
#pragma unroll
        for (int k = 0; k < E; ++k) acc1[k] = add_mod(acc1[k], orig[k], lc.q);
    }
    B::store_top(tid, acc1, out2 + ((bi * 2 + 1) * L + limb) * N);
}

// ------------------------------------------------------------------------------------------------
// N3: HOISTED rotations - k rotations of one ciphertext share the digit decomposition and its forward
// transforms.  `digits` holds NTT(lift([c1]_{q_j})) for every digit j and limb i ([Ld][L][N], forward-output order).  In that
// order position p carries the evaluation at psi^(2 brv(p) + 1), and sigma_g: a(X) -> a(X^g) only permutes evaluation points:
//   NTT(sigma_g a)[p] = NTT(a)[p'],   2 brv(p') + 1 = g (2 brv(p) + 1)  (mod 2N).
// One workgroup per (rotation, limb, key component): gather the permuted digit words (L2-resident), multiply-accumulate with the
// rotation's key polynomial, ONE inverse transform, store to the work buffer [k][2][L][N] (divide-by-P + add sigma_g(c0) follow
// in rescale_kernel).  Per rotation and limb: 2 transforms instead of Ld + 2.  (Measured at N=8192, 5+1 limbs, 31 rotations:
// 69 us against 86 us for the un-hoisted pass; replacing the gather by coalesced loads changes nothing - what bounds one
// token is each CU streaming its 320 KB of key tiles at ~10 B/clk, i.e. the 122 MB of keys spread over few workgroups.)
// ------------------------------------------------------------------------------------------------
template <class Arith, int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE), (Arith::kFold ? 2 : 1)) void hoisted_ks_kernel(u64* __restrict__ work, const u64* __restrict__ digits,
                                                                                           const u64* __restrict__ keys, size_t key_stride, GaloisElts elts,
                                                                                           unsigned n_items, unsigned n_tiles, DevTables<Arith> tb) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    static_assert(LOGE == kFusedLoge, "the fused kernels read the fused twiddle layout (DevTables::fwd4 / inv4)");
    constexpr int E = B::E, N = B::G::N;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    int tid = threadIdx.x;
    const int L = tb.n_limbs, Ld = L - 1;
    // one workgroup per (rotation, limb, key component): twice the workgroups of the fused key-switch kernel and half the serial
    // chain each - the k rotations of one token would otherwise fill a fraction of the chip with long-running workgroups
    // Several input ciphertexts (tokens) share the rotations: block = ((rotation * L + limb) * 2 + comp) * n_items + token, so
    // that the workgroups reading one key tile are neighbours in time (the tile is fetched once into L2 / Infinity Cache).
    // ... and on the same XCD: ids are dealt to the 8 XCDs round-robin, so id = ((tile / 8) * n_items + token) * 8 + tile % 8 keeps the
    // n_items workgroups of a tile on XCD tile % 8, adjacent in time - the key tile comes from HBM once, the other tokens hit that L2
    const unsigned q = blockIdx.x >> 3, token = q % n_items, tile = (q / n_items) * 8u + (blockIdx.x & 7u);
    if (tile >= n_tiles) return;
    const int comp = (int)(tile & 1u);
    const unsigned La = tb.n_active ? (unsigned)tb.n_active : (unsigned)L;   // limbs this launch works on (one class of a mixed context, or all)
    const size_t rot = (tile >> 1) / La;
    int limb = (int)((tile >> 1) % La);
    if (tb.n_active) limb = (int)((tb.active_map >> (4u * (unsigned)limb)) & 15u);
    const size_t item = rot * n_items + token;        // output / work item
    digits += (size_t)token * (size_t)(L - 1) * L * N;
    const LimbConst lc = tb.lc[limb];
    const InvLast<typename B::Tw> last = tb.last[limb];
    const unsigned g = elts.v[rot];
    const u64* evk = keys + rot * key_stride;
    // source positions of this thread's E output positions p = tid E + kk
    unsigned src[E];
#pragma unroll
    for (int kk = 0; kk < E; ++kk) {
        const unsigned p = (unsigned)tid * E + kk;
        const unsigned e = 2u * (__brev(p) >> (32 - LOGN)) + 1u;
        const unsigned e2 = (g * e) & (2u * N - 1u);
        src[kk] = __brev((e2 - 1u) >> 1) >> (32 - LOGN);
    }
    u64 acc[E], x[E], e[E];
#pragma unroll
    for (int k = 0; k < E; ++k) acc[k] = 0;
    // software pipeline over the digits: the key tile and the permuted digit words of digit j + 1 are requested before the
    // products of digit j (a workgroup is alone on its SIMDs here: nothing else hides the L2 / HBM latency of the key tiles)
    B::load_bot(tid, e, evk + ((size_t)comp * L + limb) * N);
    {
        const u64* d = digits + (size_t)limb * N;
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = d[src[k]];
    }
#pragma unroll 1
    for (int j = 0; j < Ld; ++j) {
        u64 en[E], xn[E];
        const int jn = j + 1 < Ld ? j + 1 : j;   // the last iteration re-requests its own tile (cache hit) instead of branching
        B::load_bot(tid, en, evk + (((size_t)jn * 2 + comp) * L + limb) * N);
        {
            const u64* d = digits + ((size_t)jn * L + limb) * N;
#pragma unroll
            for (int k = 0; k < E; ++k) xn[k] = d[src[k]];
        }
#pragma unroll
        for (int k = 0; k < E; ++k)
            acc[k] = Arith::kFold ? FoldArith::mac_var(acc[k], x[k], e[k], lc) : add_mod(acc[k], Arith::mul_var(x[k], e[k], lc), lc.q);
#pragma unroll
        for (int k = 0; k < E; ++k) { e[k] = en[k]; x[k] = xn[k]; }
    }
    constexpr int kInvIn = Arith::kFold ? kRedB : kUnit;
    InvChain<B, B::NPH - 1, kInvIn>::run(tid, acc, lds, tb.inv4 + (size_t)limb * N, last, lc);
    B::inv_canon(acc, lc);
    B::store_top(tid, acc, work + ((item * 2 + comp) * L + limb) * N);
}

// Both key components in one workgroup (used when the grid is large enough without the split: several tokens): the permuted digit
// words are gathered ONCE for the two inner products and the two inverse transforms run side by side (InvChain2, two LDS buffers).
// One workgroup per (rotation, limb, token); same XCD-aware id layout (tile = (rotation, limb)).
template <class Arith, int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE), 2) void hoisted_ks2_kernel(u64* __restrict__ work, const u64* __restrict__ digits,
                                                                           const u64* __restrict__ keys, size_t key_stride, GaloisElts elts,
                                                                           unsigned n_items, unsigned n_tiles, DevTables<Arith> tb) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    static_assert(Arith::kFold && LOGE == kFusedLoge, "FoldArith, fused twiddle layout");
    constexpr int E = B::E, N = B::G::N, W = B::G::lds_words();
    __shared__ __attribute__((aligned(16))) u64 lds[2 * W];
    const int tid = threadIdx.x;
    const int L = tb.n_limbs, Ld = L - 1;
    const unsigned q = blockIdx.x >> 3, token = q % n_items, tile = (q / n_items) * 8u + (blockIdx.x & 7u);
    if (tile >= n_tiles) return;
    const size_t rot = tile / (unsigned)L;
    const int limb = (int)(tile % (unsigned)L);
    const size_t item = rot * n_items + token;
    digits += (size_t)token * (size_t)(L - 1) * L * N;
    const LimbConst lc = tb.lc[limb];
    const InvLast<typename B::Tw> last = tb.last[limb];
    const unsigned g = elts.v[rot];
    const u64* evk = keys + rot * key_stride;
    unsigned src[E];
#pragma unroll
    for (int kk = 0; kk < E; ++kk) {
        const unsigned p = (unsigned)tid * E + kk;
        const unsigned e = 2u * (__brev(p) >> (32 - LOGN)) + 1u;
        const unsigned e2 = (g * e) & (2u * N - 1u);
        src[kk] = __brev((e2 - 1u) >> 1) >> (32 - LOGN);
    }
    u64 acc0[E], acc1[E], x[E], e[E];
#pragma unroll
    for (int k = 0; k < E; ++k) acc0[k] = acc1[k] = 0;
    static_assert(7 * kMulB + kRedB <= kWord, "up to seven lazily added products fit a 64-bit word");
    B::load_bot(tid, e, evk + ((size_t)0 * L + limb) * N);
    {
        const u64* d = digits + (size_t)limb * N;
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = d[src[k]];
    }
#pragma unroll 1
    for (int j = 0; j < Ld; ++j) {
        // pipeline: the second component's tile is requested before the first component's products, the next digit's words and
        // first tile before the second component's products
        u64 e1[E], xn[E];
        B::load_bot(tid, e1, evk + (((size_t)j * 2 + 1) * L + limb) * N);
#pragma unroll
        for (int k = 0; k < E; ++k) acc0[k] = FoldArith::mac_var(acc0[k], x[k], e[k], lc);
        const int jn = j + 1 < Ld ? j + 1 : j;
        B::load_bot(tid, e, evk + (((size_t)jn * 2 + 0) * L + limb) * N);
        {
            const u64* d = digits + ((size_t)jn * L + limb) * N;
#pragma unroll
            for (int k = 0; k < E; ++k) xn[k] = d[src[k]];
        }
#pragma unroll
        for (int k = 0; k < E; ++k) acc1[k] = FoldArith::mac_var(acc1[k], x[k], e1[k], lc);
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = xn[k];
    }
    InvChain2<B, B::NPH - 1, kRedB>::run(tid, acc0, acc1, lds, lds + W, tb.inv4 + (size_t)limb * N, last, lc);
    B::inv_canon(acc0, lc);
    B::store_top(tid, acc0, work + ((item * 2 + 0) * L + limb) * N);
    B::inv_canon(acc1, lc);
    B::store_top(tid, acc1, work + ((item * 2 + 1) * L + limb) * N);
}

// (N3, round 3's hoisted_qp_kernel - one workgroup per (rotation, limb, token) on the transform geometry - lived here; round 4 replaced it by
// kernels_misc.h hoisted_qp_stream_kernel: no transform in that pass, so no transform geometry, and 0.99 x instead of 2.3 x its algorithmic bytes.)

}  // namespace dpfhe
