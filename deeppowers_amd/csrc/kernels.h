// kernels.h - the gfx950 kernels of the FHE hot path (device code; included by dpfhe_cabi.hip only).
//
// A1/A2 batched NTT, A3 dyadic ops, A6 fused ct x ct multiply, A7 ct x pt
// matvec, A8 shard-local reduce.  No reference counterpart (section 0).  Integer modular arithmetic:
// no MFMA; the bounds are 64-bit-multiply issue rate (NTT, ct_mul) and HBM bandwidth (dyadic, matvec).
#pragma once
#include <hip/hip_runtime.h>

#include "ntt_core.h"

namespace dpfhe {

template <class Tw>
struct InvLast {  // per limb: last inverse stage twiddles with N^-1 folded in
    Tw w_last;    // psi^-brv(1) * N^-1
    Tw w_ninv;    // N^-1
};

template <class Arith>
struct DevTables {
    const typename Arith::Tw* fwd;          // [L][N]  psi^brv(i)
    const typename Arith::Tw* inv;          // [L][N]  psi^-brv(i)
    const InvLast<typename Arith::Tw>* last;  // [L]
    const LimbConst* lc;                    // [L]
    int n_limbs;
};

// ------------------------------------------------------------------------------------------------
// register-resident transforms shared by the NTT kernels and the fused ct x ct kernel
// ------------------------------------------------------------------------------------------------
template <class B, int P>
struct FwdChain {
    static __device__ __forceinline__ void run(int tid, u64 (&x)[B::E], u64* lds, const typename B::Tw* tw, const LimbConst& lc) {
        B::template fwd_phase<P>(tid, x, tw, lc);
        if constexpr (P + 1 < B::NPH) {
            if (P > 0) __syncthreads();  // previous exchange fully read before the buffer is rewritten
            B::template lds_write<P, P, true>(tid, x, lds);
            __syncthreads();
            B::template lds_read<P, P + 1, true>(tid, x, lds);
            FwdChain<B, P + 1>::run(tid, x, lds, tw, lc);
        }
    }
};

template <class B, int P, int IN>
struct InvChain {
    static __device__ __forceinline__ void run(int tid, u64 (&x)[B::E], u64* lds, const typename B::Tw* tw,
                                               const InvLast<typename B::Tw>& last, const LimbConst& lc) {
        B::template inv_phase<P, IN>(tid, x, tw, last.w_last, last.w_ninv, lc);
        if constexpr (P > 0) {
            if (P < B::NPH - 1) __syncthreads();
            B::template lds_write<P - 1, P, false>(tid, x, lds);
            __syncthreads();
            B::template lds_read<P - 1, P - 1, false>(tid, x, lds);
            InvChain<B, P - 1, IN>::run(tid, x, lds, tw, last, lc);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// A1 / A2: one workgroup per residue polynomial
// ------------------------------------------------------------------------------------------------
template <class Arith, int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void ntt_fwd_kernel(u64* __restrict__ out, const u64* __restrict__ in,
                                                                      DevTables<Arith> tb) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    const int tid = threadIdx.x;
    const size_t p = blockIdx.x;
    const int limb = (int)(p % (size_t)tb.n_limbs);
    const LimbConst lc = tb.lc[limb];
    const typename B::Tw* tw = tb.fwd + (size_t)limb * B::G::N;
    u64 x[B::E];
    B::load_top(tid, x, in + p * B::G::N);
    FwdChain<B, 0>::run(tid, x, lds, tw, lc);
    B::fwd_canon(x, lc);
    B::store_bot(tid, x, out + p * B::G::N);
}

template <class Arith, int LOGN, int LOGE>
__global__ __launch_bounds__(1 << (LOGN - LOGE)) void ntt_inv_kernel(u64* __restrict__ out, const u64* __restrict__ in,
                                                                      DevTables<Arith> tb) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    __shared__ __attribute__((aligned(16))) u64 lds[B::G::lds_words()];
    const int tid = threadIdx.x;
    const size_t p = blockIdx.x;
    const int limb = (int)(p % (size_t)tb.n_limbs);
    const LimbConst lc = tb.lc[limb];
    const typename B::Tw* tw = tb.inv + (size_t)limb * B::G::N;
    const InvLast<typename B::Tw> last = tb.last[limb];
    u64 x[B::E];
    B::load_bot(tid, x, in + p * B::G::N);
    InvChain<B, B::NPH - 1, kUnit>::run(tid, x, lds, tw, last, lc);
    B::inv_canon(x, lc);
    B::store_top(tid, x, out + p * B::G::N);
}

// ------------------------------------------------------------------------------------------------
// A6: fused ciphertext x ciphertext multiply.  One workgroup per (ciphertext pair, limb).
//   4 forward NTTs (one code instance, looped) -> register-resident dyadic tensor product ->
//   3 inverse NTTs (one code instance, looped).  HBM traffic: 4 reads + 3 writes of a residue poly.
// ------------------------------------------------------------------------------------------------
template <class Arith, int LOGN, int LOGE, bool IN_NTT, bool OUT_NTT>
__global__ __launch_bounds__(1 << (LOGN - LOGE), 2) void ct_mul_kernel(u64* __restrict__ out3, const u64* __restrict__ a2,
                                                                     const u64* __restrict__ b2, DevTables<Arith> tb) {
    typedef NttBody<Arith, LOGN, LOGE> B;
    constexpr int E = B::E, N = B::G::N;
    __shared__ __attribute__((aligned(16))) u64 lds[(IN_NTT && OUT_NTT) ? 16 : B::G::lds_words()];
    const int tid = threadIdx.x;
    const size_t L = (size_t)tb.n_limbs;
    const size_t bi = blockIdx.x / L;
    const int limb = (int)(blockIdx.x % L);
    const LimbConst lc = tb.lc[limb];
    const u64* src_a = a2 + ((bi * 2) * L + limb) * N;  // component c at + c*L*N
    const u64* src_b = b2 + ((bi * 2) * L + limb) * N;
    u64* dst = out3 + ((bi * 3) * L + limb) * N;
    const size_t cstride = L * N;

    u64 A0[E], A1[E], B0[E], B1[E];
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        const u64* src = (s < 2 ? src_a : src_b) + (size_t)(s & 1) * cstride;
        u64 x[E];
        if (IN_NTT) {
            B::load_bot(tid, x, src);
        } else {
            B::load_top(tid, x, src);
            if (s > 0) __syncthreads();  // previous transform's last exchange fully read
            FwdChain<B, 0>::run(tid, x, lds, tb.fwd + (size_t)limb * N, lc);
            if (s >= 2 || !Arith::kFold) B::fwd_canon(x, lc);  // b-side operands must be < 2^60 for mul60
        }
        if (s == 0) {
#pragma unroll
            for (int k = 0; k < E; ++k) A0[k] = x[k];
        } else if (s == 1) {
#pragma unroll
            for (int k = 0; k < E; ++k) A1[k] = x[k];
        } else if (s == 2) {
#pragma unroll
            for (int k = 0; k < E; ++k) B0[k] = x[k];
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) B1[k] = x[k];
        }
    }
    // tensor product in registers: (A0,A1,B0) <- (a0 b0, a0 b1 + a1 b0, a1 b1)
    constexpr bool kLazyOut = Arith::kFold && !OUT_NTT;  // inverse NTT accepts words < 2 kMulB q / 1024
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const u64 a0 = A0[k], a1 = A1[k], b0 = B0[k], b1 = B1[k];
        if (kLazyOut) {
            const u32 d = (u32)lc.d;
            A0[k] = FoldArith::mul60(a0, b0, d);
            A1[k] = FoldArith::mul60(a0, b1, d) + FoldArith::mul60(a1, b0, d);
            B0[k] = FoldArith::mul60(a1, b1, d);
        } else {
            A0[k] = Arith::mul_var(a0, b0, lc);
            A1[k] = add_mod(Arith::mul_var(a0, b1, lc), Arith::mul_var(a1, b0, lc), lc.q);
            B0[k] = Arith::mul_var(a1, b1, lc);
        }
    }
    const InvLast<typename B::Tw> last = tb.last[limb];
#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
        u64 x[E];
        if (s == 0) {
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = A0[k];
        } else if (s == 1) {
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = A1[k];
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = B0[k];
        }
        if (OUT_NTT) {
            B::store_bot(tid, x, dst + (size_t)s * cstride);
        } else {
            if (s > 0 || !IN_NTT) __syncthreads();
            InvChain<B, B::NPH - 1, Arith::kFold ? 2 * kMulB : kUnit>::run(tid, x, lds, tb.inv + (size_t)limb * N, last, lc);
            B::inv_canon(x, lc);
            B::store_top(tid, x, dst + (size_t)s * cstride);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A3: coefficient-wise ops.  One workgroup per residue polynomial (limb constants are scalar loads),
// 16-byte accesses, every load of an iteration issued before its first use.
// ------------------------------------------------------------------------------------------------
enum DyOp { DY_MUL = 0, DY_MUL_ADD = 1, DY_ADD = 2, DY_SUB = 3, DY_NEG = 4 };

struct __attribute__((aligned(16))) U64x2 {
    u64 a, b;
};

template <class Arith, int OP>
__device__ __forceinline__ u64 dy_apply(u64 a, u64 b, u64 acc, const LimbConst& lc) {
    if (OP == DY_MUL) return Arith::mul_var(a, b, lc);
    if (OP == DY_MUL_ADD) return add_mod(acc, Arith::mul_var(a, b, lc), lc.q);
    if (OP == DY_ADD) return add_mod(a, b, lc.q);
    if (OP == DY_SUB) return sub_mod(a, b, lc.q);
    return neg_mod(a, lc.q);
}

template <class Arith, int OP>
__global__ __launch_bounds__(256) void dyadic_kernel(u64* out, const u64* a, const u64* b, const LimbConst* lcs, int n_limbs, int n) {
    const size_t p = blockIdx.x;
    const LimbConst lc = lcs[p % (size_t)n_limbs];
    const U64x2* pa = reinterpret_cast<const U64x2*>(a + p * n);
    const U64x2* pb = reinterpret_cast<const U64x2*>(b + p * n);
    U64x2* po = reinterpret_cast<U64x2*>(out + p * n);
    const int nv = n >> 1;
    constexpr int UN = 4;
    for (int base = threadIdx.x; base < nv; base += 256 * UN) {
        U64x2 va[UN], vb[UN], vc[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = base + u * 256;
            if (i < nv) {
                va[u] = pa[i];
                if (OP != DY_NEG) vb[u] = pb[i];
                if (OP == DY_MUL_ADD) vc[u] = po[i];
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = base + u * 256;
            if (i < nv) {
                U64x2 r;
                r.a = dy_apply<Arith, OP>(va[u].a, vb[u].a, vc[u].a, lc);
                r.b = dy_apply<Arith, OP>(va[u].b, vb[u].b, vc[u].b, lc);
                po[i] = r;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A8: out[c][l][:] = sum_i in[i][c][l][:]  (canonical modular sum; HBM-bound: one read per term)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_sum_kernel(u64* out, const u64* in, const LimbConst* lcs, int n_limbs, int n,
                                                         size_t count, size_t words_per_item) {
    const size_t p = blockIdx.x;  // residue polynomial within one item
    const u64 q = lcs[p % (size_t)n_limbs].q;
    const int nv = n >> 1;
    for (int i = threadIdx.x; i < nv; i += 256) {
        u64 s0 = 0, s1 = 0;
        const U64x2* src = reinterpret_cast<const U64x2*>(in + p * n) + i;
        for (size_t it = 0; it < count; ++it) {
            const U64x2 v = *reinterpret_cast<const U64x2*>(reinterpret_cast<const u64*>(src) + it * words_per_item);
            s0 = csub(s0 + v.a, q);
            s1 = csub(s1 + v.b, q);
        }
        reinterpret_cast<U64x2*>(out + p * n)[i] = U64x2{s0, s1};
    }
}

// ------------------------------------------------------------------------------------------------
// A7: y[i][c][l][:] = sum_j W[i][j][l][:] (.) x[j][c][l][:],  c in {0,1}.  128-bit lazy accumulation,
// one reduction at the end.  One workgroup per (row, limb); W is streamed once (the HBM-bound term),
// x (cols x 2 RNS polys) is re-read by every row through L2 / Infinity Cache.
// ------------------------------------------------------------------------------------------------
struct Acc128 {
    u64 lo, hi;
};
__device__ __forceinline__ void acc_mac(Acc128& acc, u64 a, u64 b) {
    const u64 lo = a * b, hi = mulhi64(a, b);
    acc.lo += lo;
    acc.hi += hi + (acc.lo < lo);
}
template <class Arith>
__device__ __forceinline__ u64 acc_reduce(const Acc128& acc, const LimbConst& lc, u64 two64_mod_q) {
    if (Arith::kFold) {  // hi * 2^64 + lo  =  hi * (2^64 mod q) + lo
        const u64 h = FoldArith::mul60(acc.hi, two64_mod_q, (u32)lc.d);  // < 2q
        const u64 l = FoldArith::reduce(acc.lo, lc);                      // < 2q
        return FoldArith::canon(h + l, lc);
    } else {
        const u64 h = ShoupArith::mul_var(acc.hi % lc.q, two64_mod_q, lc);
        return add_mod(h, acc.lo % lc.q, lc.q);
    }
}

template <class Arith>
__global__ __launch_bounds__(256) void matvec_kernel(u64* y, const u64* W, const u64* x, const LimbConst* lcs, int n_limbs, int n,
                                                     size_t cols) {
    const size_t L = (size_t)n_limbs;
    const size_t row = blockIdx.x / L;
    const int limb = (int)(blockIdx.x % L);
    const LimbConst lc = lcs[limb];
    const u64 two64 = lc.two64;
    const int nv = n >> 1;
    const size_t wstride = L * n, xstride = 2 * L * n;
    for (int i = threadIdx.x; i < nv; i += 256) {
        Acc128 a00{0, 0}, a01{0, 0}, a10{0, 0}, a11{0, 0};  // [component][word]
        const u64* wp = W + (row * cols * L + limb) * n + 2 * (size_t)i;
        const u64* xp = x + (size_t)limb * n + 2 * (size_t)i;
        size_t since = 0;
        for (size_t j = 0; j < cols; ++j) {
            const U64x2 w = *reinterpret_cast<const U64x2*>(wp + j * wstride);
            const U64x2 x0 = *reinterpret_cast<const U64x2*>(xp + j * xstride);
            const U64x2 x1 = *reinterpret_cast<const U64x2*>(xp + j * xstride + L * n);
            acc_mac(a00, w.a, x0.a); acc_mac(a01, w.b, x0.b);
            acc_mac(a10, w.a, x1.a); acc_mac(a11, w.b, x1.b);
            if (++since == 128) {  // 128 products of < 2^120 stay below 2^128 next to a reduced value
                a00 = Acc128{acc_reduce<Arith>(a00, lc, two64), 0}; a01 = Acc128{acc_reduce<Arith>(a01, lc, two64), 0};
                a10 = Acc128{acc_reduce<Arith>(a10, lc, two64), 0}; a11 = Acc128{acc_reduce<Arith>(a11, lc, two64), 0};
                since = 0;
            }
        }
        u64* yp = y + ((row * 2) * L + limb) * n + 2 * (size_t)i;
        *reinterpret_cast<U64x2*>(yp) = U64x2{acc_reduce<Arith>(a00, lc, two64), acc_reduce<Arith>(a01, lc, two64)};
        *reinterpret_cast<U64x2*>(yp + L * n) = U64x2{acc_reduce<Arith>(a10, lc, two64), acc_reduce<Arith>(a11, lc, two64)};
    }
}

}  // namespace dpfhe
