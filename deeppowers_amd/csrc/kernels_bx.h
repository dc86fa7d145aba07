// kernels_bx.h - exact base extension / scale-and-round (include/dpfhe.h dpfhe_base_extend, dpfhe_scale_round); see base_ext.h for the argument block.
#pragma once
#include "base_ext.h"

namespace dpfhe {

template <class Arith>
__device__ __forceinline__ u64 bx_canon(u64 v, const LimbConst& lc) {   // any word -> [0, q)
    if constexpr (Arith::kFold) return FoldArith::canon(v, lc);
    else return ShoupArith::mul_var(v, 1, lc);
}
// The number of source limbs is a TEMPLATE parameter (one kernel per count, dpfhe_cabi.hip dispatches): every index into the digit array v
// is a compile-time constant - a run-time one, which is what the unroller leaves behind once 10 x 10 x 2 bodies exceed its budget, puts the
// array in scratch memory.  The argument block is read in place (never through a reference: that would copy its 3 KiB to scratch as well).
template <class Arith, int MODE, int NS>
__global__ __launch_bounds__(256) void base_extend_kernel(u64* __restrict__ out, size_t out_stride, const u64* __restrict__ in, size_t in_stride,
                                                          size_t in_dst_off /* MODE 1: word offset of the destination limbs inside an input item */,
                                                          BaseExtArgs a, const LimbConst* __restrict__ lcs, int n, int chunks) {
    static_assert(NS >= 1 && NS <= kBxMaxSrc, "source limbs");
    const int chunk = (int)(blockIdx.x % chunks);
    const size_t p = blockIdx.x / chunks;
    const int w0 = chunk * 512 + threadIdx.x * 2;
    if (w0 >= n) return;
    const u64* src = in + p * in_stride + w0;
    u64 v[NS][2];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const LimbConst lck = lcs[a.src_limb[k]];
        const U64x2 xv = *reinterpret_cast<const U64x2*>(src + (size_t)k * n);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u64 t = h ? xv.b : xv.a;
            if (MODE == 1) t = Arith::mul_var(t, a.mul_src[k], lck);
#pragma unroll
            for (int i = 0; i < k; ++i) t = Arith::mul_var(sub_mod(t, bx_canon<Arith>(v[i][h], lck), lck.q), a.inv[i][k], lck);
            v[k][h] = t;
        }
    }
    bool neg[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // X > floor(Qs / 2): the top-most differing digit decides
        bool gt = false, decided = false;
#pragma unroll
        for (int k = NS - 1; k >= 0; --k)
            if (!decided && v[k][h] != a.half[k]) { gt = v[k][h] > a.half[k]; decided = true; }
        neg[h] = gt;
    }
    u64* dst = out + p * out_stride + w0;
    // a ROLLED loop over the destination limbs: their constants are read from the argument segment at a run-time offset
#pragma unroll 1
    for (int j = 0; j < a.n_dst; ++j) {
        const LimbConst lcj = lcs[a.dst_limb[j]];
        U64x2 r;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u64 acc = bx_canon<Arith>(v[NS - 1][h], lcj);   // Horner from the top digit
#pragma unroll
            for (int k = NS - 2; k >= 0; --k) acc = add_mod(Arith::mul_var(acc, a.q_mod[k][j], lcj), bx_canon<Arith>(v[k][h], lcj), lcj.q);
            if (neg[h]) acc = sub_mod(acc, a.Q_mod[j], lcj.q);
            (h ? r.b : r.a) = acc;
        }
        if (MODE == 1) {
            const U64x2 xv = *reinterpret_cast<const U64x2*>(in + p * in_stride + in_dst_off + (size_t)j * n + w0);
            r.a = Arith::mul_var(sub_mod(Arith::mul_var(xv.a, a.mul_dst[j], lcj), r.a, lcj.q), a.Q_inv[j], lcj);
            r.b = Arith::mul_var(sub_mod(Arith::mul_var(xv.b, a.mul_dst[j], lcj), r.b, lcj.q), a.Q_inv[j], lcj);
        }
        *reinterpret_cast<U64x2*>(dst + (size_t)j * n) = r;
    }
}

template <class Arith>
int launch_base_extend(int mode, u64* out, size_t out_stride, const u64* in, size_t in_stride, size_t in_dst_off, const BaseExtArgs& a, const LimbConst* lcs, int n,
                       int chunks, unsigned grid, hipStream_t s) {
    static_assert(kBxMaxSrc == 10, "one kernel per source-limb count below");
    // generic primes: their 128-bit Barrett bodies exceed the unroller's budget above 8 source limbs (the digit array would go to scratch);
    // every parameter set this library builds is of the 2^60 - d form, so that path stops at 8
    if (!Arith::kFold && a.n_src > kBxMaxSrcGeneric) return -1;
#define BX_LAUNCH(M, NS) \
    case NS: if constexpr (Arith::kFold || NS <= kBxMaxSrcGeneric) hipLaunchKernelGGL((base_extend_kernel<Arith, M, NS>), dim3(grid), dim3(256), 0, s, out, out_stride, in, in_stride, in_dst_off, a, lcs, n, chunks); return 0;
#define BX_SWITCH(M) \
    switch (a.n_src) { BX_LAUNCH(M, 1) BX_LAUNCH(M, 2) BX_LAUNCH(M, 3) BX_LAUNCH(M, 4) BX_LAUNCH(M, 5) BX_LAUNCH(M, 6) BX_LAUNCH(M, 7) BX_LAUNCH(M, 8) BX_LAUNCH(M, 9) \
                       BX_LAUNCH(M, 10) default: return -1; }
    if (mode == 0) { BX_SWITCH(0) } else { BX_SWITCH(1) }
#undef BX_SWITCH
#undef BX_LAUNCH
}

}  // namespace dpfhe
