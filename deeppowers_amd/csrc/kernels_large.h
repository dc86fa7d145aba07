// kernels_large.h - ring degrees above 8192 (device).
//
// The fused multiply and key-switch kernels of kernels.h keep whole polynomials in one workgroup's registers and LDS, which stops at
// N = 8192 (one 1024-thread workgroup at N = 16384 has 128 VGPRs per thread; four polynomials of 16 words per thread do not fit).
// Above, the same operations are COMPOSED (dpfhe_cabi.hip: ct_mul_composed, key_switch_composed) from the batched transforms of
// kernels.h / ntt_top.h and the streaming kernels below, one HBM pass each, 16 bytes per lane, one workgroup per 512-word chunk of
// one residue polynomial.  All words canonical in and out; results equal the fused kernels' (tests/test_gpu_parity.py (test_large_rings_multiply_and_key_switch_composed_behind_the_c_abi and the slicing test)).
#pragma once
#include "kernels_misc.h"

namespace dpfhe {

// (a0, a1) (x) (b0, b1) -> (a0 b0, a0 b1 + a1 b0, a1 b1), NTT domain.  a2, b2: [batch][2][L][N]; out3: [batch][3][L][N].
template <class Arith>
__global__ __launch_bounds__(256) void tensor3_kernel(u64* __restrict__ out3, const u64* __restrict__ a2, const u64* __restrict__ b2, const LimbConst* lcs,
                                                      int n_limbs, int n, int chunks) {
    const int chunk = (int)(blockIdx.x % chunks);
    const int limb = (int)((blockIdx.x / chunks) % n_limbs);
    const size_t pair = blockIdx.x / chunks / n_limbs;
    const int w0 = chunk * 512 + threadIdx.x * 2;
    if (w0 >= n) return;
    const LimbConst lc = lcs[limb];
    const size_t poly = (size_t)n_limbs * n, off = (size_t)limb * n + w0;
    const U64x2 a0 = *reinterpret_cast<const U64x2*>(a2 + (pair * 2 + 0) * poly + off), a1 = *reinterpret_cast<const U64x2*>(a2 + (pair * 2 + 1) * poly + off);
    const U64x2 b0 = *reinterpret_cast<const U64x2*>(b2 + (pair * 2 + 0) * poly + off), b1 = *reinterpret_cast<const U64x2*>(b2 + (pair * 2 + 1) * poly + off);
    U64x2 d0, d1, d2;
    d0.a = Arith::mul_var(a0.a, b0.a, lc); d0.b = Arith::mul_var(a0.b, b0.b, lc);
    d1.a = add_mod(Arith::mul_var(a0.a, b1.a, lc), Arith::mul_var(a1.a, b0.a, lc), lc.q);
    d1.b = add_mod(Arith::mul_var(a0.b, b1.b, lc), Arith::mul_var(a1.b, b0.b, lc), lc.q);
    d2.a = Arith::mul_var(a1.a, b1.a, lc); d2.b = Arith::mul_var(a1.b, b1.b, lc);
    *reinterpret_cast<U64x2*>(out3 + (pair * 3 + 0) * poly + off) = d0;
    *reinterpret_cast<U64x2*>(out3 + (pair * 3 + 1) * poly + off) = d1;
    *reinterpret_cast<U64x2*>(out3 + (pair * 3 + 2) * poly + off) = d2;
}

// RNS digits of the key-switched component, lifted to every limb: out[item][j][i][k] = src[item][j][k] mod q_i  (src = component
// `comp` of an `in_comps`-component ciphertext, coefficient domain, limb j holds values < q_j).  out: [batch][L][L][N].
template <class Arith>
__global__ __launch_bounds__(256) void lift_rns_digits_kernel(u64* __restrict__ out, const u64* __restrict__ in, int in_comps, int comp, const LimbConst* lcs,
                                                              int n_limbs, int n, int chunks) {
    const int chunk = (int)(blockIdx.x % chunks);
    const int limb = (int)((blockIdx.x / chunks) % n_limbs);
    const size_t dg = blockIdx.x / chunks / n_limbs, item = dg / (unsigned)n_limbs, digit = dg % (unsigned)n_limbs;
    const int w0 = chunk * 512 + threadIdx.x * 2;
    if (w0 >= n) return;
    const LimbConst lc = lcs[limb];
    const U64x2 v = *reinterpret_cast<const U64x2*>(in + (((item * in_comps + comp) * n_limbs) + digit) * n + w0);
    U64x2 r;
    if ((int)digit == limb) r = v;   // already canonical modulo its own prime
    else {
        r.a = Arith::kFold ? FoldArith::canon(v.a, lc) : ShoupArith::mul_var(v.a, 1, lc);
        r.b = Arith::kFold ? FoldArith::canon(v.b, lc) : ShoupArith::mul_var(v.b, 1, lc);
    }
    *reinterpret_cast<U64x2*>(out + ((item * n_limbs + digit) * n_limbs + limb) * n + w0) = r;
}

// acc[item][c][i] = sum_{j < n_digits} x[item][j][i] (.) evk[j][c][i], c = 0, 1 (NTT domain; x: [batch][n_digits][L][N], evk: [n_digits][2][L][N];
// its tiles are re-read from L2).  acc: [batch][2][L][N].  n_digits = L (RNS-digit keys) or L - 1 (hybrid keys: the last limb is the special prime,
// the sum is divided by it afterwards).  Per-item keys (round 5: the batched rotations and the giant steps of a packed layer above N = 8192):
// item i of this launch uses the key at evk + ((item0 + i) / key_group) * key_stride; key_stride = 0 is one key for the whole batch.
template <class Arith>
__global__ __launch_bounds__(256) void key_inner_product_kernel(u64* __restrict__ acc, const u64* __restrict__ x, const u64* __restrict__ evk, const LimbConst* lcs,
                                                                int n_digits, int n_limbs, int n, int chunks, size_t key_stride = 0, unsigned key_group = 1,
                                                                unsigned item0 = 0) {
    const int chunk = (int)(blockIdx.x % chunks);
    const int limb = (int)((blockIdx.x / chunks) % n_limbs);
    const size_t item = blockIdx.x / chunks / n_limbs;
    const int w0 = chunk * 512 + threadIdx.x * 2;
    if (w0 >= n) return;
    const LimbConst lc = lcs[limb];
    evk += ((item + item0) / key_group) * key_stride;
    U64x2 s0{0, 0}, s1{0, 0};
#pragma unroll 2
    for (int j = 0; j < n_digits; ++j) {
        const U64x2 d = *reinterpret_cast<const U64x2*>(x + ((item * n_digits + j) * n_limbs + limb) * n + w0);
        const U64x2 k0 = *reinterpret_cast<const U64x2*>(evk + (((size_t)j * 2 + 0) * n_limbs + limb) * n + w0);
        const U64x2 k1 = *reinterpret_cast<const U64x2*>(evk + (((size_t)j * 2 + 1) * n_limbs + limb) * n + w0);
        s0.a = add_mod(s0.a, Arith::mul_var(d.a, k0.a, lc), lc.q); s0.b = add_mod(s0.b, Arith::mul_var(d.b, k0.b, lc), lc.q);
        s1.a = add_mod(s1.a, Arith::mul_var(d.a, k1.a, lc), lc.q); s1.b = add_mod(s1.b, Arith::mul_var(d.b, k1.b, lc), lc.q);
    }
    *reinterpret_cast<U64x2*>(acc + ((item * 2 + 0) * n_limbs + limb) * n + w0) = s0;
    *reinterpret_cast<U64x2*>(acc + ((item * 2 + 1) * n_limbs + limb) * n + w0) = s1;
}

// out2[item][c] += in[item][c] for the components c whose bit is set in `mask` (in: [batch][in_comps][L][N], coefficient domain)
__global__ __launch_bounds__(256) void add_back_kernel(u64* __restrict__ out2, const u64* __restrict__ in, int in_comps, int mask, const LimbConst* lcs, int n_limbs,
                                                       int n, int chunks) {
    const int chunk = (int)(blockIdx.x % chunks);
    const int limb = (int)((blockIdx.x / chunks) % n_limbs);
    const size_t pc = blockIdx.x / chunks / n_limbs, item = pc >> 1;
    const int comp = (int)(pc & 1);
    const int w0 = chunk * 512 + threadIdx.x * 2;
    if (w0 >= n || !((mask >> comp) & 1)) return;
    const u64 q = lcs[limb].q;
    U64x2* o = reinterpret_cast<U64x2*>(out2 + ((item * 2 + comp) * n_limbs + limb) * n + w0);
    const U64x2 v = *reinterpret_cast<const U64x2*>(in + ((item * in_comps + comp) * n_limbs + limb) * n + w0);
    U64x2 r = *o;
    r.a = add_mod(r.a, v.a, q); r.b = add_mod(r.b, v.b, q);
    *o = r;
}

}  // namespace dpfhe
