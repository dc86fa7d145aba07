// base_ext.h - argument block and host launcher of the exact base extension / scale-and-round kernels (kernels_bx.h).
#pragma once
#include <hip/hip_runtime.h>

#include "modarith.h"

namespace dpfhe {

// ------------------------------------------------------------------------------------------------
// Round 4 (towards configs[4] with a non-linear layer): EXACT base extension between limb ranges of one context, and the
// scale-and-round of an exact (BFV-style) ciphertext multiply.  Per coefficient, from the residues x_i mod q_i of the source limbs:
//   mixed-radix (Garner) digits  v_0 = x_0,  v_k = ((x_k - v_0) q_0^-1 - v_1) q_1^-1 ... mod q_k,   X = v_0 + q_0 (v_1 + q_1 (v_2 + ...)) in [0, Qs);
//   centred: X > floor(Qs / 2) (compared digit by digit from the top) means the represented integer is X - Qs;
//   MODE 0 (extend):      out_j = X mod p_j                                   for every destination limb j
//   MODE 1 (scale-round): the sources are mul * x_i (the input times a small multiplier, e.g. the plaintext modulus t), the input also
//                         holds the destination limbs, and  out_j = (mul x_j - X) Qs^-1 mod p_j  =  round(mul x / Qs) mod p_j  exactly.
// At most 10 source limbs (600 bits: the ciphertext modulus of the level a multiply runs at, or the workspace limbs on the way back) and 20
// destination limbs; the constants travel as kernel arguments (3.1 KiB of the 4 KiB argument segment).  One thread per pair of words.
// ------------------------------------------------------------------------------------------------
constexpr int kBxMaxSrc = 10, kBxMaxDst = 20, kBxMaxSrcGeneric = 8;
static_assert(sizeof(int) * (2 + kBxMaxSrc + kBxMaxDst) + 8 * (kBxMaxSrc * kBxMaxSrc + 2 * kBxMaxSrc + kBxMaxSrc * kBxMaxDst + 3 * kBxMaxDst) <= 3400,
              "BaseExtArgs + the kernel's other arguments must stay inside the 4 KiB kernel-argument segment");
struct BaseExtArgs {
    int n_src, n_dst;
    int src_limb[kBxMaxSrc], dst_limb[kBxMaxDst];   // indices into the context's limb constants
    u64 inv[kBxMaxSrc][kBxMaxSrc];                  // inv[i][k] = q_i^-1 mod q_k, i < k (source limbs)
    u64 half[kBxMaxSrc];                            // mixed-radix digits of floor(Qs / 2)
    u64 q_mod[kBxMaxSrc][kBxMaxDst];                // q_i mod p_j
    u64 Q_mod[kBxMaxDst];                           // Qs mod p_j
    u64 Q_inv[kBxMaxDst];                           // Qs^-1 mod p_j              (MODE 1)
    u64 mul_src[kBxMaxSrc], mul_dst[kBxMaxDst];     // the multiplier mod q_i / mod p_j (MODE 1)
};
// mode 0 = extend, 1 = scale-and-round; returns 0, or -1 when a.n_src has no compiled kernel.  Launch errors are left in hipGetLastError().
template <class Arith>
int launch_base_extend(int mode, u64* out, size_t out_stride, const u64* in, size_t in_stride, size_t in_dst_off, const BaseExtArgs& a, const LimbConst* lcs, int n,
                       int chunks, unsigned grid, hipStream_t s);

}  // namespace dpfhe
