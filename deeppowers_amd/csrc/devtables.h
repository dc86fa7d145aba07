// devtables.h - device-side table handles passed to the kernels by value.
#pragma once
#include "modarith.h"

namespace dpfhe {

struct QuartersTop;   // ntt_quarters.h

template <class Tw>
struct InvLast {  // per limb: last inverse stage twiddles with N^-1 folded in
    Tw w_last;    // psi^-brv(1) * N^-1
    Tw w_ninv;    // N^-1
};

// Twiddle tables are stored in the layout of the kernel geometry that reads them (tables.h permute_window0):
// fwd/inv for the batched NTT kernels (launch.h ntt_loge), fwd4/inv4 for the fused kernels, which always run 16
// words per thread.  The two coincide unless the NTT geometry is not LOGE = 4 (N = 8192).
template <class Arith>
struct DevTables {
    const typename Arith::Tw* fwd;          // [L][N]  psi^brv(i)
    const typename Arith::Tw* inv;          // [L][N]  psi^-brv(i)
    const typename Arith::Tw* fwd4;
    const typename Arith::Tw* inv4;
    // split transform (N > 16384): fwd / inv / last then hold one N2-point table per (limb, block) - [L][n_sub][N2] and
    // [L][n_sub] - and the first log2(n_sub) stages run in ntt_top_kernel with these workgroup-uniform twiddles
    const typename Arith::Tw* top_fwd;      // [L][n_sub]  psi^brv(i), entry m + i of the full table, i < n_sub
    const typename Arith::Tw* top_inv;      // [L][n_sub]  psi^-brv(i)
    const InvLast<typename Arith::Tw>* top_last;  // [L]   last inverse stage with N^-1 folded in
    int n_sub;                              // 1: single-kernel transform
    // N = 8192 in "halves" form (ntt_halves.h; null at other ring degrees): per limb the two 4096-point sub-tree tables (roots 2 and 3 of the
    // N = 8192 table) in the (12, 4) kernel layout, the column stage's twiddle psi^brv(1), and the inverse column stage with N^-1 folded in
    const typename Arith::Tw* hfwd;         // [L][2][4096]
    const typename Arith::Tw* hinv;         // [L][2][4096]
    const typename Arith::Tw* htop_fwd;     // [L]
    const InvLast<typename Arith::Tw>* htop_last;  // [L]  {psi^-brv(1) N^-1, N^-1}
    // N = 16384 in "quarters" form (ntt_quarters.h; FoldArith contexts, null elsewhere): per limb the four 4096-point sub-tree tables (roots 4..7 of the
    // N = 16384 table) in the (12, 4) kernel layout, the forward column stages' twiddles, the inverse ones' (psi^-brv(2), psi^-brv(3)) and the last stage
    const typename Arith::Tw* qfwd;         // [L][4][4096]
    const typename Arith::Tw* qinv;         // [L][4][4096]
    const struct QuartersTop* qtop_fwd;     // [L]
    const typename Arith::Tw* qtop_inv;     // [L][2]
    const InvLast<typename Arith::Tw>* qtop_last;  // [L]  {psi^-brv(1) N^-1, N^-1}
    const InvLast<typename Arith::Tw>* last;  // [L]
    // FoldScaledArith class only: the same with s^-1 = 2^-(60-k) folded in - the last stage of an inverse transform whose input is a PRODUCT of two
    // scaled words (the fused multiply's lazy tensor step).  Null elsewhere.
    const InvLast<typename Arith::Tw>* last2;  // [L]
    const LimbConst* lc;                    // [L]
    int n_limbs;
    // Per-limb arithmetic classes (round 6, dpfhe_cabi.hip): a launch may cover only SOME limbs of the context - those whose primes this policy
    // serves.  n_active = 0: all n_limbs limbs (the uniform contexts; block b works on limb b mod n_limbs).  Otherwise block b works on item
    // b / n_active and limb (active_map >> 4 (b mod n_active)) & 15; tables and data stay indexed by the limb's number in the context
    // (n_limbs = the stride), so a class's tables simply leave the other limbs' slots unused.  Contexts of more than 16 limbs are uniform.
    int n_active;
    unsigned long long active_map;
};

// (item, limb) of a workgroup of the transform / fused-multiply kernels; `blk` = blockIdx.x
template <class TB>
DPF_HD void block_item_limb(const TB& tb, size_t blk, size_t& item, int& limb) {
    if (tb.n_active) {
        item = blk / (unsigned)tb.n_active;
        limb = (int)((tb.active_map >> (4u * (unsigned)(blk % (unsigned)tb.n_active))) & 15u);
    } else {
        item = blk / (unsigned)tb.n_limbs;
        limb = (int)(blk % (unsigned)tb.n_limbs);
    }
}

// the batched transforms' view of a context with per-limb arithmetic classes (kernels.h ntt_classes_kernel, dpfhe_cabi.hip mixed_layout)
struct MixedTables {
    const void* fwd;            // [L][N] twiddles, slot l in limb l's format, kernel layout of the batched transforms
    const void* inv;
    const void* last;           // [L] InvLast
    const LimbConst* lc;        // [L], each as its class reads it (tables.h limb_const_of_class)
    int n_limbs;
    unsigned long long cls_map; // 4 bits per limb: tables.h LimbClass
};

}  // namespace dpfhe
