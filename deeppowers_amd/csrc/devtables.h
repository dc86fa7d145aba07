// devtables.h - device-side table handles passed to the kernels by value.
#pragma once
#include "modarith.h"

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

}  // namespace dpfhe
