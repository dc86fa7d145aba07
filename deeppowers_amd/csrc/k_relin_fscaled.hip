#include "launch_impl.h"
// round 6: the key-switching kernels of a context whose limbs are ALL of the FoldScaledArith limb class (2^k - d0 primes): the generic forms of relin_kernel /
// hoisted_ks_kernel / ntt_inv_galois_kernel (canonical words between the transforms and the key products), with this class's transforms and products
namespace dpfhe {
template int launch_relin<FoldScaledArith>(int, int, u64*, const u64*, const u64*, size_t, unsigned, size_t, const DevTables<FoldScaledArith>&, hipStream_t);
template int launch_hoisted_ks<FoldScaledArith>(int, u64*, const u64*, const u64*, size_t, const unsigned*, size_t, size_t, const DevTables<FoldScaledArith>&, hipStream_t);
template int launch_ntt_inv_galois<FoldScaledArith>(int, u64*, const u64*, const unsigned*, size_t, size_t, const DevTables<FoldScaledArith>&, hipStream_t);
}
