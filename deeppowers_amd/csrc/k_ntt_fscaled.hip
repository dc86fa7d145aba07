#include "launch_impl.h"
// round 6: the transform kernels of the FoldScaledArith limb class (2^k - d0 primes carried scaled to 2^60 - d - modarith.h)
namespace dpfhe {
template int launch_ntt<FoldScaledArith>(int, bool, u64*, const u64*, size_t, const DevTables<FoldScaledArith>&, hipStream_t);
}
