#include "launch_impl.h"
// round 6: the transform kernels of the F64WideArith limb class (primes of 47 ... 50 bits: doubles, with reductions inside the transforms - modarith.h)
namespace dpfhe {
template int launch_ntt<F64WideArith>(int, bool, u64*, const u64*, size_t, const DevTables<F64WideArith>&, hipStream_t);
}
