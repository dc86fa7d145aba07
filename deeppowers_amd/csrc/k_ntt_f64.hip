#include "launch_impl.h"
// round 6: the transform kernels of the F64Arith limb class (primes below 2^47, residues as doubles inside a transform - modarith.h)
namespace dpfhe {
template int launch_ntt<F64Arith>(int, bool, u64*, const u64*, size_t, const DevTables<F64Arith>&, hipStream_t);
}
