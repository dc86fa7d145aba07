#include "launch_impl.h"
namespace dpfhe {
template int launch_ntt<FoldArith>(int, bool, u64*, const u64*, size_t, const DevTables<FoldArith>&, hipStream_t);
template int launch_ntt_inv_galois<FoldArith>(int, u64*, const u64*, const unsigned*, size_t, size_t, const DevTables<FoldArith>&, hipStream_t);
}
