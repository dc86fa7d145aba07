#include "launch_impl.h"
namespace dpfhe {
template int launch_ct_mul<FoldArith>(int, unsigned, u64*, const u64*, const u64*, size_t, const DevTables<FoldArith>&, hipStream_t);
template int launch_relin<FoldArith>(int, int, u64*, const u64*, const u64*, size_t, unsigned, size_t, const DevTables<FoldArith>&, hipStream_t);
template int launch_hoisted_ks<FoldArith>(int, u64*, const u64*, const u64*, size_t, const unsigned*, size_t, size_t, const DevTables<FoldArith>&, hipStream_t);
}
