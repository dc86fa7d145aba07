#include "launch_impl.h"
namespace dpfhe {
template int launch_ct_mul<FoldScaledArith>(int, unsigned, u64*, const u64*, const u64*, size_t, const DevTables<FoldScaledArith>&, hipStream_t);
}
