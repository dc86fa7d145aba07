#include "launch_impl.h"
namespace dpfhe {
template int launch_ntt<FoldArith>(int, bool, u64*, const u64*, size_t, const DevTables<FoldArith>&, hipStream_t);
}
