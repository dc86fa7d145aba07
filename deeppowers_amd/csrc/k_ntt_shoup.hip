#include "launch_impl.h"
namespace dpfhe {
template int launch_ntt<ShoupArith>(int, bool, u64*, const u64*, size_t, const DevTables<ShoupArith>&, hipStream_t);
}
