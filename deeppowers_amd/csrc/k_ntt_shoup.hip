#include "launch_impl.h"
namespace dpfhe {
template int launch_ntt<ShoupArith>(int, bool, u64*, const u64*, size_t, const DevTables<ShoupArith>&, hipStream_t);
template int launch_ntt_inv_galois<ShoupArith>(int, u64*, const u64*, const unsigned*, size_t, size_t, const DevTables<ShoupArith>&, hipStream_t);
}
