#include "launch_impl.h"
namespace dpfhe {
template int launch_ct_mul<F64WideArith>(int, unsigned, u64*, const u64*, const u64*, size_t, const DevTables<F64WideArith>&, hipStream_t);
}
