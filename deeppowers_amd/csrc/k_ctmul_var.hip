#include "launch_impl.h"
// the named forms of the fused multiply that dpfhe_ctx_autotune probes, and the traced quad form (diagnostics)
namespace dpfhe {
template int launch_ct_mul_variant<FoldArith>(int, int, u64*, const u64*, const u64*, size_t, const DevTables<FoldArith>&, hipStream_t);
template int launch_ct_mul_variant<ShoupArith>(int, int, u64*, const u64*, const u64*, size_t, const DevTables<ShoupArith>&, hipStream_t);
template int launch_ct_mul_trace<FoldArith>(int, u64*, const u64*, const u64*, size_t, const DevTables<FoldArith>&, u64*, hipStream_t);
template int launch_ct_mul_trace<ShoupArith>(int, u64*, const u64*, const u64*, size_t, const DevTables<ShoupArith>&, u64*, hipStream_t);
}
