#include "launch_impl.h"
namespace dpfhe {
template int launch_ct_mul<FoldArith>(int, unsigned, u64*, const u64*, const u64*, size_t, const DevTables<FoldArith>&, hipStream_t);
template int launch_relin<FoldArith>(int, int, u64*, const u64*, const u64*, size_t, unsigned, size_t, const DevTables<FoldArith>&, hipStream_t);
template int launch_hoisted_ks<FoldArith>(int, u64*, const u64*, const u64*, size_t, const unsigned*, size_t, size_t, const DevTables<FoldArith>&, hipStream_t);
}
#ifdef DPFHE_DIAGNOSTICS   // diagnostic builds only: 8 words per workgroup of the last traced launch (kernels.h relin_kernel TRACE)
extern "C" int dpfhe_debug_relin_trace_read(unsigned long long* host, size_t max_blocks) {
    if (!dpfhe::g_relin_trace || !host) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    const size_t nb = dpfhe::g_relin_trace_blocks < max_blocks ? dpfhe::g_relin_trace_blocks : max_blocks;
    if (hipMemcpy(host, dpfhe::g_relin_trace, nb * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -3;
    return (int)nb;
}
#endif
