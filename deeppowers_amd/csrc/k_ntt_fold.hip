#include "launch_impl.h"
namespace dpfhe {
template int launch_ntt<FoldArith>(int, bool, u64*, const u64*, size_t, const DevTables<FoldArith>&, hipStream_t);
template int launch_ntt_inv_galois<FoldArith>(int, u64*, const u64*, const unsigned*, size_t, size_t, const DevTables<FoldArith>&, hipStream_t);
}
#ifdef DPFHE_DIAGNOSTICS   // diagnostic builds only: 8 words per workgroup of the last traced forward transform (kernels_trace.h)
extern "C" int dpfhe_debug_ntt_trace_read(unsigned long long* host, size_t max_blocks) {
    if (!dpfhe::g_ntt_trace || !host) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    const size_t nb = dpfhe::g_ntt_trace_blocks < max_blocks ? dpfhe::g_ntt_trace_blocks : max_blocks;
    if (hipMemcpy(host, dpfhe::g_ntt_trace, nb * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -3;
    return (int)nb;
}
#endif
