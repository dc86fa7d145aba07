#include "launch_impl.h"
#include "tables.h"
// round 6: one launch for the batched transforms of a context whose limbs run on different arithmetic classes (kernels.h ntt_classes_kernel)
namespace dpfhe {
int launch_ntt_classes(int log2n, bool inverse, u64* out, const u64* in, size_t npolys, const MixedTables& tb, hipStream_t s) {
    for (int l = 0; l < tb.n_limbs; ++l)
        if (((tb.cls_map >> (4 * l)) & 15) == (unsigned long long)kClassF64Wide) return 1;   // no arm for this class in the merged kernel (kernels.h)
// returns 0, -1 (no geometry), or 1: this geometry has no merged kernel - launch per class instead (the forward kernels at N = 256 and N = 16384
// spill a few registers when the four arms share one kernel; they are not instantiated)
#define NC_CASE(LN, LE)                                                                                                                  \
    if (inverse) hipLaunchKernelGGL((ntt_classes_kernel<LN, LE, false>), dim3((unsigned)npolys), dim3(Geo<LN, LE>::T), 0, s, out, in, tb); \
    else if constexpr (LN == 8 || LN == 14) return 1;                                                                                    \
    else hipLaunchKernelGGL((ntt_classes_kernel<LN, LE, true>), dim3((unsigned)npolys), dim3(Geo<LN, LE>::T), 0, s, out, in, tb)
    DPFHE_NTT_GEO_SWITCH(log2n, NC_CASE)
#undef NC_CASE
    return 0;
}
}
