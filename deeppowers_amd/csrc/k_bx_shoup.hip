#include "kernels_bx.h"
namespace dpfhe {
template int launch_base_extend<ShoupArith>(int, u64*, size_t, const u64*, size_t, size_t, const BaseExtArgs&, const LimbConst*, int, int, unsigned, hipStream_t);
}
