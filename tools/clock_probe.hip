// clock_probe.hip - shader clock while a library kernel runs (MEASUREMENT TOOL): a one-wave probe kernel on a second stream samples
// clock64() (shader cycles) and wall_clock64() (constant 100 MHz) while ct_mul / NTT launches occupy the chip.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include -o tools/clock_probe tools/clock_probe.hip -Ldeeppowers_amd -ldpfhe_hip -Wl,-rpath,$PWD/deeppowers_amd
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dpfhe.h"

#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

__global__ void probe_kernel(unsigned long long* out, int samples, int spin) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < samples; ++i) {
        out[2 * i] = clock64();
        out[2 * i + 1] = wall_clock64();
        for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(64);
    }
}

static double probe_mhz(hipStream_t s, unsigned long long* d, int samples, int spin, int wall_khz) {
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, s, d, samples, spin);
    HIPCHECK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(2 * samples);
    HIPCHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    const double dc = (double)(h[2 * (samples - 1)] - h[2]), dw = (double)(h[2 * (samples - 1) + 1] - h[3]);
    return dc / dw * wall_khz / 1e3;
}

int main() {
    const uint64_t mod[4] = {1152921504606830593ull, 1152921504606748673ull, 1152921504606683137ull, 1152921504606601217ull};
    const uint64_t psi[4] = {116777451583545ull, 271802498405390ull, 134367042585739ull, 276147373136904ull};
    dpfhe_ctx* ctx;
    if (dpfhe_ctx_create(&ctx, 12, 4, mod, psi, 0)) { std::printf("ctx: %s\n", dpfhe_last_error()); return 1; }
    int wall_khz = 100000;
    HIPCHECK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    const size_t B = 8192, N = 4096, L = 4;
    uint64_t *a, *b, *c, *x, *y;
    HIPCHECK(hipMalloc(&a, B * 2 * L * N * 8)); HIPCHECK(hipMalloc(&b, B * 2 * L * N * 8)); HIPCHECK(hipMalloc(&c, B * 3 * L * N * 8));
    HIPCHECK(hipMalloc(&x, 8192 * L * N * 8)); HIPCHECK(hipMalloc(&y, 8192 * L * N * 8));
    HIPCHECK(hipMemset(a, 1, B * 2 * L * N * 8)); HIPCHECK(hipMemset(b, 2, B * 2 * L * N * 8)); HIPCHECK(hipMemset(x, 3, 8192 * L * N * 8));
    unsigned long long* d;
    HIPCHECK(hipMalloc(&d, 2 * 4096 * 8));
    hipStream_t s1, s2;
    HIPCHECK(hipStreamCreate(&s1)); HIPCHECK(hipStreamCreate(&s2));
    std::printf("wall clock rate %d kHz\n", wall_khz);
    std::printf("idle chip (probe only):                         %7.0f MHz\n", probe_mhz(s2, d, 200, 40, wall_khz));
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 12; ++i) dpfhe_ct_mul(ctx, c, a, b, B, 0, s1);          // ~45 ms of multiplies
        hipEvent_t go; HIPCHECK(hipEventCreate(&go));
        std::printf("during ct_mul (N=4096, L=4, 8192 pairs/launch):  %7.0f MHz\n", probe_mhz(s2, d, 400, 40, wall_khz));
        HIPCHECK(hipStreamSynchronize(s1));
        for (int i = 0; i < 60; ++i) dpfhe_ntt_fwd_oop(ctx, y, x, 8192, s1);          // ~35 ms of forward NTTs
        std::printf("during forward NTT (8192 RNS polys/launch):      %7.0f MHz\n", probe_mhz(s2, d, 400, 40, wall_khz));
        HIPCHECK(hipStreamSynchronize(s1));
    }
    dpfhe_ctx_destroy(ctx);
    return 0;
}
