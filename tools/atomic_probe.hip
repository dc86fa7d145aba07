// Tool (not product): what does it cost to fold a shard-local lazy reduction into the producing kernel with no-return 64-bit
// atomic adds?  Models ct_mul's store pattern at BASELINE configs[3]'s shard: `pairs` x L workgroups of 256 threads, each adding
// 3 x 4096 words into accumulator slot (pair / group) - `group` <= 16 canonical 60-bit values fit a u64 without reduction.
// Compared with a plain streaming read of the same words (what the separate reduce kernel pays).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_probe tools/atomic_probe.hip && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int N = 4096, L = 4, T = 256, E = 16;

template <int MODE>   // 0: atomics agent scope, 1: plain load+add+store (slot private to the workgroup sequence - NOT race free, timing only), 2: read only
__global__ __launch_bounds__(T) void probe(u64* __restrict__ acc, const u64* __restrict__ src, int group, u64* sink) {
    const size_t bi = blockIdx.x / L;
    const int limb = blockIdx.x % L;
    const int tid = threadIdx.x;
    u64 s = 0;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const u64* p = src + ((bi * 3 + c) * L + limb) * N;
        u64* a = acc + (((bi / group) * 3 + c) * L + limb) * N;
        u64 x[E];
#pragma unroll
        for (int k = 0; k < E; k += 2) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(p + (k / 2) * (2 * T) + 2 * tid);
            x[k] = v.x; x[k + 1] = v.y;
        }
#pragma unroll
        for (int k = 0; k < E; ++k) {
            u64* q = a + (k / 2) * (2 * T) + 2 * tid + (k & 1);
            if (MODE == 0) __hip_atomic_fetch_add(q, x[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 1) *q += x[k];
            else s += x[k];
        }
    }
    if (MODE == 2 && s == 0x1234567) *sink = s;
}

// MODE 3 of the question: one workgroup walks `group` consecutive pairs of its limb and keeps a PRIVATE lazy accumulator in global
// memory (first pair: plain store; later pairs: load + add + store) - no atomics, no races; the accumulator lines live in L2 /
// the Infinity Cache between visits.
__global__ __launch_bounds__(T) void probe_private(u64* __restrict__ acc, const u64* __restrict__ src, int group, u64* sink) {
    const size_t g = blockIdx.x / L;
    const int limb = blockIdx.x % L;
    const int tid = threadIdx.x;
#pragma unroll 1
    for (int i = 0; i < group; ++i) {
        const size_t bi = g * group + i;
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            const u64* p = src + ((bi * 3 + c) * L + limb) * N;
            u64* a = acc + ((g * 3 + c) * L + limb) * N;
            ulonglong2 x[E / 2], y[E / 2];
#pragma unroll
            for (int k = 0; k < E / 2; ++k) x[k] = *reinterpret_cast<const ulonglong2*>(p + k * (2 * T) + 2 * tid);
            if (i > 0) {
#pragma unroll
                for (int k = 0; k < E / 2; ++k) y[k] = *reinterpret_cast<const ulonglong2*>(a + k * (2 * T) + 2 * tid);
#pragma unroll
                for (int k = 0; k < E / 2; ++k) { x[k].x += y[k].x; x[k].y += y[k].y; }
            }
#pragma unroll
            for (int k = 0; k < E / 2; ++k) *reinterpret_cast<ulonglong2*>(a + k * (2 * T) + 2 * tid) = x[k];
        }
    }
}

int main() {
    const int pairs = 8192;
    const size_t words = (size_t)pairs * 3 * L * N;
    u64 *src, *acc, *sink;
    CK(hipMalloc(&src, words * 8));
    CK(hipMalloc(&acc, words * 8));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(src, 1, words * 8));
    CK(hipMemset(acc, 0, words * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, int group) {
        std::vector<float> ms;
        for (int i = 0; i < 7; ++i) {
            CK(hipEventRecord(e0));
            kern<<<pairs * L, T>>>(acc, src, group, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-34s group %5d: median %8.1f us  min %8.1f us   (%.2f TB/s of source reads)\n", name, group, ms[3] * 1e3, ms[0] * 1e3, words * 8 / (ms[3] * 1e-3) / 1e12);
    };
    for (int group : {1, 16, 64, 8192}) run("atomic add (agent scope, no return)", probe<0>, group);
    for (int group : {1, 16}) run("load + add + store", probe<1>, group);
    run("read only", probe<2>, 1);
    for (int group : {4, 8, 16}) {
        std::vector<float> ms;
        for (int i = 0; i < 7; ++i) {
            CK(hipEventRecord(e0));
            probe_private<<<pairs / group * L, T>>>(acc, src, group, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-34s group %5d: median %8.1f us  min %8.1f us   (%.2f TB/s of source reads)\n", "private accumulator (RMW, no race)", group, ms[3] * 1e3, ms[0] * 1e3, words * 8 / (ms[3] * 1e-3) / 1e12);
    }
    return 0;
}
