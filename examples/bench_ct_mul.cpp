// bench_ct_mul.cpp - ciphertext-mul/s through the C++ operator API only (no Python, no PyTorch): host code in C++,
// HIP kernels behind the C ABI.  Mirrors the measurement loop of the reference's own harness
// (/root/reference/examples/performance_benchmark.cpp:15-22: warm-up, timed iterations, throughput), with HIP events.
//
//   g++ -O2 -std=c++17 -Iinclude -I/opt/rocm/include examples/bench_ct_mul.cpp -o bench_ct_mul \
//       -Ldeeppowers_amd -ldpfhe_api -ldpfhe_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/deeppowers_amd -Wl,-rpath,/opt/rocm/lib
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "deeppowers/fhe.hpp"

using namespace deeppowers::fhe;

int main(int argc, char** argv) {
    const size_t batch = argc > 1 ? (size_t)std::atol(argv[1]) : 2048;
    const int steps = argc > 2 ? std::atoi(argv[2]) : 10;
    try {
        const FheParams p = FheParams::n4096_l4();
        Context ctx(p, 0);
        Evaluator ev(ctx);
        const size_t n = p.n(), L = p.n_limbs();
        std::vector<uint64_t> host(batch * 2 * L * n);
        uint64_t s = 1;
        for (size_t i = 0; i < host.size(); ++i) {   // splitmix64 words reduced mod q_limb (SURVEY.md Appendix B generator)
            s += 0x9E3779B97F4A7C15ull;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            host[i] = (z ^ (z >> 31)) % p.moduli[(i / n) % L];
        }
        Ciphertext a(ctx, 2, batch), b(ctx, 2, batch), c(ctx, 3, batch);
        a.copy_from_host(host.data());
        b.copy_from_host(host.data());
        for (int i = 0; i < 3; ++i) ev.multiply(a, b, c);
        ctx.synchronize();
        auto hip = [](hipError_t e, const char* what) { if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); std::exit(3); } };
        hipEvent_t e0, e1;
        hip(hipEventCreate(&e0), "hipEventCreate"); hip(hipEventCreate(&e1), "hipEventCreate");
        hip(hipEventRecord(e0, nullptr), "hipEventRecord");
        for (int i = 0; i < steps; ++i) ev.multiply(a, b, c);
        hip(hipEventRecord(e1, nullptr), "hipEventRecord");
        hip(hipEventSynchronize(e1), "hipEventSynchronize");
        float ms = 0;
        hip(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
        const double per_s = (double)batch * steps / (ms * 1e-3);
        std::printf("{\"metric\": \"ciphertext-mul/s (N=4096, 4 RNS limbs)\", \"host\": \"c++\", \"batch\": %zu, \"steps\": %d, "
                    "\"ms_per_step\": %.3f, \"value\": %.1f, \"algorithmic_GBps\": %.1f}\n",
                    batch, steps, ms / steps, per_s, per_s * 917504.0 / 1e9);
        return 0;
    } catch (const Exception& e) {
        std::fprintf(stderr, "deeppowers::fhe error %d: %s\n", (int)e.code(), e.what());
        return 2;
    }
}
