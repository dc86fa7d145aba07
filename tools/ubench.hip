// ubench.hip - gfx950 micro-benchmarks that decide the NTT kernel design (MEASUREMENT TOOL, not product):
//   1. issue rate of the integer instructions a 60-bit modular multiply is made of
//   2. register-only butterfly throughput of the two arithmetic policies (the ALU roofline of the NTT)
//   3. HBM copy bandwidth with the NTT's access pattern (8 B/lane reads, 16 B/lane row-per-lane writes)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench tools/ubench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../deeppowers_amd/csrc/modarith.h"
#include "../deeppowers_amd/csrc/tables.h"

using namespace dpfhe;

#define HIPCHECK(x)                                                                  \
    do {                                                                             \
        hipError_t e = (x);                                                          \
        if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// 1. instruction issue rates.  8 independent chains per wave, ITER iterations, asm so nothing is folded.
// ---------------------------------------------------------------------------------------------------
enum Op { ADD_U32, MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MUL_U32_U24, MAD_U32_U24, MUL_HI_U32_U24, LSHL_ADD_U64, ADDC_PAIR, ALIGNBIT, MOV_B32,
          MOV_B64, FMA_F64, MAD_U32_U16, DOT4_U32_U8, CNDMASK, CMP_GE_U64, AND_B32, ADD3_U32, FMA_F32, PK_FMA_F32, MAD_I64_I32, LSHR_B64, LSHL_B64, PK_MOV_B32, BFI_B32, PERM_B32, LSHL_OR_B32, NOT_B32, SUB_U32, NOPS };
static const char* kOpName[] = {"v_add_u32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_hi_u32_u24",
                                "v_lshl_add_u64", "v_add_co+v_addc (pair)", "v_alignbit_b32", "v_mov_b32", "v_mov_b64", "v_fma_f64", "v_mad_u32_u16",
                                "v_dot4_u32_u8", "v_cndmask_b32", "v_cmp_ge_u64", "v_and_b32", "v_add3_u32", "v_fma_f32", "v_pk_fma_f32", "v_mad_i64_i32", "v_lshrrev_b64", "v_lshlrev_b64", "v_pk_mov_b32", "v_bfi_b32", "v_perm_b32", "v_lshl_or_b32", "v_not_b32", "v_sub_u32"};

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long* out, unsigned long long* cycles, int iters, unsigned seed) {
    unsigned long long a[8];
    unsigned b = seed + threadIdx.x, c = seed * 3 + 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (unsigned long long)(threadIdx.x + i) * 0x9E3779B97F4A7C15ull + seed;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned lo = (unsigned)a[i], hi = (unsigned)(a[i] >> 32);
            if (OP == ADD_U32) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(hi) : "v"(c)); }
            if (OP == MAD_U64_U32) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(c), "v"(b) : "vcc"); continue; }
            if (OP == MAD_I64_I32) { asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(c), "v"(b) : "vcc"); continue; }
            if (OP == MUL_LO_U32) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(hi) : "v"(c)); }
            if (OP == MUL_HI_U32) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(hi) : "v"(c)); }
            if (OP == MUL_U32_U24) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(hi) : "v"(c)); }
            if (OP == MAD_U32_U24) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(c)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(hi) : "v"(c), "v"(b)); }
            if (OP == MUL_HI_U32_U24) { asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(lo) : "v"(b)); asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(hi) : "v"(c)); }
            if (OP == LSHL_ADD_U64) { unsigned long long k = ((unsigned long long)b << 32) | c; asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(k)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(k)); continue; }
            if (OP == ADDC_PAIR) { asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(b), "v"(c) : "vcc"); asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(c), "v"(b) : "vcc"); }
            if (OP == ALIGNBIT) { asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(lo) : "v"(hi)); asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(hi) : "v"(b)); }
            if (OP == MOV_B32) { asm volatile("v_mov_b32 %0, %1" : "+v"(lo) : "v"(hi)); asm volatile("v_mov_b32 %0, %1" : "+v"(hi) : "v"(b)); }
            if (OP == MOV_B64) { unsigned long long k = ((unsigned long long)b << 32) | c; asm volatile("v_mov_b64 %0, %1" : "+v"(a[i]) : "v"(k)); asm volatile("v_mov_b64 %0, %1" : "+v"(a[i]) : "v"(k)); continue; }
            if (OP == FMA_F64) { double k = 1.000001; asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k)); continue; }
            if (OP == MAD_U32_U16) { asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(c)); asm volatile("v_mad_u32_u16 %0, %0, %1, %2" : "+v"(hi) : "v"(c), "v"(b)); }
            if (OP == DOT4_U32_U8) { asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(c)); asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(hi) : "v"(c), "v"(b)); }
            if (OP == CNDMASK) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(b) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(hi) : "v"(c) : "vcc"); }
            if (OP == CMP_GE_U64) { unsigned long long k = ((unsigned long long)b << 32) | c; asm volatile("v_cmp_ge_u64 vcc, %0, %1" : : "v"(a[i]), "v"(k) : "vcc"); asm volatile("v_cmp_ge_u64 vcc, %1, %0" : : "v"(a[i]), "v"(k) : "vcc"); continue; }
            if (OP == AND_B32) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(hi) : "v"(c)); }
            if (OP == ADD3_U32) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(c)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(hi) : "v"(c), "v"(b)); }
            if (OP == FMA_F32) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(lo) : "v"(1.0001f)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(hi) : "v"(1.0001f)); }
            if (OP == PK_FMA_F32) { unsigned long long k = 0x3f8000013f800001ull; asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k)); continue; }
            if (OP == LSHR_B64) { asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(a[i])); asm volatile("v_lshrrev_b64 %0, 32, %0" : "+v"(a[i])); continue; }
            if (OP == LSHL_B64) { asm volatile("v_lshlrev_b64 %0, 4, %0" : "+v"(a[i])); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i])); continue; }
            if (OP == PK_MOV_B32) { unsigned long long k = ((unsigned long long)b << 32) | c; asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(a[i]) : "v"(k)); asm volatile("v_pk_mov_b32 %0, %1, %0 op_sel:[0,1]" : "+v"(a[i]) : "v"(k)); continue; }
            if (OP == BFI_B32) { asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(c)); asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(hi) : "v"(c), "v"(b)); }
            if (OP == PERM_B32) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(c)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(hi) : "v"(c), "v"(b)); }
            if (OP == LSHL_OR_B32) { asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(lo) : "v"(b)); asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(hi) : "v"(c)); }
            if (OP == NOT_B32) { asm volatile("v_not_b32 %0, %0" : "+v"(lo)); asm volatile("v_not_b32 %0, %0" : "+v"(hi)); }
            if (OP == SUB_U32) { asm volatile("v_sub_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); asm volatile("v_sub_u32 %0, %0, %1" : "+v"(hi) : "v"(c)); }
            a[i] = ((unsigned long long)hi << 32) | lo;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run_rate(int n_cu, double clk_ghz_nominal) {
    const int iters = 2000, blocks = n_cu * 8;  // 8 blocks x 4 waves = 8 waves per SIMD
    unsigned long long *d_out, *d_cyc;
    HIPCHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 8));
    HIPCHECK(hipMalloc(&d_cyc, (size_t)blocks * 8));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 10, 1u);
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters, 7u);
    HIPCHECK(hipEventRecord(e1));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc(blocks);
    HIPCHECK(hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto c : cyc) avg += (double)c;
    avg /= blocks;
    // per wave: iters * 8 chains * 2 instructions (ADDC pair counts as 2 pairs = 4 instrs)
    const double instr_per_wave = (double)iters * 16;
    const double waves_per_simd = 8.0;
    // wall-time view: instructions issued per SIMD / time
    const double inst_per_simd = instr_per_wave * waves_per_simd;
    const double ns_per_inst = ms * 1e6 / inst_per_simd;
    // avg = readcyclecounter ticks one wave spent in the loop while 8 waves shared its SIMD
    std::printf("%-24s  %8.3f ms   %6.2f ns/wave-inst/SIMD   = %5.2f cyc @%.1fGHz   (counter: %.2f ticks/inst/SIMD)\n", kOpName[OP], ms, ns_per_inst,
                ns_per_inst * clk_ghz_nominal, clk_ghz_nominal, avg / inst_per_simd);
    HIPCHECK(hipFree(d_out)); HIPCHECK(hipFree(d_cyc));
}

// ---------------------------------------------------------------------------------------------------
// 2. register-only butterfly throughput: radix-16 network (32 butterflies) on 16 words, repeated
// ---------------------------------------------------------------------------------------------------
// the twiddle multiply FoldArith used before the split-twiddle form (kept as the comparison point): mul60 on a plain
// 60-bit twiddle, 7 v_mad_u64_u32 + 3 v_mov + 3 v_alignbit + 2 v_and per product
struct Mul60Arith {
    struct Tw { u64 w; };
    static constexpr bool kFold = true;
    static DPF_HD u64 mul_tw(u64 y, const Tw& t, const LimbConst& c) { return FoldArith::mul60(y, t.w, (u32)c.d); }
};

template <class Arith>
__global__ __launch_bounds__(256) void bfly_kernel(u64* out, const typename Arith::Tw* tw, LimbConst lc, int iters) {
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = (u64)(threadIdx.x * 16 + k) * 0x9E3779B97F4A7C15ull % lc.q;
    typename Arith::Tw w[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) w[k] = tw[(threadIdx.x & 63) * 15 + k];
    const u64 two_q = 2 * lc.q;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int lb = 3 - u;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k & (1 << lb)) continue;
                u64 a = x[k];
                if (Arith::kFold) { if (u == 0) a = FoldArith::reduce(a, lc); } else a = csub(a, two_q);
                const u64 t = Arith::mul_tw(x[k | (1 << lb)], w[(1 << u) - 1 + (k >> (lb + 1))], lc);
                x[k] = a + t;
                x[k | (1 << lb)] = a - t + two_q;
            }
        }
    }
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class Arith>
static void run_bfly(const char* name, int n_cu, int blocks_per_cu) {
    const int iters = 500, blocks = n_cu * blocks_per_cu;
    u64* d_out; typename Arith::Tw* d_tw;
    HIPCHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 8));
    std::vector<typename Arith::Tw> tw(64 * 15);
    LimbConst lc{};
    lc.q = 1152921504606830593ull; lc.d = (1ull << 60) - lc.q;
    for (size_t i = 0; i < tw.size(); ++i) {
        u64 w = (i + 12345) * 0x9E3779B97F4A7C15ull % lc.q;
        std::memset(&tw[i], 0, sizeof(tw[i]));
        tw[i].w = w;
        if (!Arith::kFold) reinterpret_cast<u64*>(&tw[i])[1] = (u64)(((unsigned __int128)w << 64) / lc.q);
        if (sizeof(tw[i]) == 16 && Arith::kFold) {
            const TwFold t = h_tw_fold(w, lc.q);
            std::memcpy(&tw[i], &t, 16);
        }
    }
    HIPCHECK(hipMalloc(&d_tw, tw.size() * sizeof(tw[0])));
    HIPCHECK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(tw[0]), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bfly_kernel<Arith>, dim3(blocks), dim3(256), 0, 0, d_out, d_tw, lc, 5);
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(bfly_kernel<Arith>, dim3(blocks), dim3(256), 0, 0, d_out, d_tw, lc, iters);
    HIPCHECK(hipEventRecord(e1));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bflies = (double)blocks * 256 * iters * 32;
    const double gb_s_equiv = bflies / (ms * 1e-3) * (65536.0 / 24576.0) / 1e9;  // NTT N=4096: 65536 B per 24576 butterflies
    std::printf("butterflies %-6s %d blk/CU: %8.3f ms  %.3f T bfly/s  -> NTT(N=4096) ALU roofline %.0f GB/s = %.1f%% of 8 TB/s\n", name, blocks_per_cu, ms,
                bflies / (ms * 1e-3) / 1e12, gb_s_equiv, gb_s_equiv / 80.0);
    HIPCHECK(hipFree(d_out)); HIPCHECK(hipFree(d_tw));
}

// ---------------------------------------------------------------------------------------------------
// 3. HBM copy with the NTT's access pattern: one 256-thread block per 32 KiB "polynomial"
// ---------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) V2 { u64 a, b; };
template <int MODE>  // 0: 8B column loads + 16B row-per-lane stores (128 B lane stride); 1: 16B coalesced both ways;
                     // 2: 8B column loads + 16B coalesced stores; 3: 16B coalesced loads + 8B column stores
__global__ __launch_bounds__(256) void copy_kernel(u64* __restrict__ out, const u64* __restrict__ in) {
    const size_t base = (size_t)blockIdx.x * 4096;
    const int tid = threadIdx.x;
    if (MODE == 0) {
        u64 x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = in[base + k * 256 + tid];
        V2* p = reinterpret_cast<V2*>(out + base + tid * 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] = V2{x[2 * k], x[2 * k + 1]};
    } else if (MODE == 2) {
        u64 x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = in[base + k * 256 + tid];
        V2* po = reinterpret_cast<V2*>(out + base);
#pragma unroll
        for (int k = 0; k < 8; ++k) po[k * 256 + tid] = V2{x[2 * k], x[2 * k + 1]};
    } else if (MODE == 3) {
        const V2* pi = reinterpret_cast<const V2*>(in + base);
        V2 x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = pi[k * 256 + tid];
#pragma unroll
        for (int k = 0; k < 8; ++k) { out[base + (2 * k) * 256 + tid] = x[k].a; out[base + (2 * k + 1) * 256 + tid] = x[k].b; }
    } else {
        const V2* pi = reinterpret_cast<const V2*>(in + base);
        V2* po = reinterpret_cast<V2*>(out + base);
        V2 x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = pi[k * 256 + tid];
#pragma unroll
        for (int k = 0; k < 8; ++k) po[k * 256 + tid] = x[k];
    }
}

template <int MODE>
static void run_copy(const char* name) {
    const size_t polys = 4096 * 4;  // 512 MiB in + 512 MiB out (beyond the 256 MiB Infinity Cache)
    u64 *d_in, *d_out;
    HIPCHECK(hipMalloc(&d_in, polys * 32768)); HIPCHECK(hipMalloc(&d_out, polys * 32768));
    HIPCHECK(hipMemset(d_in, 1, polys * 32768));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(copy_kernel<MODE>, dim3(polys), dim3(256), 0, 0, d_out, d_in);
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy_kernel<MODE>, dim3(polys), dim3(256), 0, 0, d_out, d_in);
    HIPCHECK(hipEventRecord(e1));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("copy %-34s %8.3f ms/iter  %.0f GB/s (read+write)\n", name, ms / 10, 2.0 * polys * 32768 / (ms / 10 * 1e-3) / 1e9);
    HIPCHECK(hipFree(d_in)); HIPCHECK(hipFree(d_out));
}

static int selftest_mul_tw() {  // host check of FoldArith::mul_tw against __int128 (any 64-bit y)
    LimbConst lc{};
    lc.q = 1152921504606830593ull; lc.d = (1ull << 60) - lc.q;
    u64 s = 12345; int bad = 0;
    auto next = [&]() { s += 0x9E3779B97F4A7C15ull; u64 z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    for (int i = 0; i < 2000000; ++i) {
        u64 y = next(), w = next() % lc.q;
        if (i % 7 == 0) y = ~0ull - (i & 15);
        if (i % 11 == 0) w = lc.q - 1 - (i & 3);
        const u64 r = FoldArith::mul_tw(y, h_tw_fold(w, lc.q), lc);
        const u64 ref = (u64)(((unsigned __int128)y * w) % lc.q);
        if (r % lc.q != ref || r >= (1ull << 60) + 16 * lc.d) ++bad;
    }
    std::printf("FoldArith::mul_tw host selftest: %d mismatches / 2000000\n", bad);
    return bad;
}

int main(int argc, char** argv) {
    if (selftest_mul_tw()) return 1;
    if (argc > 1 && !std::strcmp(argv[1], "copy")) {
        run_copy<0>("8B col loads + 16B row stores"); run_copy<1>("16B coalesced"); run_copy<2>("8B col loads + 16B coalesced stores"); run_copy<3>("16B coalesced loads + 8B col stores");
        return 0;
    }
    if (argc > 1 && !std::strcmp(argv[1], "bfly")) {
        hipDeviceProp_t pp;
        HIPCHECK(hipGetDeviceProperties(&pp, 0));
        const int n = pp.multiProcessorCount;
        run_bfly<FoldArith>("fold", n, 4); run_bfly<FoldArith>("fold", n, 8); run_bfly<Mul60Arith>("mul60", n, 4); run_bfly<Mul60Arith>("mul60", n, 8);
        return 0;
    }
    hipDeviceProp_t p;
    HIPCHECK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate / 1e6;
    std::printf("device: %s  CUs=%d  clock=%.2f GHz  LDS/block=%zu  L2=%d MiB\n", p.name, p.multiProcessorCount, ghz, p.sharedMemPerBlock, p.l2CacheSize >> 20);
    const int n_cu = p.multiProcessorCount;
    std::printf("--- instruction issue (8 waves/SIMD, independent chains) ---\n");
    run_rate<ADD_U32>(n_cu, ghz); run_rate<FMA_F32>(n_cu, ghz); run_rate<PK_FMA_F32>(n_cu, ghz); run_rate<MAD_U64_U32>(n_cu, ghz); run_rate<MAD_I64_I32>(n_cu, ghz);
    run_rate<MUL_LO_U32>(n_cu, ghz); run_rate<MUL_HI_U32>(n_cu, ghz); run_rate<MUL_U32_U24>(n_cu, ghz); run_rate<MAD_U32_U24>(n_cu, ghz); run_rate<MUL_HI_U32_U24>(n_cu, ghz);
    run_rate<MAD_U32_U16>(n_cu, ghz); run_rate<DOT4_U32_U8>(n_cu, ghz); run_rate<LSHL_ADD_U64>(n_cu, ghz); run_rate<ADDC_PAIR>(n_cu, ghz); run_rate<ALIGNBIT>(n_cu, ghz);
    run_rate<MOV_B32>(n_cu, ghz); run_rate<MOV_B64>(n_cu, ghz); run_rate<AND_B32>(n_cu, ghz); run_rate<ADD3_U32>(n_cu, ghz); run_rate<CNDMASK>(n_cu, ghz);
    run_rate<CMP_GE_U64>(n_cu, ghz); run_rate<FMA_F64>(n_cu, ghz);
    run_rate<LSHR_B64>(n_cu, ghz); run_rate<LSHL_B64>(n_cu, ghz); run_rate<PK_MOV_B32>(n_cu, ghz); run_rate<BFI_B32>(n_cu, ghz); run_rate<PERM_B32>(n_cu, ghz);
    run_rate<LSHL_OR_B32>(n_cu, ghz); run_rate<NOT_B32>(n_cu, ghz); run_rate<SUB_U32>(n_cu, ghz);
    std::printf("--- register-only butterflies ---\n");
    run_bfly<FoldArith>("fold", n_cu, 4); run_bfly<FoldArith>("fold", n_cu, 8); run_bfly<Mul60Arith>("mul60", n_cu, 4); run_bfly<Mul60Arith>("mul60", n_cu, 8); run_bfly<ShoupArith>("shoup", n_cu, 4); run_bfly<ShoupArith>("shoup", n_cu, 8);
    std::printf("--- HBM copy ---\n");
    run_copy<0>("8B col loads + 16B row stores"); run_copy<1>("16B coalesced"); run_copy<2>("8B col loads + 16B coalesced stores"); run_copy<3>("16B coalesced loads + 8B col stores");
    return 0;
}
