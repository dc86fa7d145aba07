// ubench2.hip - gfx950 VALU issue rates and the register-only butterfly ceiling IN REAL SHADER CYCLES (MEASUREMENT TOOL, not product).
//
// Round-1's tools/ubench divided wall time by an ASSUMED 2.4 GHz.  Here every wave stamps s_memtime (shader cycles) and
// s_memrealtime (constant 100 MHz) around its loop and records its hardware placement (HW_ID: SE / CU / SIMD, XCC_ID), so that
//   * the shader clock while the loop runs is measured in the same kernel:  clock = d(memtime) / d(memrealtime) * 100 MHz;
//   * cycles per wave-instruction per SIMD = d(memtime) / (instructions per wave x waves resident on THAT SIMD), with the
//     residency counted from the recorded placements (not assumed from the grid size), and the overlap of all waves in
//     real time checked;
//   * the sweep over 1 / 2 / 4 / 8 waves per SIMD separates issue cost from dependency latency.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench2 tools/ubench2.hip
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <dirent.h>
#include <fstream>
#include <string>
#include <thread>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "../deeppowers_amd/csrc/modarith.h"
#include "../deeppowers_amd/csrc/tables.h"

using namespace dpfhe;

#define HIPCHECK(x)                                                                                                             \
    do {                                                                                                                        \
        hipError_t e = (x);                                                                                                     \
        if (e != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); std::exit(1); } \
    } while (0)

struct Stamp {
    unsigned long long t0, t1, r0, r1;
    unsigned hw_id, xcc_id, pad0, pad1;
};

__device__ __forceinline__ void stamp_begin(Stamp& s) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    s.hw_id = hw; s.xcc_id = xcc;
    s.r0 = wall_clock64();
    s.t0 = clock64();
}
__device__ __forceinline__ void stamp_end(Stamp& s) {
    s.t1 = clock64();
    s.r1 = wall_clock64();
}

enum Op { MOV_B32, AND_B32, ADD_U32, LSHR_B32, XOR_B32, FMA_F32, FMA_F32_3SRC, PK_FMA_F32, FMA_F64, MAD_U64_U32, MAD_U64_U32_S, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24,
          LSHL_ADD_U64, LSHL_ADD_U64_S, SUBB_PAIR, ADD3_U32, ALIGNBIT, MOV_B64, LSHR_B64, NUM_OPS };
static const char* kOpName[] = {"v_mov_b32", "v_and_b32", "v_add_u32", "v_lshrrev_b32", "v_xor_b32", "v_fma_f32 (d,d,k,k)", "v_fma_f32 (d,d,k1,k2)", "v_pk_fma_f32",
                                "v_fma_f64", "v_mad_u64_u32 (vgpr)", "v_mad_u64_u32 (sgpr x vgpr)", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_lshl_add_u64 (v+v)",
                                "v_lshl_add_u64 (v+s)", "v_sub_co+v_subb (pair=2)", "v_add3_u32", "v_alignbit_b32", "v_mov_b64", "v_lshrrev_b64"};

// 8 independent chains x 2 instructions per iteration = 16 instructions per iteration per wave
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long* out, Stamp* stamps, int iters, unsigned seed) {
    unsigned long long a[8];
    unsigned b = seed + threadIdx.x, c = seed * 3 + 1;
    float k1 = 1.0001f, k2 = 0.9999f;
    unsigned sb, sc;
    asm volatile("s_mov_b32 %0, 0x12345" : "=s"(sb));
    asm volatile("s_mov_b32 %0, 0x6789b" : "=s"(sc));
    unsigned long long s64 = ((unsigned long long)sb << 32) | sc;
    asm volatile("" : "+s"(s64));
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (unsigned long long)(threadIdx.x + i) * 0x9E3779B97F4A7C15ull + seed;
    Stamp st;
    stamp_begin(st);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned lo = (unsigned)a[i], hi = (unsigned)(a[i] >> 32);
            if (OP == MOV_B32) { asm volatile("v_mov_b32 %0, %1" : "+v"(lo) : "v"(hi)); asm volatile("v_mov_b32 %0, %1" : "+v"(hi) : "v"(b)); }
            if (OP == AND_B32) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo) : "v"(b)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(hi) : "v"(c)); }
            if (OP == ADD_U32) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(hi) : "v"(c)); }
            if (OP == LSHR_B32) { asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(lo)); asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(hi)); }
            if (OP == XOR_B32) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(lo) : "v"(b)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(hi) : "v"(c)); }
            if (OP == FMA_F32) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(lo) : "v"(k1)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(hi) : "v"(k1)); }
            if (OP == FMA_F32_3SRC) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(lo) : "v"(k1), "v"(k2)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(hi) : "v"(k2), "v"(k1)); }
            if (OP == PK_FMA_F32) { unsigned long long k = 0x3f8000013f800001ull; asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k)); continue; }
            if (OP == FMA_F64) { double k = 1.000001; asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k)); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(k)); continue; }
            if (OP == MAD_U64_U32) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(c), "v"(b) : "vcc"); continue; }
            if (OP == MAD_U64_U32_S) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "s"(sb), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "s"(sc), "v"(b) : "vcc"); continue; }
            if (OP == MUL_LO_U32) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(hi) : "v"(c)); }
            if (OP == MUL_HI_U32) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(b)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(hi) : "v"(c)); }
            if (OP == MAD_U32_U24) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(c)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(hi) : "v"(c), "v"(b)); }
            if (OP == LSHL_ADD_U64) { unsigned long long k = ((unsigned long long)b << 32) | c; asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(k)); asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(k)); continue; }
            if (OP == LSHL_ADD_U64_S) { asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "s"(s64)); asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(a[i]) : "s"(s64)); continue; }
            if (OP == SUBB_PAIR) { asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n\ts_nop 1\n\tv_subb_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(b), "v"(c) : "vcc"); }
            if (OP == ADD3_U32) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(b), "v"(c)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(hi) : "v"(c), "v"(b)); }
            if (OP == ALIGNBIT) { asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(lo) : "v"(hi)); asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(hi) : "v"(b)); }
            if (OP == MOV_B64) { unsigned long long k = ((unsigned long long)b << 32) | c; asm volatile("v_mov_b64 %0, %1" : "+v"(a[i]) : "v"(k)); asm volatile("v_mov_b64 %0, %1" : "+v"(a[i]) : "v"(k)); continue; }
            if (OP == LSHR_B64) { asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(a[i])); asm volatile("v_lshrrev_b64 %0, 32, %0" : "+v"(a[i])); continue; }
            a[i] = ((unsigned long long)hi << 32) | lo;
        }
    }
    stamp_end(st);
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;
}

// ---------------------------------------------------------------------------------------------------
// register-only butterflies: a radix-16 Cooley-Tukey network (32 butterflies) on 16 words, repeated.
//   VARIANT 0: round-1 butterfly  t = w y; x' = a + t; y' = a - t + 2q            (13 VALU)
//   VARIANT 1: fused sum          x' = reduce(a + w y) in the product's own chain; y' = 2a + 2q - x'   (12 VALU)
// ---------------------------------------------------------------------------------------------------
template <int VARIANT>
__global__ __launch_bounds__(256) void bfly_kernel(u64* out, Stamp* stamps, const TwFold* tw, LimbConst lc, int iters) {
    u64 x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = (u64)(threadIdx.x * 16 + k) * 0x9E3779B97F4A7C15ull % lc.q;
    TwFold w[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) w[k] = tw[(threadIdx.x & 63) * 15 + k];
    const u64 two_q = 2 * lc.q;
    Stamp st;
    stamp_begin(st);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int lb = 3 - u;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k & (1 << lb)) continue;
                u64 a = x[k];
                if (u == 0) a = FoldArith::reduce(a, lc);   // one reduction per 4 stages keeps every variant inside its bounds
                if (VARIANT == 0) {
                    const u64 t = FoldArith::mul_tw(x[k | (1 << lb)], w[(1 << u) - 1 + (k >> (lb + 1))], lc);
                    x[k] = a + t;
                    x[k | (1 << lb)] = a - t + two_q;
                } else {
                    // fused needs a < 4 * 2^60 and y' = 2a + 2q - x' doubles a's bound: x' -> 1, y' -> 4 or 10 (units of 2^60);
                    // only the words that reach 10 are reduced before serving as `a` again (same rule as ntt_core.h make_ctf_plan)
                    if ((u == 2 && (k == 12 || k == 13)) || (u == 3 && k == 6)) a = FoldArith::reduce(a, lc);
                    const u64 s = FoldArith::mul_tw_add(x[k | (1 << lb)], w[(1 << u) - 1 + (k >> (lb + 1))], lc, a);
                    x[k] = s;
                    x[k | (1 << lb)] = shl1_add(a, two_q) - s;
                }
            }
        }
    }
    stamp_end(st);
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s ^= x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = st;
}

struct Summary {
    double clock_mhz, cyc_per_inst, ns_per_inst, wave_cyc_per_inst, resident_avg, resident_max, ms;
    int simds_used;
};

static unsigned long long simd_key(const Stamp& s) {
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...; XCC_ID[3:0]
    return ((unsigned long long)(s.xcc_id & 0xf) << 32) | (s.hw_id & 0xff30u);
}

// instr_per_wave: VALU instructions one wave executes between its stamps.
// Per SIMD:  cycles per instruction = (last end - first start of its waves, in shader cycles) / (instructions all its waves issued);
// this does not assume how many waves were resident together (the dispatcher may run a SIMD's waves in several rounds).
// The number that WERE resident together is measured from the real-time stamps and reported beside it.
static Summary summarise(const std::vector<Stamp>& st, double instr_per_wave, float ms) {
    std::map<unsigned long long, std::vector<const Stamp*>> per_simd;
    for (const Stamp& s : st) per_simd[simd_key(s)].push_back(&s);
    double cyc = 0, clk = 0, wave_cyc = 0, res_sum = 0, res_max = 0;
    for (const Stamp& s : st) {
        clk += (double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 100.0;   // s_memrealtime: 100 MHz
        wave_cyc += (double)(s.t1 - s.t0) / instr_per_wave;
    }
    for (auto& kv : per_simd) {
        unsigned long long t0 = ~0ull, t1 = 0;
        std::vector<std::pair<unsigned long long, int>> ev;
        for (const Stamp* s : kv.second) {
            t0 = std::min(t0, s->t0); t1 = std::max(t1, s->t1);
            ev.push_back({s->r0, +1}); ev.push_back({s->r1, -1});
        }
        std::sort(ev.begin(), ev.end());
        int cur = 0, mx = 0;
        for (auto& e : ev) { cur += e.second; mx = std::max(mx, cur); }
        cyc += (double)(t1 - t0) / (instr_per_wave * kv.second.size());
        res_sum += mx; res_max = std::max(res_max, (double)mx);
    }
    Summary r;
    r.cyc_per_inst = cyc / per_simd.size();
    r.clock_mhz = clk / st.size();
    r.wave_cyc_per_inst = wave_cyc / st.size();
    r.resident_avg = res_sum / per_simd.size();
    r.resident_max = res_max;
    r.simds_used = (int)per_simd.size();
    r.ms = ms;
    r.ns_per_inst = r.cyc_per_inst / r.clock_mhz * 1e3;
    return r;
}

template <int OP>
static void run_rate(int n_cu, int blocks_per_cu, int iters) {
    const int blocks = n_cu * blocks_per_cu, waves = blocks * 4;
    unsigned long long* d_out; Stamp* d_st;
    HIPCHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 8));
    HIPCHECK(hipMalloc(&d_st, (size_t)waves * sizeof(Stamp)));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters / 4, 1u);   // warm the clocks
    HIPCHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters, 7u);
    HIPCHECK(hipEventRecord(e1));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> st(waves);
    HIPCHECK(hipMemcpy(st.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
    const Summary s = summarise(st, (double)iters * 16, ms);
    std::printf("%-28s %d blk/CU: resident %.2f avg %.0f max waves/SIMD on %4d SIMDs  clock %6.0f MHz  %6.2f cyc/inst/SIMD  %5.2f ns  (one wave sees %6.2f cyc between its own instructions)  [%.3f ms]\n",
                kOpName[OP], blocks_per_cu, s.resident_avg, s.resident_max, s.simds_used, s.clock_mhz, s.cyc_per_inst, s.ns_per_inst, s.wave_cyc_per_inst, ms);
    HIPCHECK(hipFree(d_out)); HIPCHECK(hipFree(d_st));
}

template <int OP>
static void sweep_rate(int n_cu) {
    run_rate<OP>(n_cu, 1, 20000); run_rate<OP>(n_cu, 2, 10000); run_rate<OP>(n_cu, 4, 5000); run_rate<OP>(n_cu, 8, 2500);
}

template <int VARIANT>
static void run_bfly(const char* name, int n_cu, int blocks_per_cu, int iters, double instr_per_bfly) {
    const int blocks = n_cu * blocks_per_cu, waves = blocks * 4;
    u64* d_out; Stamp* d_st; TwFold* d_tw;
    HIPCHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 8));
    HIPCHECK(hipMalloc(&d_st, (size_t)waves * sizeof(Stamp)));
    std::vector<TwFold> tw(64 * 15);
    LimbConst lc{};
    lc.q = 1152921504606830593ull; lc.d = (1ull << 60) - lc.q;
    for (size_t i = 0; i < tw.size(); ++i) tw[i] = h_tw_fold((i + 12345) * 0x9E3779B97F4A7C15ull % lc.q, lc.q);
    HIPCHECK(hipMalloc(&d_tw, tw.size() * sizeof(tw[0])));
    HIPCHECK(hipMemcpy(d_tw, tw.data(), tw.size() * sizeof(tw[0]), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bfly_kernel<VARIANT>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, d_tw, lc, iters / 4);
    HIPCHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(bfly_kernel<VARIANT>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, d_tw, lc, iters);
    HIPCHECK(hipEventRecord(e1));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> st(waves);
    HIPCHECK(hipMemcpy(st.data(), d_st, waves * sizeof(Stamp), hipMemcpyDeviceToHost));
    const Summary s = summarise(st, (double)iters * 32, ms);   // "instruction" = one butterfly here
    const double bfly_per_s = (double)blocks * 256 * iters * 32 / (ms * 1e-3);
    const double ntt_equiv = bfly_per_s * (65536.0 / 24576.0);  // N = 4096: 65536 algorithmic bytes per 24576 butterflies
    std::printf("butterflies %-10s %d blk/CU: resident %.2f avg waves/SIMD  clock %6.0f MHz  %6.2f cyc/bfly/SIMD = %.2f cyc per VALU (%.2f VALU/bfly)  "
                "%.3f T bfly/s = NTT(4096) %.0f GB/s = %.1f%% of 8 TB/s;  scaled to 2400 MHz: %.3f T bfly/s   [%.3f ms]\n",
                name, blocks_per_cu, s.resident_avg, s.clock_mhz, s.cyc_per_inst, s.cyc_per_inst / instr_per_bfly, instr_per_bfly, bfly_per_s / 1e12,
                ntt_equiv / 1e9, ntt_equiv / 80e9, bfly_per_s / 1e12 * 2400.0 / s.clock_mhz, ms);
    HIPCHECK(hipFree(d_out)); HIPCHECK(hipFree(d_st)); HIPCHECK(hipFree(d_tw));
}

// ---- energy per instruction: each rate kernel back to back for ~1.2 s while a host thread samples the board power (hwmon) ----
static std::string hwmon_dir_of_device0() {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, 0) != hipSuccess) return "";
    for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
    DIR* d = opendir("/sys/class/drm");
    if (!d) return "";
    std::string found;
    while (dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name.rfind("card", 0) != 0 || name.find('-') != std::string::npos) continue;
        char real[512];
        const std::string dev = "/sys/class/drm/" + name + "/device";
        if (!realpath(dev.c_str(), real)) continue;
        const std::string r = real;
        if (r.size() < std::strlen(bus) || r.compare(r.size() - std::strlen(bus), std::strlen(bus), bus) != 0) continue;
        DIR* h = opendir((dev + "/hwmon").c_str());
        if (!h) continue;
        while (dirent* he = readdir(h))
            if (std::string(he->d_name).rfind("hwmon", 0) == 0) found = dev + "/hwmon/" + he->d_name;
        closedir(h);
    }
    closedir(d);
    return found;
}
static double read_num(const std::string& path) {
    std::ifstream f(path);
    double v = -1;
    f >> v;
    return v;
}
struct PowerStat { double watts, mhz; int n; };
template <class Launch>
static PowerStat sample_while(const std::string& hw, double seconds, Launch&& launch_batch) {
    std::atomic<bool> stop{false};
    double sum_w = 0, sum_f = 0;
    int n = 0;
    std::thread t([&] {
        std::this_thread::sleep_for(std::chrono::milliseconds(300));   // let the averaged sensor settle on this load
        while (!stop.load()) {
            const double w = read_num(hw + "/power1_input"), f = read_num(hw + "/freq1_input");
            if (w > 0) { sum_w += w / 1e6; sum_f += f / 1e6; ++n; }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    });
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        launch_batch();
        HIPCHECK(hipDeviceSynchronize());
    }
    stop.store(true);
    t.join();
    return PowerStat{n ? sum_w / n : 0, n ? sum_f / n : 0, n};
}
template <int OP>
static void power_rate(const std::string& hw, int n_cu, double idle_w) {
    const int blocks = n_cu * 4, iters = 40000;
    unsigned long long* d_out; Stamp* d_st;
    HIPCHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 8));
    HIPCHECK(hipMalloc(&d_st, (size_t)blocks * 4 * sizeof(Stamp)));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    double ms_sum = 0; int launches = 0;
    const PowerStat ps = sample_while(hw, 1.3, [&] {
        HIPCHECK(hipEventRecord(e0));
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters, 7u);
        HIPCHECK(hipEventRecord(e1));
        HIPCHECK(hipEventSynchronize(e1));
        float ms = 0; HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        ms_sum += ms; launches += 8;
    });
    const double wave_inst = (double)blocks * 4 * iters * 16 * launches;   // 16 instructions per loop iteration and wave
    const double rate = wave_inst / (ms_sum * 1e-3);
    std::printf("%-28s board %7.1f W  sclk %5.0f MHz  %7.1f G wave-inst/s  -> %5.2f nJ per wave-instruction above idle (%d samples)\n", kOpName[OP], ps.watts, ps.mhz,
                rate / 1e9, (ps.watts - idle_w) / rate * 1e9, ps.n);
    HIPCHECK(hipFree(d_out)); HIPCHECK(hipFree(d_st));
}
static void power_mode(int n_cu) {
    const std::string hw = hwmon_dir_of_device0();
    if (hw.empty()) { std::printf("no hwmon directory for device 0\n"); return; }
    double idle = 0; int n = 0;
    for (int i = 0; i < 25; ++i) { const double w = read_num(hw + "/power1_input"); if (w > 0) { idle += w / 1e6; ++n; } usleep(20000); }
    idle = n ? idle / n : 0;
    std::printf("--- energy per VALU instruction (4 workgroups per CU, 1.3 s per op, hwmon %s), idle %.1f W ---\n", hw.c_str(), idle);
    power_rate<MOV_B32>(hw, n_cu, idle); power_rate<AND_B32>(hw, n_cu, idle); power_rate<ADD_U32>(hw, n_cu, idle); power_rate<LSHR_B32>(hw, n_cu, idle);
    power_rate<FMA_F32_3SRC>(hw, n_cu, idle); power_rate<FMA_F64>(hw, n_cu, idle); power_rate<MAD_U64_U32>(hw, n_cu, idle); power_rate<MAD_U64_U32_S>(hw, n_cu, idle);
    power_rate<MUL_LO_U32>(hw, n_cu, idle); power_rate<MUL_HI_U32>(hw, n_cu, idle); power_rate<MAD_U32_U24>(hw, n_cu, idle); power_rate<LSHL_ADD_U64>(hw, n_cu, idle);
    power_rate<SUBB_PAIR>(hw, n_cu, idle); power_rate<ADD3_U32>(hw, n_cu, idle); power_rate<ALIGNBIT>(hw, n_cu, idle); power_rate<MOV_B64>(hw, n_cu, idle);
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    HIPCHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    int wall_khz = 0;
    HIPCHECK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    std::printf("device: %s  CUs=%d  nominal clock=%.2f GHz  wall clock rate %d kHz\n", p.name, n_cu, p.clockRate / 1e6, wall_khz);
    if (argc > 1 && !std::strcmp(argv[1], "power")) { power_mode(n_cu); return 0; }
    const bool only_bfly = argc > 1 && !std::strcmp(argv[1], "bfly");
    if (!only_bfly) {
        std::printf("--- VALU issue, 8 independent chains per wave, in-kernel clocks ---\n");
        sweep_rate<MOV_B32>(n_cu); sweep_rate<AND_B32>(n_cu); sweep_rate<ADD_U32>(n_cu); sweep_rate<LSHR_B32>(n_cu); sweep_rate<XOR_B32>(n_cu);
        sweep_rate<FMA_F32>(n_cu); sweep_rate<FMA_F32_3SRC>(n_cu); sweep_rate<PK_FMA_F32>(n_cu); sweep_rate<FMA_F64>(n_cu);
        sweep_rate<MAD_U64_U32>(n_cu); sweep_rate<MAD_U64_U32_S>(n_cu); sweep_rate<MUL_LO_U32>(n_cu); sweep_rate<MUL_HI_U32>(n_cu); sweep_rate<MAD_U32_U24>(n_cu);
        sweep_rate<LSHL_ADD_U64>(n_cu); sweep_rate<LSHL_ADD_U64_S>(n_cu); sweep_rate<SUBB_PAIR>(n_cu); sweep_rate<ADD3_U32>(n_cu); sweep_rate<ALIGNBIT>(n_cu);
        sweep_rate<MOV_B64>(n_cu); sweep_rate<LSHR_B64>(n_cu);
    }
    std::printf("--- register-only butterflies (FoldArith), in-kernel clocks ---\n");
    for (int b : {1, 2, 4, 8}) run_bfly<0>("plain13", n_cu, b, 4000 / b, 13.0 + 3.0 * 8 / 32);
    for (int b : {1, 2, 4, 8}) run_bfly<1>("fused12", n_cu, b, 4000 / b, 12.0 + 3.0 * 11 / 32);
    return 0;
}
