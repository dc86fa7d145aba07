// dpfhe_cabi.hip - libdpfhe_hip.so: the C ABI declared in include/dpfhe.h over the gfx950 kernels.
//
// Reference seam this stands in for: none exists.  Conventions follow the
// reference's HAL: device chosen by id (/root/reference/src/api/cpp/src/deeppowers.cpp:15), raw device
// pointers (/root/reference/src/core/hal/hal.hpp:44-48), optional stream last (hal.hpp:95), errors as
// deeppowers::common::ErrorCode numbers (/root/reference/src/common/error.hpp:10-40).
// Unlike /root/reference/src/core/distributed/distributed_context.cpp:88,109 nothing here allocates,
// creates streams or synchronises per call.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/dpfhe.h"
#include "base_ext.h"
#include "kernels_large.h"
#include "kernels_misc.h"
#include "launch.h"
#include "ntt_core.h"
#include "ntt_quarters.h"
#include "tables.h"

using namespace dpfhe;

static thread_local std::string g_last_error;

static int fail(int code, const char* what, const char* detail) {
    g_last_error = std::string(what) + ": " + (detail ? detail : "");
    return code;
}
#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return fail(_e == hipErrorOutOfMemory ? DPFHE_OUT_OF_MEMORY : DPFHE_DEVICE_ERROR, #expr, hipGetErrorString(_e)); \
    } while (0)

// Every compute entry point runs on the CONTEXT's device (tables and, by contract, the caller's buffers live there), whatever
// the calling thread's current device is: the guard switches to it for the duration of the call and restores the caller's
// device afterwards.  Same device: two cheap runtime queries, no switch.
struct DeviceGuard {
    int prev = -1, want = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) : want(device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != want) err = hipSetDevice(want);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define DPFHE_ON_DEVICE(ctx, what)                                                                            \
    DeviceGuard _guard((ctx)->device);                                                                         \
    if (_guard.err != hipSuccess) return fail(DPFHE_DEVICE_ERROR, what, hipGetErrorString(_guard.err))

// half-open word ranges [a, a + na) and [b, b + nb) intersect
static inline bool overlaps(const uint64_t* a, size_t na, const uint64_t* b, size_t nb) {
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(a), b0 = reinterpret_cast<uintptr_t>(b);
    return a0 < b0 + nb * 8 && b0 < a0 + na * 8;
}

static inline bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }
static const size_t kMaxGrid = 0x7fffffff;

struct dpfhe_ctx {
    uint32_t log2n = 0, n_limbs = 0;
    int device = 0;
    bool fold = false;
    int n_cu = 256;          // compute units of the device (launch-size caps of the streaming kernels)
    std::vector<uint64_t> moduli;   // host copy (constants of dpfhe_base_extend / dpfhe_scale_round)
    uint64_t p_special = 0;  // the LAST modulus (the special prime of hybrid key switching when this is an extended context)
    void* d_blob = nullptr;  // one allocation: LimbConst[L] | fwd | inv | last | RescaleConst[L]  (both arithmetic layouts share it)
    const RescaleConst* d_rescale = nullptr;
    DevTables<ShoupArith> shoup{};
    DevTables<FoldArith> foldt{};
    // Round 6 - per-limb arithmetic classes.  A context whose limbs are not ALL of the pinned 2^60 - d shape used to run every limb on the
    // generic (Harvey / Shoup) kernels.  Now each limb gets the fastest policy its prime admits (tables.h limb_class) for the batched transforms and
    // the fused multiply: one launch per class present, each over that class's limbs only (devtables.h DevTables::n_active / active_map).  The
    // generic tables above stay complete (every limb), so the key-switching kernels of such a context run as before.
    // `classes` is set when L <= 16, 8 <= log2 N <= 14 and at least one limb has a faster class than the context-wide policy.
    bool classes = false;
    int uniform_cls = kClassShoup;                    // the class ALL limbs share (kClassShoup when they differ, or when `classes` is off): which policy's KEY-SWITCHING
                                                      // kernels the context runs (with_policy below); a mixture keeps the generic ones, on the complete generic tables
    unsigned char limb_cls[16] = {};                  // LimbClass of limb i
    unsigned long long cls_map = 0;                   // the same, 4 bits per limb
    void* class_blob = nullptr;                       // ONE blob of tables whose per-limb slots are in their limb's class format (mixed_layout)
    DevTables<FoldArith> cls_fold{};                  // typed views of it; the active-limb maps are set per launch (for_each_class)
    DevTables<F64Arith> cls_f64{};
    DevTables<FoldScaledArith> cls_fscaled{};
    DevTables<F64WideArith> cls_f64w{};
    DevTables<ShoupArith> cls_shoup{};
    MixedTables mixed{};                              // the batched transforms' one-launch view (kernels.h ntt_classes_kernel)
    // scratch of the composed large-ring operations: a pool of this context's own (created on first use) that keeps what it has been
    // given until the context goes - the default pool hands its memory back at every synchronisation and pays the mapping again
    struct ScratchArena { hipStream_t stream; u64* p; size_t words; uint64_t last_use; };
    std::vector<ScratchArena> scratch_arenas;   // one per stream that ran a composed operation (StreamScratch below); at most kMaxScratchArenas, least recently used evicted
    uint64_t scratch_clock = 0;                 // guarded by scratch_mutex
    std::mutex scratch_mutex;
    // which form of the fused multiply dpfhe_ct_mul(flags = 0) launches (launch.h CtMulVariant): the default of the ring degree until
    // dpfhe_ctx_autotune or dpfhe_ctx_set_ct_mul_variant says otherwise.  Both forms give the same words.
    std::atomic<int> ct_mul_variant{0};
    dpfhe_tune_info tune{};           // guarded by tune_mutex (the launch path reads ct_mul_variant only)
    mutable std::mutex tune_mutex;
    std::atomic<size_t> scratch_limit_words{(size_t)1024 << 17};   // slice size of the composed large-ring operations (dpfhe_ctx_set_scratch_limit; read on the compute path)
};


// ------------------------------------------------------------------------------------------------
// Variant choice of the fused multiply (the reference sketches this class of mechanism as an auto-tuner:
// /root/reference/src/core/inference/auto_tuner.hpp:26-64).  The two forms move the same bytes and give the same words.
// dpfhe_ctx_create NEVER measures: it takes the ring degree's default form (or what an earlier EXPLICIT dpfhe_ctx_autotune on the same
// (device, log2 N, L) found - a process-wide cache), allocates nothing besides the tables and launches nothing (round 5: the constructor-time
// probe of round 4 cost 256 MiB and 30 launches per context to confirm a default that won on every box measured).
// dpfhe_ctx_autotune, on CALLER scratch: `reps` back-to-back launches per form over `pairs` synthetic ciphertext pairs, two passes in opposite
// orders, best pass per form; a non-default form is taken only when it is at least 3 % faster than the default.
// ------------------------------------------------------------------------------------------------
static const char* const kCtMulVariantNames[kCtMulVariants] = {"quad", "dual"};
static const float kTuneMargin = 0.97f;

__global__ __launch_bounds__(256) void tune_fill_kernel(u64* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        p[i] = ((i + 1) * 0x9E3779B97F4A7C15ull) >> 5;   // < 2^59: a canonical residue of every FoldArith prime
}

static bool ct_mul_variant_compiled(const dpfhe_ctx* c, int v) {
    return c->fold && (c->log2n == 12 || c->log2n == 13) && v >= 0 && v < kCtMulVariants;
}

// process-wide results of explicit probes, keyed (device, log2 N, L): a later context of the same shape starts from them
struct TuneKey {
    int device; uint32_t log2n, n_limbs;
    bool operator==(const TuneKey& o) const { return device == o.device && log2n == o.log2n && n_limbs == o.n_limbs; }
};
static std::mutex g_tune_cache_mutex;
static std::vector<std::pair<TuneKey, dpfhe_tune_info>> g_tune_cache;

// probes on `s`; synchronises it.  us[v] < 0: not available.  Returns a hipError_t.
static hipError_t probe_ct_mul(dpfhe_ctx* c, u64* work, size_t pairs, unsigned reps, hipStream_t s, float us[kCtMulVariants]) {
    const size_t poly = (size_t)c->n_limbs << c->log2n, blocks = pairs * c->n_limbs;
    u64 *a = work, *b = work + 2 * pairs * poly, *o = work + 4 * pairs * poly;
    hipLaunchKernelGGL(tune_fill_kernel, dim3(c->n_cu * 8), dim3(256), 0, s, work, 4 * pairs * poly);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    for (int v = 0; v < kCtMulVariants; ++v) us[v] = -1.0f;
    for (int pass = 0; pass < 2 && err == hipSuccess; ++pass) {
        for (int i = 0; i < kCtMulVariants && err == hipSuccess; ++i) {
            const int v = pass ? kCtMulVariants - 1 - i : i;
            if (!ct_mul_variant_compiled(c, v)) continue;
            if (launch_ct_mul_variant<FoldArith>((int)c->log2n, v, o, a, b, blocks, c->foldt, s)) continue;    // warm-up (code object, clocks)
            (void)hipEventRecord(e0, s);
            for (unsigned r = 0; r < reps; ++r) (void)launch_ct_mul_variant<FoldArith>((int)c->log2n, v, o, a, b, blocks, c->foldt, s);
            (void)hipEventRecord(e1, s);
            err = hipEventSynchronize(e1);
            float ms = 0;
            if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
            if (err == hipSuccess) { const float t = ms * 1e3f / reps; if (us[v] < 0 || t < us[v]) us[v] = t; }
        }
    }
    if (err == hipSuccess) err = hipGetLastError();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return err;
}

static void adopt_probe(dpfhe_ctx* c, const float us[kCtMulVariants], size_t pairs, unsigned reps) {
    const int def = ct_mul_default_variant((int)c->log2n);
    int best = def;
    for (int v = 0; v < kCtMulVariants; ++v)
        if (us[v] > 0 && us[def] > 0 && us[v] < kTuneMargin * us[def] && (best == def || us[v] < us[best])) best = v;
    dpfhe_tune_info t{};
    t.n_variants = kCtMulVariants;
    t.source = DPFHE_TUNE_EXPLICIT;
    t.probe_pairs = (uint32_t)pairs;
    t.probe_reps = reps;
    for (int v = 0; v < 8; ++v) t.probe_us[v] = v < kCtMulVariants ? us[v] : -1.0f;
    t.chosen = best;
    {
        std::lock_guard<std::mutex> lk(c->tune_mutex);
        c->tune = t;
        c->ct_mul_variant.store(best);
    }
    const TuneKey key{c->device, c->log2n, c->n_limbs};
    std::lock_guard<std::mutex> lk(g_tune_cache_mutex);
    for (auto& e : g_tune_cache)
        if (e.first == key) { e.second = t; return; }
    g_tune_cache.emplace_back(key, t);
}

// at context creation: the default form, or the cached result of an explicit probe of this shape on this device.  No device work.
static void tune_at_create(dpfhe_ctx* c) {
    const int def = ct_mul_default_variant((int)c->log2n);
    dpfhe_tune_info t{};
    t.chosen = def;
    t.source = DPFHE_TUNE_DEFAULT;
    t.n_variants = ct_mul_variant_compiled(c, def) ? kCtMulVariants : 0;
    for (int v = 0; v < 8; ++v) t.probe_us[v] = -1.0f;
    if (t.n_variants) {
        const TuneKey key{c->device, c->log2n, c->n_limbs};
        std::lock_guard<std::mutex> lk(g_tune_cache_mutex);
        for (const auto& e : g_tune_cache)
            if (e.first == key) { t = e.second; t.source = DPFHE_TUNE_CACHED; break; }
    }
    c->tune = t;
    c->ct_mul_variant.store(t.chosen);
}

extern "C" void dpfhe_tune_cache_clear(void) {
    std::lock_guard<std::mutex> lk(g_tune_cache_mutex);
    g_tune_cache.clear();
}

extern "C" int dpfhe_ctx_autotune(dpfhe_ctx* c, uint64_t* d_work, size_t work_words, uint32_t reps, void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_autotune", "null context");
    if (!ct_mul_variant_compiled(c, ct_mul_default_variant((int)c->log2n))) return DPFHE_SUCCESS;   // one form only: nothing to choose
    const size_t poly = (size_t)c->n_limbs << c->log2n, pairs = work_words / (7 * poly);
    if (!d_work || misaligned(d_work) || pairs == 0) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_autotune", "d_work must hold at least 7 L N words (one ciphertext pair + its product)");
    if (pairs * c->n_limbs > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_autotune", "work buffer too large for one launch");
    if (reps == 0) reps = 3;
    DPFHE_ON_DEVICE(c, "dpfhe_ctx_autotune");
    float us[kCtMulVariants];
    HIP_TRY(probe_ct_mul(c, d_work, pairs, reps, static_cast<hipStream_t>(stream), us));
    adopt_probe(c, us, pairs, reps);
    return DPFHE_SUCCESS;
}
extern "C" int dpfhe_ctx_tune_info(const dpfhe_ctx* c, dpfhe_tune_info* out) {
    if (!c || !out) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_tune_info", "null argument");
    std::lock_guard<std::mutex> lk(c->tune_mutex);
    *out = c->tune;
    out->chosen = c->ct_mul_variant.load();
    return DPFHE_SUCCESS;
}
extern "C" int dpfhe_ctx_set_ct_mul_variant(dpfhe_ctx* c, int variant) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_set_ct_mul_variant", "null context");
    if (!ct_mul_variant_compiled(c, variant)) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_set_ct_mul_variant", "this context has no such form of the fused multiply");
    std::lock_guard<std::mutex> lk(c->tune_mutex);
    c->ct_mul_variant.store(variant);
    c->tune.chosen = variant;
    c->tune.source = DPFHE_TUNE_FORCED;
    return DPFHE_SUCCESS;
}
extern "C" const char* dpfhe_ct_mul_variant_name(int variant) {
    return variant >= 0 && variant < kCtMulVariants ? kCtMulVariantNames[variant] : "";
}

// ------------------------------------------------------------------------------------------------
// Tables of a context with per-limb arithmetic classes (round 6): ONE blob, LimbConst[L] | fwd4 | inv4 | (fwd | inv when the batched transforms use another
// layout) | last[L] | last2[L], every per-limb slot in the format of that limb's class (all twiddle types are 16 bytes: the strides agree).  The classes'
// DevTables are typed views of the same blob, each with its own active-limb map.  Single-kernel transforms only (log2 N <= 14).
static_assert(sizeof(TwShoup) == 16 && sizeof(TwFold) == 16 && sizeof(TwF64) == 16, "the per-limb slots of the mixed tables share one stride");
struct MixedLayout { size_t o_lc, o_fwd4, o_inv4, o_fwd, o_inv, o_last, o_last2, total; };
static MixedLayout mixed_layout(int log2n, size_t L) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t n = (size_t)1 << log2n, tab = L * n * 16;
    const bool two_geo = ntt_loge(log2n) != kFusedLoge;
    MixedLayout m;
    m.o_lc = 0; m.o_fwd4 = up(L * sizeof(LimbConst)); m.o_inv4 = up(m.o_fwd4 + tab);
    m.o_fwd = two_geo ? up(m.o_inv4 + tab) : m.o_fwd4; m.o_inv = two_geo ? up(m.o_fwd + tab) : m.o_inv4;
    m.o_last = up(m.o_inv + tab); m.o_last2 = up(m.o_last + L * 32); m.total = up(m.o_last2 + L * 32);
    return m;
}
template <class Tw, class MakeTw>
static void fill_mixed_limb(std::vector<unsigned char>& blob, const MixedLayout& m, int log2n, size_t l, const HostLimbTables& t, LimbClass cls, MakeTw make_tw) {
    static_assert(sizeof(Tw) == 16 && sizeof(InvLast<Tw>) == 32, "slot sizes");
    const size_t n = (size_t)1 << log2n;
    const u64 q = t.lc.q;
    const LimbConst lc = limb_const_of_class(t.lc, cls);
    std::memcpy(&blob[m.o_lc + l * sizeof(LimbConst)], &lc, sizeof(LimbConst));
    auto pack = [&](const std::vector<u64>& words, int loge, size_t off) {
        std::vector<Tw> tw(words.size());
        for (size_t i = 0; i < words.size(); ++i) tw[i] = make_tw(words[i], q);
        permute_window0(tw, log2n, loge, geo_perm_stages(log2n, loge));
        std::memcpy(&blob[off], tw.data(), tw.size() * sizeof(Tw));
    };
    pack(t.rp, kFusedLoge, m.o_fwd4 + l * n * 16);
    pack(t.irp, kFusedLoge, m.o_inv4 + l * n * 16);
    if (m.o_fwd != m.o_fwd4) { pack(t.rp, ntt_loge(log2n), m.o_fwd + l * n * 16); pack(t.irp, ntt_loge(log2n), m.o_inv + l * n * 16); }
    reinterpret_cast<InvLast<Tw>*>(&blob[m.o_last])[l] = InvLast<Tw>{make_tw(t.w_last, q), make_tw(t.lc.ninv, q)};
    // products of two scaled words carry s = 2^(60-k) twice: their inverse transform ends on twiddles with s^-1 folded in (DevTables::last2)
    const u64 sinv = cls == kClassFoldScaled ? h_powmod((1ull << fold_scaled_shift(q)) % q, q - 2, q) : 1;
    reinterpret_cast<InvLast<Tw>*>(&blob[m.o_last2])[l] = InvLast<Tw>{make_tw(h_mulmod(t.w_last, sinv, q), q), make_tw(h_mulmod(t.lc.ninv, sinv, q), q)};
}
template <class Arith>
static DevTables<Arith> mixed_view(const unsigned char* b, const MixedLayout& m, size_t L) {
    typedef typename Arith::Tw Tw;
    DevTables<Arith> tb{};
    tb.lc = reinterpret_cast<const LimbConst*>(b + m.o_lc);
    tb.fwd = reinterpret_cast<const Tw*>(b + m.o_fwd); tb.inv = reinterpret_cast<const Tw*>(b + m.o_inv);
    tb.fwd4 = reinterpret_cast<const Tw*>(b + m.o_fwd4); tb.inv4 = reinterpret_cast<const Tw*>(b + m.o_inv4);
    tb.last = reinterpret_cast<const InvLast<Tw>*>(b + m.o_last);
    tb.last2 = reinterpret_cast<const InvLast<Tw>*>(b + m.o_last2);
    tb.n_sub = 1;
    tb.n_limbs = (int)L;
    return tb;
}

extern "C" int dpfhe_ctx_create(dpfhe_ctx** out, uint32_t log2_n, uint32_t n_limbs, const uint64_t* moduli,
                                const uint64_t* psi, int device_id) {
    if (!out || !moduli || !psi) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_create", "null argument");
    if (log2_n < 8 || log2_n > (uint32_t)kMaxLog2N) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_create", "log2_n must be in [8, 16]");
    if (n_limbs == 0 || n_limbs > 1024) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_create", "n_limbs must be in [1, 1024]");
    const size_t n = (size_t)1 << log2_n, L = n_limbs;
    std::vector<HostLimbTables> ht(L);
    bool fold = true;
    for (size_t l = 0; l < L; ++l) {
        // inverses (N^-1, psi^-1, rescale constants) are Fermat powers: a composite modulus that happens to satisfy
        // psi^N = -1 would make them silently wrong, so primality is checked (deterministic Miller-Rabin, one-off)
        if (!h_is_prime(moduli[l])) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_create", "modulus is not prime");
        int rc = build_limb_tables((int)log2_n, moduli[l], psi[l], ht[l]);
        if (rc) return fail(rc, "dpfhe_ctx_create", "modulus must be < 2^60 and 1 mod 2N, psi a primitive 2N-th root");
        fold = fold && fold_eligible(moduli[l]);
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_create", "no such device");
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device_id));

    dpfhe_ctx* c = new (std::nothrow) dpfhe_ctx;
    if (!c) return fail(DPFHE_OUT_OF_MEMORY, "dpfhe_ctx_create", "host allocation");
    c->log2n = log2_n; c->n_limbs = n_limbs; c->device = device_id; c->fold = fold; c->p_special = moduli[L - 1];
    c->moduli.assign(moduli, moduli + L);
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) c->n_cu = cus; }

    // blob layout (all 256-byte aligned sections).  One twiddle table pair per kernel geometry in use: slot 0 = the
    // fused kernels' LOGE 4 layout, slot 1 = the batched NTT kernels' layout when that differs.  Split transforms
    // (N > 16384) store, per limb, n_sub tables of N2 points (sub-trees of the full table) plus the top-stage twiddles.
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t tw_sz = fold ? sizeof(TwFold) : sizeof(TwShoup);
    const int log_n1 = split_log_n1((int)log2_n), log_n2 = (int)log2_n - log_n1;
    const size_t n_sub = (size_t)1 << log_n1, n2 = (size_t)1 << log_n2;
    const int loge_ntt = ntt_loge(log_n2);
    const bool split = log_n1 > 0, two_geo = !split && loge_ntt != kFusedLoge;
    const size_t tab = L * n * tw_sz;
    const size_t o_lc = 0, o_fwd4 = up(o_lc + L * sizeof(LimbConst)), o_inv4 = up(o_fwd4 + tab),
                 o_fwd = two_geo ? up(o_inv4 + tab) : o_fwd4, o_inv = two_geo ? up(o_fwd + tab) : o_inv4,
                 o_last = up(o_inv + tab), o_top_fwd = up(o_last + L * n_sub * 2 * tw_sz), o_top_inv = up(o_top_fwd + L * n_sub * tw_sz),
                 o_top_last = up(o_top_inv + L * n_sub * tw_sz), o_resc = up(o_top_last + L * 2 * tw_sz);
    // N = 8192: the "halves" tables next to the one-piece ones (ntt_halves.h; the fused kernels keep the one-piece layout)
    const bool halves = log2_n == 13 && fold;   // the halves tables (launch.h): large batched transforms at N = 8192
    const size_t o_hfwd = up(o_resc + L * sizeof(RescaleConst)), o_hinv = halves ? up(o_hfwd + tab) : o_hfwd, o_htop_fwd = halves ? up(o_hinv + tab) : o_hfwd,
                 o_htop_last = halves ? up(o_htop_fwd + L * tw_sz) : o_hfwd, o_q0 = halves ? up(o_htop_last + L * 2 * tw_sz) : o_hfwd;
    // N = 16384, FoldArith: the "quarters" tables next to the one-piece ones (ntt_quarters.h: four sub-tree tables per limb + the column stages' twiddles)
    const bool quarters = log2_n == 14 && fold;
    const size_t o_qfwd = o_q0, o_qinv = quarters ? up(o_qfwd + tab) : o_q0, o_qtop_fwd = quarters ? up(o_qinv + tab) : o_q0,
                 o_qtop_inv = quarters ? up(o_qtop_fwd + L * sizeof(QuartersTop)) : o_q0, o_qtop_last = quarters ? up(o_qtop_inv + L * 2 * tw_sz) : o_q0,
                 total = quarters ? up(o_qtop_last + L * 2 * tw_sz) : o_q0;
    std::vector<unsigned char> blob(total, 0);
    auto fill = [&](auto tw_tag) {
        typedef decltype(tw_tag) Tw;
        for (size_t l = 0; l < L; ++l) {
            const u64 q = moduli[l];
            auto pack = [&](const std::vector<u64>& words, int logn_tab, int loge, size_t off) {   // one N-point table in kernel layout
                std::vector<Tw> t(words.size());
                for (size_t i = 0; i < words.size(); ++i) t[i] = h_make_tw<Tw>(words[i], q);
                permute_window0(t, logn_tab, loge, geo_perm_stages(logn_tab, loge));
                std::memcpy(&blob[off], t.data(), t.size() * sizeof(Tw));
            };
            InvLast<Tw>* lasts = reinterpret_cast<InvLast<Tw>*>(&blob[o_last]);
            if (!split) {
                for (int geo = 0; geo < (two_geo ? 2 : 1); ++geo) {
                    const int loge = geo ? loge_ntt : kFusedLoge;
                    pack(ht[l].rp, (int)log2_n, loge, (geo ? o_fwd : o_fwd4) + l * n * tw_sz);
                    pack(ht[l].irp, (int)log2_n, loge, (geo ? o_inv : o_inv4) + l * n * tw_sz);
                }
                lasts[l] = InvLast<Tw>{h_make_tw<Tw>(ht[l].w_last, q), h_make_tw<Tw>(ht[l].lc.ninv, q)};
                if constexpr (std::is_same<Tw, TwFold>::value) {
                    if (quarters) {
                        for (size_t r = 0; r < 4; ++r) {
                            pack(subtree_table(ht[l].rp, 14, 2, r), 12, 4, o_qfwd + (l * 4 + r) * (n / 4) * tw_sz);
                            pack(subtree_table(ht[l].irp, 14, 2, r), 12, 4, o_qinv + (l * 4 + r) * (n / 4) * tw_sz);
                        }
                        reinterpret_cast<QuartersTop*>(&blob[o_qtop_fwd])[l] = QuartersTop{h_tw_fold(ht[l].rp[1], q), h_tw_fold(ht[l].rp[2], q), h_tw_fold(ht[l].rp[3], q)};
                        reinterpret_cast<TwFold*>(&blob[o_qtop_inv])[2 * l] = h_tw_fold(ht[l].irp[2], q);
                        reinterpret_cast<TwFold*>(&blob[o_qtop_inv])[2 * l + 1] = h_tw_fold(ht[l].irp[3], q);
                        reinterpret_cast<InvLast<TwFold>*>(&blob[o_qtop_last])[l] = InvLast<TwFold>{h_tw_fold(ht[l].w_last, q), h_tw_fold(ht[l].lc.ninv, q)};
                    }
                }
                if (halves) {
                    for (size_t r = 0; r < 2; ++r) {
                        pack(subtree_table(ht[l].rp, 13, 1, r), 12, 4, o_hfwd + (l * 2 + r) * (n / 2) * tw_sz);
                        pack(subtree_table(ht[l].irp, 13, 1, r), 12, 4, o_hinv + (l * 2 + r) * (n / 2) * tw_sz);
                    }
                    reinterpret_cast<Tw*>(&blob[o_htop_fwd])[l] = h_make_tw<Tw>(ht[l].rp[1], q);
                    reinterpret_cast<InvLast<Tw>*>(&blob[o_htop_last])[l] = lasts[l];   // the column stage IS the one-piece transform's last stage
                }
            } else {
                for (size_t r = 0; r < n_sub; ++r) {
                    const std::vector<u64> f = subtree_table(ht[l].rp, (int)log2_n, log_n1, r), v = subtree_table(ht[l].irp, (int)log2_n, log_n1, r);
                    pack(f, log_n2, loge_ntt, o_fwd + (l * n_sub + r) * n2 * tw_sz);
                    pack(v, log_n2, loge_ntt, o_inv + (l * n_sub + r) * n2 * tw_sz);
                    // generic primes: no N^-1 inside a block.  FoldArith: the block's last stage divides its sums by N2 exactly (FoldArith::mul_ninv),
                    // so its differences carry N2^-1 in their twiddle; the column stage then multiplies by N1^-1 (top_last below)
                    const u64 n2inv = std::is_same<Tw, TwFold>::value ? h_powmod((u64)n2 % q, q - 2, q) : 1;
                    lasts[l * n_sub + r] = InvLast<Tw>{h_make_tw<Tw>(h_mulmod(v[1], n2inv, q), q), h_make_tw<Tw>(1, q)};
                }
                Tw* tf = reinterpret_cast<Tw*>(&blob[o_top_fwd]) + l * n_sub;
                Tw* tv = reinterpret_cast<Tw*>(&blob[o_top_inv]) + l * n_sub;
                for (size_t i = 1; i < n_sub; ++i) { tf[i] = h_make_tw<Tw>(ht[l].rp[i], q); tv[i] = h_make_tw<Tw>(ht[l].irp[i], q); }
                // FoldArith sub-transforms divide by their own length N2 in their last stage (ntt_core.h: FoldArith::mul_ninv, exact division), so the
                // column stage multiplies by N1^-1 = N^-1 N2 only; generic-prime sub-transforms multiply by 1 there and the column stage by N^-1
                const u64 up = std::is_same<Tw, TwFold>::value ? (u64)n2 % q : 1;
                reinterpret_cast<InvLast<Tw>*>(&blob[o_top_last])[l] = InvLast<Tw>{h_make_tw<Tw>(h_mulmod(ht[l].w_last, up, q), q), h_make_tw<Tw>(h_mulmod(ht[l].lc.ninv, up, q), q)};
            }
        }
    };
    for (size_t l = 0; l < L; ++l) std::memcpy(&blob[o_lc + l * sizeof(LimbConst)], &ht[l].lc, sizeof(LimbConst));
    if (fold) fill(TwFold{}); else fill(TwShoup{});
    {   // rescale constants relative to the LAST prime (used only when L >= 2)
        const u64 ql = moduli[L - 1], hh = ql / 2;
        RescaleConst* r = reinterpret_cast<RescaleConst*>(&blob[o_resc]);
        for (size_t l = 0; l + 1 < L; ++l) {
            const u64 q = moduli[l];
            r[l].h_mod = hh % q; r[l].inv = h_powmod(ql % q, q - 2, q); r[l].q_last = ql; r[l].h = hh;
        }
    }
    hipError_t e = hipMalloc(&c->d_blob, total);
    if (e == hipSuccess) e = hipMemcpy(c->d_blob, blob.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (c->d_blob) (void)hipFree(c->d_blob);
        delete c;
        (void)hipSetDevice(prev);
        return fail(e == hipErrorOutOfMemory ? DPFHE_OUT_OF_MEMORY : DPFHE_DEVICE_ERROR, "dpfhe_ctx_create: table upload", hipGetErrorString(e));
    }
    unsigned char* d = static_cast<unsigned char*>(c->d_blob);
    c->d_rescale = reinterpret_cast<const RescaleConst*>(d + o_resc);
    if (fold) {
        c->foldt.lc = reinterpret_cast<const LimbConst*>(d + o_lc);
        c->foldt.fwd = reinterpret_cast<const TwFold*>(d + o_fwd);
        c->foldt.inv = reinterpret_cast<const TwFold*>(d + o_inv);
        c->foldt.fwd4 = reinterpret_cast<const TwFold*>(d + o_fwd4);
        c->foldt.inv4 = reinterpret_cast<const TwFold*>(d + o_inv4);
        c->foldt.last = reinterpret_cast<const InvLast<TwFold>*>(d + o_last);
        c->foldt.top_fwd = reinterpret_cast<const TwFold*>(d + o_top_fwd);
        c->foldt.top_inv = reinterpret_cast<const TwFold*>(d + o_top_inv);
        c->foldt.top_last = reinterpret_cast<const InvLast<TwFold>*>(d + o_top_last);
        c->foldt.n_sub = (int)n_sub;
        c->foldt.n_limbs = (int)n_limbs;
        if (quarters) {
            c->foldt.qfwd = reinterpret_cast<const TwFold*>(d + o_qfwd); c->foldt.qinv = reinterpret_cast<const TwFold*>(d + o_qinv);
            c->foldt.qtop_fwd = reinterpret_cast<const QuartersTop*>(d + o_qtop_fwd); c->foldt.qtop_inv = reinterpret_cast<const TwFold*>(d + o_qtop_inv);
            c->foldt.qtop_last = reinterpret_cast<const InvLast<TwFold>*>(d + o_qtop_last);
        }
        if (halves) {
            c->foldt.hfwd = reinterpret_cast<const TwFold*>(d + o_hfwd); c->foldt.hinv = reinterpret_cast<const TwFold*>(d + o_hinv);
            c->foldt.htop_fwd = reinterpret_cast<const TwFold*>(d + o_htop_fwd); c->foldt.htop_last = reinterpret_cast<const InvLast<TwFold>*>(d + o_htop_last);
        }
    } else {
        c->shoup.lc = reinterpret_cast<const LimbConst*>(d + o_lc);
        c->shoup.fwd = reinterpret_cast<const TwShoup*>(d + o_fwd);
        c->shoup.inv = reinterpret_cast<const TwShoup*>(d + o_inv);
        c->shoup.fwd4 = reinterpret_cast<const TwShoup*>(d + o_fwd4);
        c->shoup.inv4 = reinterpret_cast<const TwShoup*>(d + o_inv4);
        c->shoup.last = reinterpret_cast<const InvLast<TwShoup>*>(d + o_last);
        c->shoup.top_fwd = reinterpret_cast<const TwShoup*>(d + o_top_fwd);
        c->shoup.top_inv = reinterpret_cast<const TwShoup*>(d + o_top_inv);
        c->shoup.top_last = reinterpret_cast<const InvLast<TwShoup>*>(d + o_top_last);
        c->shoup.n_sub = (int)n_sub;
        c->shoup.n_limbs = (int)n_limbs;
        if (halves) {
            c->shoup.hfwd = reinterpret_cast<const TwShoup*>(d + o_hfwd); c->shoup.hinv = reinterpret_cast<const TwShoup*>(d + o_hinv);
            c->shoup.htop_fwd = reinterpret_cast<const TwShoup*>(d + o_htop_fwd); c->shoup.htop_last = reinterpret_cast<const InvLast<TwShoup>*>(d + o_htop_last);
        }
    }
    // per-limb arithmetic classes of a non-uniform context (see dpfhe_ctx::classes)
    if (!fold && L <= 16 && log2_n >= 8 && log2_n <= 14) {
        bool any_fast = false;
        for (size_t l = 0; l < L; ++l) { c->limb_cls[l] = (unsigned char)limb_class(moduli[l]); any_fast = any_fast || c->limb_cls[l] != kClassShoup; }
        if (any_fast) {
            const MixedLayout m = mixed_layout((int)log2_n, L);
            std::vector<unsigned char> mb(m.total, 0);
            for (size_t l = 0; l < L; ++l) {
                const LimbClass k = (LimbClass)c->limb_cls[l];
                c->cls_map |= (unsigned long long)k << (4 * l);
                if (k == kClassFold) fill_mixed_limb<TwFold>(mb, m, (int)log2_n, l, ht[l], k, [](u64 w, u64 q) { return h_tw_fold(w, q); });
                else if (k == kClassF64 || k == kClassF64Wide) fill_mixed_limb<TwF64>(mb, m, (int)log2_n, l, ht[l], k, [](u64 w, u64 q) { return h_make_tw<TwF64>(w, q); });
                else if (k == kClassFoldScaled) fill_mixed_limb<TwFold>(mb, m, (int)log2_n, l, ht[l], k, [](u64 w, u64 q) { return h_tw_fold_scaled(w, q, fold_scaled_shift(q)); });
                else fill_mixed_limb<TwShoup>(mb, m, (int)log2_n, l, ht[l], k, [](u64 w, u64 q) { return h_make_tw<TwShoup>(w, q); });
            }
            hipError_t ce = hipMalloc(&c->class_blob, m.total);
            if (ce == hipSuccess) ce = hipMemcpy(c->class_blob, mb.data(), m.total, hipMemcpyHostToDevice);
            if (ce != hipSuccess) {
                if (c->class_blob) (void)hipFree(c->class_blob);
                (void)hipFree(c->d_blob);
                delete c;
                (void)hipSetDevice(prev);
                return fail(ce == hipErrorOutOfMemory ? DPFHE_OUT_OF_MEMORY : DPFHE_DEVICE_ERROR, "dpfhe_ctx_create: class table upload", hipGetErrorString(ce));
            }
            const unsigned char* b = static_cast<const unsigned char*>(c->class_blob);
            c->cls_fold = mixed_view<FoldArith>(b, m, L);
            c->cls_f64 = mixed_view<F64Arith>(b, m, L);
            c->cls_fscaled = mixed_view<FoldScaledArith>(b, m, L);
            c->cls_f64w = mixed_view<F64WideArith>(b, m, L);
            c->cls_shoup = mixed_view<ShoupArith>(b, m, L);
            c->mixed.fwd = b + m.o_fwd; c->mixed.inv = b + m.o_inv; c->mixed.last = b + m.o_last;
            c->mixed.lc = reinterpret_cast<const LimbConst*>(b + m.o_lc);
            c->mixed.n_limbs = (int)L;
            c->mixed.cls_map = c->cls_map;
            c->classes = true;
            bool same = true;
            for (size_t l = 1; l < L; ++l) same = same && c->limb_cls[l] == c->limb_cls[0];
            if (same) c->uniform_cls = c->limb_cls[0];
        }
    }
    tune_at_create(c);   // default form of the fused multiply, or a cached explicit probe of this shape: no device work
    (void)hipSetDevice(prev);
    *out = c;
    return DPFHE_SUCCESS;
}

extern "C" int dpfhe_ctx_destroy(dpfhe_ctx* c) {
    if (!c) return DPFHE_SUCCESS;
    if (c->d_blob) (void)hipFree(c->d_blob);
    if (c->class_blob) (void)hipFree(c->class_blob);
    for (auto& a : c->scratch_arenas) if (a.p) (void)hipFree(a.p);
    delete c;
    return DPFHE_SUCCESS;
}
extern "C" int dpfhe_ctx_set_scratch_limit(dpfhe_ctx* c, size_t mib) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_set_scratch_limit", "null context");
    if (mib == 0 || mib > ((size_t)1 << 20)) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_set_scratch_limit", "limit must be in [1 MiB, 1 TiB]");
    c->scratch_limit_words.store(mib << 17, std::memory_order_relaxed);
    return DPFHE_SUCCESS;
}
extern "C" uint32_t dpfhe_ctx_log2n(const dpfhe_ctx* c) { return c ? c->log2n : 0; }
extern "C" uint32_t dpfhe_ctx_limbs(const dpfhe_ctx* c) { return c ? c->n_limbs : 0; }
extern "C" int dpfhe_ctx_uses_fold(const dpfhe_ctx* c) { return c && c->fold ? 1 : 0; }
extern "C" int dpfhe_ctx_limb_class(const dpfhe_ctx* c, uint32_t limb) {
    if (!c || limb >= c->n_limbs) return -1;
    if (c->classes) return (int)c->limb_cls[limb];
    return c->fold ? DPFHE_ARITH_FOLD : DPFHE_ARITH_SHOUP;
}
static_assert(DPFHE_ARITH_SHOUP == kClassShoup && DPFHE_ARITH_FOLD == kClassFold && DPFHE_ARITH_F64 == kClassF64 && DPFHE_ARITH_FOLD_SCALED == kClassFoldScaled &&
              DPFHE_ARITH_F64_WIDE == kClassF64Wide, "dpfhe.h <-> tables.h");

// ------------------------------------------------------------------------------------------------
// words per thread of the FoldArith matvec kernels: 2 right-hand-side polynomials per workgroup / 4 (kernels_misc.h matvec_fold_kernel)
constexpr int kMatvecWpt2 = 2, kMatvecWpt4 = 1;
constexpr int kMatvecWd = 1;     // columns of W lookahead in the multi-right-hand-side product (2 measured no faster: profiles/r04_ab_matvec_full.txt)
constexpr int kMatvecRt4 = 4;    // rows per workgroup of the 4-polynomial (2-token) kernel (8 measured slower)

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DPFHE_DEVICE_ERROR, what, hipGetErrorString(e));
    return DPFHE_SUCCESS;
}

template <class Arith>
static const DevTables<Arith>& tables_of(const dpfhe_ctx* c);
template <>
const DevTables<ShoupArith>& tables_of<ShoupArith>(const dpfhe_ctx* c) { return c->shoup; }
template <>
const DevTables<FoldArith>& tables_of<FoldArith>(const dpfhe_ctx* c) { return c->foldt; }

// the widest grid a batched transform of `npolys` residue polynomials launches fits one launch
static bool ntt_grid_fits(const dpfhe_ctx* c, size_t npolys) {
    // split transforms (N > 16384) launch npolys * N1 sub-transforms and npolys * N2 / 256 column workgroups
    const int log_n1 = split_log_n1((int)c->log2n);
    const size_t widest = log_n1 ? ((npolys << log_n1) > (npolys << (kSplitLog2N2 - 8)) ? (npolys << log_n1) : (npolys << (kSplitLog2N2 - 8))) : npolys;
    return npolys <= kMaxGrid && widest <= kMaxGrid;
}
// arguments validated, device selected by the caller
// One launch per arithmetic class present among the first `limbs_used` limbs of a non-uniform context (dpfhe_ctx::classes): `fn(tables)` launches
// over the class's limbs.
template <class Fn>
static int for_each_class(const dpfhe_ctx* c, size_t limbs_used, Fn fn) {
    auto restrict_to = [&](auto tb, LimbClass k) {
        int na = 0;
        unsigned long long map = 0;
        for (size_t l = 0; l < limbs_used; ++l) if (c->limb_cls[l] == (unsigned char)k) map |= (unsigned long long)l << (4 * na++);
        tb.n_limbs = (int)limbs_used;
        tb.n_active = na;
        tb.active_map = map;
        return tb;
    };
    int rc = 0;
    { auto tb = restrict_to(c->cls_fold, kClassFold); if (tb.n_active && !rc) rc = fn(tb); }
    { auto tb = restrict_to(c->cls_f64, kClassF64); if (tb.n_active && !rc) rc = fn(tb); }
    { auto tb = restrict_to(c->cls_fscaled, kClassFoldScaled); if (tb.n_active && !rc) rc = fn(tb); }
    { auto tb = restrict_to(c->cls_f64w, kClassF64Wide); if (tb.n_active && !rc) rc = fn(tb); }
    { auto tb = restrict_to(c->cls_shoup, kClassShoup); if (tb.n_active && !rc) rc = fn(tb); }
    return rc;
}

// The policy of the KEY-SWITCHING kernels (relin_kernel, hoisted_ks_kernel, ntt_inv_galois_kernel): fold for the pinned primes; the class's own for a
// context whose limbs all share one class (round 6: the generic forms of those kernels with that class's transforms and products - an all-f64 context no
// longer key-switches at the generic kernels' rate); the generic policy on the complete generic tables otherwise (a MIXTURE of classes does not come here for
// its transform-bearing kernels: relin_launch / with_policy_or_classes below launch once per class).  fn(tables) launches.
template <class Fn>
static int with_policy(const dpfhe_ctx* c, Fn fn) {
    if (c->fold) return fn(c->foldt);
    if (c->classes) {
        if (c->uniform_cls == kClassF64) return fn(c->cls_f64);
        if (c->uniform_cls == kClassF64Wide) return fn(c->cls_f64w);
        if (c->uniform_cls == kClassFoldScaled) return fn(c->cls_fscaled);
    }
    return fn(c->shoup);
}

// relin_kernel launches (`blocks` = items x L): on the context's policy, or - a MIXTURE of classes - one launch per class present, each over its own limbs
// (relin_kernel maps its workgroups through DevTables::active_map: the digits of every limb still enter every limb's transforms)
static int relin_launch(const dpfhe_ctx* c, int mode, u64* out, const u64* in, const u64* evk, size_t key_stride, unsigned key_group, size_t blocks, hipStream_t s) {
    if (c->classes && c->uniform_cls == kClassShoup) {
        const size_t items = blocks / c->n_limbs;
        return for_each_class(c, c->n_limbs, [&](const auto& tb) { return launch_relin((int)c->log2n, mode, out, in, evk, key_stride, key_group, items * (size_t)tb.n_active, tb, s); });
    }
    return with_policy(c, [&](const auto& tb) { return launch_relin((int)c->log2n, mode, out, in, evk, key_stride, key_group, blocks, tb, s); });
}

// the same choice for kernels whose workgroups map through DevTables::active_map (hoisted_ks_kernel, ntt_inv_galois_kernel): a mixture of classes launches once per class
template <class Fn>
static int with_policy_or_classes(const dpfhe_ctx* c, Fn fn) {
    if (c->classes && c->uniform_cls == kClassShoup) return for_each_class(c, c->n_limbs, fn);
    return with_policy(c, fn);
}

// batched transform of `items` RNS polynomials over the first `limbs_used` limbs (limbs_used = 0: all); arguments validated, device selected by the caller
static int ntt_launch_items(dpfhe_ctx* c, bool inverse, uint64_t* out, const uint64_t* in, size_t items, size_t limbs_used, hipStream_t s) {
    const size_t Lu = limbs_used ? limbs_used : c->n_limbs;
    int rc;
    if (c->classes) {
        // several classes among these limbs: one launch, the class branch inside the kernel.  One class only: that class's own kernel (the merged
        // kernel carries the register budget of its widest arm: measured 62 against 69 % of HBM peak on an all-f64 context)
        bool several = false;
        for (size_t l = 1; l < Lu; ++l) several = several || c->limb_cls[l] != c->limb_cls[0];
        rc = 1;
        if (several) {
            MixedTables mt = c->mixed;
            mt.n_limbs = (int)Lu;
            rc = launch_ntt_classes((int)c->log2n, inverse, out, in, items * Lu, mt, s);
        }
        if (rc == 1)                                                                   // (one class, or a geometry without the merged kernel)
            rc = for_each_class(c, Lu, [&](const auto& tb) { return launch_ntt((int)c->log2n, inverse, out, in, items * (size_t)tb.n_active, tb, s); });
    } else if (c->fold) {
        DevTables<FoldArith> td = c->foldt; td.n_limbs = (int)Lu;
        rc = launch_ntt<FoldArith>((int)c->log2n, inverse, out, in, items * Lu, td, s);
    } else {
        DevTables<ShoupArith> td = c->shoup; td.n_limbs = (int)Lu;
        rc = launch_ntt<ShoupArith>((int)c->log2n, inverse, out, in, items * Lu, td, s);
    }
    if (rc) return fail(DPFHE_INVALID_STATE, "ntt", "no kernel geometry for this log2_n");
    return check_launch("ntt kernel launch");
}
static int ntt_launch(dpfhe_ctx* c, bool inverse, uint64_t* out, const uint64_t* in, size_t npolys, hipStream_t s) {
    return ntt_launch_items(c, inverse, out, in, npolys / c->n_limbs, 0, s);
}

static int ntt_entry(dpfhe_ctx* c, bool inverse, uint64_t* out, const uint64_t* in, size_t n_rns_polys, void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "ntt", "null context");
    if (n_rns_polys == 0) return DPFHE_SUCCESS;
    if (!out || !in || misaligned(out) || misaligned(in)) return fail(DPFHE_INVALID_ARGUMENT, "ntt", "null or misaligned buffer");
    const size_t npolys = n_rns_polys * c->n_limbs;
    if (!ntt_grid_fits(c, npolys)) return fail(DPFHE_INVALID_ARGUMENT, "ntt", "batch too large for one launch");
    DPFHE_ON_DEVICE(c, "ntt");
    return ntt_launch(c, inverse, out, in, npolys, static_cast<hipStream_t>(stream));
}

extern "C" int dpfhe_ntt_fwd(dpfhe_ctx* c, uint64_t* d_io, size_t n, void* s) { return ntt_entry(c, false, d_io, d_io, n, s); }
extern "C" int dpfhe_ntt_inv(dpfhe_ctx* c, uint64_t* d_io, size_t n, void* s) { return ntt_entry(c, true, d_io, d_io, n, s); }
extern "C" int dpfhe_ntt_fwd_oop(dpfhe_ctx* c, uint64_t* o, const uint64_t* i, size_t n, void* s) { return ntt_entry(c, false, o, i, n, s); }
extern "C" int dpfhe_ntt_inv_oop(dpfhe_ctx* c, uint64_t* o, const uint64_t* i, size_t n, void* s) { return ntt_entry(c, true, o, i, n, s); }

// ------------------------------------------------------------------------------------------------
template <class Arith, int OP>
static void launch_dy(dpfhe_ctx* c, u64* out, const u64* a, const u64* b, size_t npolys, hipStream_t s, int b_period = 0) {
    // distinct streams of the launch: a, b (unless broadcast or the same buffer), the result (unless in place; read as well by mul_add)
    const int streams = 1 + ((OP != DY_NEG && !b_period && b != a) ? 1 : 0) + ((out != a && out != b) ? 1 : 0);
    const bool nt = (npolys << c->log2n) * sizeof(u64) * (size_t)streams > ((size_t)256 << 20);   // cannot stay in the Infinity Cache
    if (nt) hipLaunchKernelGGL((dyadic_kernel<Arith, OP, true>), dim3((unsigned)npolys), dim3(256), 0, s, out, a, b, tables_of<Arith>(c).lc,
                               (int)c->n_limbs, 1 << c->log2n, b_period);
    else hipLaunchKernelGGL((dyadic_kernel<Arith, OP, false>), dim3((unsigned)npolys), dim3(256), 0, s, out, a, b, tables_of<Arith>(c).lc,
                            (int)c->n_limbs, 1 << c->log2n, b_period);
}

static int dyadic_entry(dpfhe_ctx* c, int op, uint64_t* out, const uint64_t* a, const uint64_t* b, size_t n_rns_polys, void* stream, bool broadcast_b = false) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dyadic", "null context");
    if (n_rns_polys == 0) return DPFHE_SUCCESS;
    if (!out || !a || (op != DY_NEG && !b) || misaligned(out) || misaligned(a) || misaligned(b))
        return fail(DPFHE_INVALID_ARGUMENT, "dyadic", "null or misaligned buffer");
    const size_t npolys = n_rns_polys * c->n_limbs;
    if (npolys > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dyadic", "batch too large for one launch");
    DPFHE_ON_DEVICE(c, "dyadic");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (op == DY_NEG) b = a;
    const int b_period = broadcast_b ? (int)c->n_limbs : 0;
#define DY_CASE(OP)                                                             \
    case OP:                                                                    \
        if (c->fold) launch_dy<FoldArith, OP>(c, out, a, b, npolys, s, b_period); \
        else launch_dy<ShoupArith, OP>(c, out, a, b, npolys, s, b_period);      \
        break
    switch (op) {
        DY_CASE(DY_MUL); DY_CASE(DY_MUL_ADD); DY_CASE(DY_ADD); DY_CASE(DY_SUB); DY_CASE(DY_NEG);
        default: return fail(DPFHE_INVALID_ARGUMENT, "dyadic", "bad op");
    }
#undef DY_CASE
    return check_launch("dyadic kernel launch");
}

extern "C" int dpfhe_dyadic_mul(dpfhe_ctx* c, uint64_t* o, const uint64_t* a, const uint64_t* b, size_t n, void* s) { return dyadic_entry(c, DY_MUL, o, a, b, n, s); }
extern "C" int dpfhe_dyadic_mul_add(dpfhe_ctx* c, uint64_t* acc, const uint64_t* a, const uint64_t* b, size_t n, void* s) { return dyadic_entry(c, DY_MUL_ADD, acc, a, b, n, s); }
extern "C" int dpfhe_add(dpfhe_ctx* c, uint64_t* o, const uint64_t* a, const uint64_t* b, size_t n, void* s) { return dyadic_entry(c, DY_ADD, o, a, b, n, s); }
extern "C" int dpfhe_sub(dpfhe_ctx* c, uint64_t* o, const uint64_t* a, const uint64_t* b, size_t n, void* s) { return dyadic_entry(c, DY_SUB, o, a, b, n, s); }
extern "C" int dpfhe_negate(dpfhe_ctx* c, uint64_t* o, const uint64_t* a, size_t n, void* s) { return dyadic_entry(c, DY_NEG, o, a, nullptr, n, s); }
// A7: every residue polynomial of a times ONE plaintext (an RNS polynomial of L limbs), one launch for the whole batch
extern "C" int dpfhe_multiply_plain(dpfhe_ctx* c, uint64_t* o, const uint64_t* a, const uint64_t* pt, size_t n, void* s) { return dyadic_entry(c, DY_MUL, o, a, pt, n, s, true); }

// ------------------------------------------------------------------------------------------------
// ---- ring degrees above 8192: the fused kernels stop there (kernels_large.h); the same operations composed from the batched transforms
// and one-pass streaming kernels.  Their scratch comes from per-stream arenas (StreamScratch below).
static const uint32_t kFusedMaxLog2N = 13;

// Scratch of the composed large-ring operations: one ARENA per (context, stream), a plain hipMalloc made the first time that stream runs a
// composed operation and grown (hipStreamSynchronize of that stream + hipFree + hipMalloc) only when a larger slice than ever before arrives;
// kept until dpfhe_ctx_destroy, dpfhe_ctx_release_scratch or eviction.  Work on one stream is ordered, so consecutive calls share their stream's arena without a fence; different
// streams never share one.  Steady state: no allocation, no synchronisation.
// (Rounds 2-4 took the scratch from a stream-ordered memory pool - hipMallocFromPoolAsync / hipFreeAsync.  Round 5 found blocks of tens of MiB from
// that pool giving wrong results and millisecond allocation times in a process without PyTorch - the C++ programs; tools/gpu_r05_f.sh -, while the
// same calls under PyTorch were exact: the arena has no such dependence on the runtime's allocator state.)
// The arena is created and grown on the CONTEXT's device whatever the calling thread's current device is (round 6: the advisor's finding - one caller
// allocated before it took the device guard).  A context keeps at most kMaxScratchArenas arenas: a caller that rotates through a pool of streams evicts
// the least recently used one (its stream is synchronised first); dpfhe_ctx_release_scratch hands an arena back explicitly (a stream about to be
// destroyed, a phase that will not run composed operations again).
static const size_t kMaxScratchArenas = 16;
struct StreamScratch {
    u64* p = nullptr;
    dpfhe_ctx* c;
    hipStream_t s;
    StreamScratch(dpfhe_ctx* ctx, hipStream_t st) : c(ctx), s(st) {}
    int alloc(size_t words, const char* what) {
        std::lock_guard<std::mutex> lock(c->scratch_mutex);
        dpfhe_ctx::ScratchArena* arena = nullptr;
        for (auto& a : c->scratch_arenas) if (a.stream == s) arena = &a;
        if (arena && arena->words >= words) { arena->last_use = ++c->scratch_clock; p = arena->p; return DPFHE_SUCCESS; }
        DeviceGuard guard(c->device);   // everything below touches the device: the context's, not the caller's current one
        if (guard.err != hipSuccess) return fail(DPFHE_DEVICE_ERROR, what, hipGetErrorString(guard.err));
        if (arena && arena->p) {   // growth: the stream's earlier launches may still be using the old block
            hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess) return fail(DPFHE_DEVICE_ERROR, what, hipGetErrorString(e));
            (void)hipFree(arena->p);
            arena->p = nullptr; arena->words = 0;
        }
        if (!arena && c->scratch_arenas.size() >= kMaxScratchArenas) {   // evict the least recently used arena (another stream's: wait for that stream)
            size_t lru = 0;
            for (size_t i = 1; i < c->scratch_arenas.size(); ++i) if (c->scratch_arenas[i].last_use < c->scratch_arenas[lru].last_use) lru = i;
            if (c->scratch_arenas[lru].p) {
                if (hipStreamSynchronize(c->scratch_arenas[lru].stream) != hipSuccess) (void)hipGetLastError();   // a destroyed stream: hipFree below synchronises the device anyway
                (void)hipFree(c->scratch_arenas[lru].p);
            }
            c->scratch_arenas.erase(c->scratch_arenas.begin() + (long)lru);
        }
        const size_t gran = (size_t)1 << 21;                                    // 16 MiB steps
        const size_t want = (words + gran - 1) / gran * gran;
        u64* q = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&q), want * sizeof(u64));
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(DPFHE_OUT_OF_MEMORY, what, hipGetErrorString(e)); }
        if (arena) { arena->p = q; arena->words = want; arena->last_use = ++c->scratch_clock; }
        else c->scratch_arenas.push_back(dpfhe_ctx::ScratchArena{s, q, want, ++c->scratch_clock});
        p = q;
        return DPFHE_SUCCESS;
    }
};

extern "C" int dpfhe_ctx_release_scratch(dpfhe_ctx* c, void* stream, uint32_t flags) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_release_scratch", "null context");
    if (flags & ~(uint32_t)DPFHE_SCRATCH_ALL) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ctx_release_scratch", "unknown flag");
    DPFHE_ON_DEVICE(c, "dpfhe_ctx_release_scratch");
    std::lock_guard<std::mutex> lock(c->scratch_mutex);
    int rc = DPFHE_SUCCESS;
    for (size_t i = 0; i < c->scratch_arenas.size();) {
        dpfhe_ctx::ScratchArena& a = c->scratch_arenas[i];
        if (!(flags & DPFHE_SCRATCH_ALL) && a.stream != static_cast<hipStream_t>(stream)) { ++i; continue; }
        if (a.p) {
            hipError_t e = hipStreamSynchronize(a.stream);   // the stream's composed operations may still be reading the block
            if (e != hipSuccess) { (void)hipGetLastError(); rc = fail(DPFHE_DEVICE_ERROR, "dpfhe_ctx_release_scratch", hipGetErrorString(e)); }
            (void)hipFree(a.p);
        }
        c->scratch_arenas.erase(c->scratch_arenas.begin() + (long)i);
    }
    return rc;
}
extern "C" size_t dpfhe_ctx_scratch_bytes(const dpfhe_ctx* c) {
    if (!c) return 0;
    std::lock_guard<std::mutex> lock(const_cast<dpfhe_ctx*>(c)->scratch_mutex);
    size_t w = 0;
    for (const auto& a : c->scratch_arenas) w += a.words;
    return w * sizeof(u64);
}

static int ct_mul_composed_slice(dpfhe_ctx* c, uint64_t* d_out3, const uint64_t* d_a2, const uint64_t* d_b2, size_t batch, uint32_t flags, hipStream_t s) {
    const size_t L = c->n_limbs, n = (size_t)1 << c->log2n, poly = L * n;
    const int chunks = (int)(n / 512);
    const size_t grid = batch * L * (size_t)chunks;
    if (grid > kMaxGrid || !ntt_grid_fits(c, batch * 3 * L)) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ct_mul", "batch too large for one launch");
    StreamScratch ws(c, s);
    const u64 *pa = d_a2, *pb = d_b2;
    if (!(flags & DPFHE_IN_NTT)) {   // transformed copies of the operands: [a | b], 2 * batch * 2 * L residue polynomials
        const bool square = d_a2 == d_b2;
        if (int rc = ws.alloc((square ? 2 : 4) * batch * poly, "dpfhe_ct_mul (scratch for the transformed operands)")) return rc;
        if (int rc = ntt_launch(c, false, ws.p, d_a2, batch * 2 * L, s)) return rc;
        if (!square) if (int rc = ntt_launch(c, false, ws.p + 2 * batch * poly, d_b2, batch * 2 * L, s)) return rc;
        pa = ws.p;
        pb = square ? ws.p : ws.p + 2 * batch * poly;
    }
    const LimbConst* lc = c->fold ? c->foldt.lc : c->shoup.lc;
    if (c->fold) hipLaunchKernelGGL((tensor3_kernel<FoldArith>), dim3((unsigned)grid), dim3(256), 0, s, d_out3, pa, pb, lc, (int)L, (int)n, chunks);
    else hipLaunchKernelGGL((tensor3_kernel<ShoupArith>), dim3((unsigned)grid), dim3(256), 0, s, d_out3, pa, pb, lc, (int)L, (int)n, chunks);
    if (int rc = check_launch("tensor product kernel launch")) return rc;
    if (!(flags & DPFHE_OUT_NTT)) return ntt_launch(c, true, d_out3, d_out3, batch * 3 * L, s);
    return DPFHE_SUCCESS;
}

// RNS-digit key switch: out2 = [mask-selected components of in] + sum_j NTT^-1( NTT([in_{last comp}]_j mod q_i) (.) evk[j] )
static int key_switch_composed_slice(dpfhe_ctx* c, uint64_t* d_out2, const uint64_t* d_in, int in_comps, int add_mask, const uint64_t* d_evk, size_t batch,
                                     hipStream_t s, const char* what) {
    const size_t L = c->n_limbs, n = (size_t)1 << c->log2n;
    const int chunks = (int)(n / 512);
    const size_t lift_grid = batch * L * L * (size_t)chunks;
    if (lift_grid > kMaxGrid || 2 * batch * L * (size_t)chunks > kMaxGrid || !ntt_grid_fits(c, batch * L * L)) return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch");
    StreamScratch ws(c, s);
    if (int rc = ws.alloc(batch * L * L * n, what)) return rc;
    const LimbConst* lc = c->fold ? c->foldt.lc : c->shoup.lc;
    if (c->fold) hipLaunchKernelGGL((lift_rns_digits_kernel<FoldArith>), dim3((unsigned)lift_grid), dim3(256), 0, s, ws.p, d_in, in_comps, in_comps - 1, lc, (int)L, (int)n, chunks);
    else hipLaunchKernelGGL((lift_rns_digits_kernel<ShoupArith>), dim3((unsigned)lift_grid), dim3(256), 0, s, ws.p, d_in, in_comps, in_comps - 1, lc, (int)L, (int)n, chunks);
    if (int rc = check_launch("digit lift kernel launch")) return rc;
    if (int rc = ntt_launch(c, false, ws.p, ws.p, batch * L * L, s)) return rc;
    const unsigned grid = (unsigned)(batch * L * (size_t)chunks);
    if (c->fold) hipLaunchKernelGGL((key_inner_product_kernel<FoldArith>), dim3(grid), dim3(256), 0, s, d_out2, ws.p, d_evk, lc, (int)L, (int)L, (int)n, chunks);
    else hipLaunchKernelGGL((key_inner_product_kernel<ShoupArith>), dim3(grid), dim3(256), 0, s, d_out2, ws.p, d_evk, lc, (int)L, (int)L, (int)n, chunks);
    if (int rc = check_launch("key inner product kernel launch")) return rc;
    if (int rc = ntt_launch(c, true, d_out2, d_out2, batch * 2 * L, s)) return rc;
    hipLaunchKernelGGL(add_back_kernel, dim3(2 * grid), dim3(256), 0, s, d_out2, d_in, in_comps, add_mask, lc, (int)L, (int)n, chunks);
    return check_launch("add-back kernel launch");
}

// Large batches go through in slices whose scratch stays below the context's scratch limit (the slices run back to back on the caller's stream and reuse
// the pool's block): the scratch of a composed operation is 4 (multiply) or L^2 / 2 (key switch) times its input.
static size_t slice_items(const dpfhe_ctx* c, size_t batch, size_t scratch_words_per_item) {   // 1 GiB of scratch unless dpfhe_ctx_set_scratch_limit said otherwise
    const size_t fit = c->scratch_limit_words.load(std::memory_order_relaxed) / scratch_words_per_item;
    return fit == 0 ? 1 : (fit < batch ? fit : batch);
}
static int ct_mul_composed(dpfhe_ctx* c, uint64_t* d_out3, const uint64_t* d_a2, const uint64_t* d_b2, size_t batch, uint32_t flags, hipStream_t s) {
    const size_t poly = (size_t)c->n_limbs << c->log2n, per = slice_items(c, batch, 4 * poly);
    for (size_t i = 0; i < batch; i += per) {
        const size_t m = batch - i < per ? batch - i : per;
        if (int rc = ct_mul_composed_slice(c, d_out3 + i * 3 * poly, d_a2 + i * 2 * poly, d_b2 + i * 2 * poly, m, flags, s)) return rc;
    }
    return DPFHE_SUCCESS;
}
static int key_switch_composed(dpfhe_ctx* c, uint64_t* d_out2, const uint64_t* d_in, int in_comps, int add_mask, const uint64_t* d_evk, size_t batch,
                               hipStream_t s, const char* what) {
    const size_t poly = (size_t)c->n_limbs << c->log2n, per = slice_items(c, batch, c->n_limbs * poly);
    for (size_t i = 0; i < batch; i += per) {
        const size_t m = batch - i < per ? batch - i : per;
        if (int rc = key_switch_composed_slice(c, d_out2 + i * 2 * poly, d_in + i * (size_t)in_comps * poly, in_comps, add_mask, d_evk, m, s, what)) return rc;
    }
    return DPFHE_SUCCESS;
}

extern "C" int dpfhe_ct_mul(dpfhe_ctx* c, uint64_t* d_out3, const uint64_t* d_a2, const uint64_t* d_b2, size_t batch,
                            uint32_t flags, void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ct_mul", "null context");
    if (flags & ~(DPFHE_IN_NTT | DPFHE_OUT_NTT)) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ct_mul", "unknown flag");
    if (batch == 0) return DPFHE_SUCCESS;
    if (!d_out3 || !d_a2 || !d_b2 || misaligned(d_out3) || misaligned(d_a2) || misaligned(d_b2))
        return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ct_mul", "null or misaligned buffer");
    {   // output items are 3 L N words apart, input items 2 L N: a workgroup would overwrite operands another one has not read yet
        const size_t poly = (size_t)c->n_limbs << c->log2n;
        if (overlaps(d_out3, batch * 3 * poly, d_a2, batch * 2 * poly) || overlaps(d_out3, batch * 3 * poly, d_b2, batch * 2 * poly))
            return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ct_mul", "output overlaps an operand");
    }
    const size_t blocks = batch * c->n_limbs;
    if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_ct_mul", "batch too large for one launch");
    DPFHE_ON_DEVICE(c, "dpfhe_ct_mul");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c->log2n > kFusedMaxLog2N) return ct_mul_composed(c, d_out3, d_a2, d_b2, batch, flags, s);
    if (flags == 0 && c->fold) {   // coefficient domain in and out: the form chosen for this box (bit-identical results)
        const int v = c->ct_mul_variant.load(std::memory_order_relaxed);
        if (v != ct_mul_default_variant((int)c->log2n) && ct_mul_variant_compiled(c, v)) {
            if (launch_ct_mul_variant<FoldArith>((int)c->log2n, v, d_out3, d_a2, d_b2, blocks, c->foldt, s)) return fail(DPFHE_INVALID_STATE, "dpfhe_ct_mul", "variant not compiled");
            return check_launch("ct_mul kernel launch");
        }
    }
    int rc;
    if (c->classes && c->log2n <= kFusedMaxLog2N) {   // one launch per arithmetic class of the limbs (dpfhe_ctx::classes)
        const size_t pairs = blocks / c->n_limbs;
        rc = for_each_class(c, c->n_limbs, [&](const auto& tb) { return launch_ct_mul((int)c->log2n, flags, d_out3, d_a2, d_b2, pairs * (size_t)tb.n_active, tb, s); });
    } else {
        rc = c->fold ? launch_ct_mul<FoldArith>((int)c->log2n, flags, d_out3, d_a2, d_b2, blocks, c->foldt, s)
                     : launch_ct_mul<ShoupArith>((int)c->log2n, flags, d_out3, d_a2, d_b2, blocks, c->shoup, s);
    }
    if (rc) return fail(DPFHE_INVALID_STATE, "dpfhe_ct_mul", "no kernel geometry for this log2_n");
    return check_launch("ct_mul kernel launch");
}

// diagnostics: the quad form with per-workgroup timestamps (kernels.h ct_mul_quad_kernel<..., TRACE>); FoldArith contexts at N = 4096
extern "C" int dpfhe_debug_ct_mul_trace(dpfhe_ctx* c, uint64_t* d_out3, const uint64_t* d_a2, const uint64_t* d_b2, size_t batch, uint64_t* d_trace, void* stream) {
    if (!c || !d_out3 || !d_a2 || !d_b2 || !d_trace) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_debug_ct_mul_trace", "null argument");
    if (!c->fold || c->log2n != 12) return fail(DPFHE_INVALID_STATE, "dpfhe_debug_ct_mul_trace", "FoldArith contexts at N = 4096 only");
    const size_t blocks = batch * c->n_limbs, poly = (size_t)c->n_limbs << c->log2n;
    if (batch == 0 || blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_debug_ct_mul_trace", "bad batch");
    if (overlaps(d_out3, batch * 3 * poly, d_a2, batch * 2 * poly) || overlaps(d_out3, batch * 3 * poly, d_b2, batch * 2 * poly))
        return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_debug_ct_mul_trace", "output overlaps an operand");
    DPFHE_ON_DEVICE(c, "dpfhe_debug_ct_mul_trace");
    if (launch_ct_mul_trace<FoldArith>((int)c->log2n, d_out3, d_a2, d_b2, blocks, c->foldt, d_trace, static_cast<hipStream_t>(stream)))
        return fail(DPFHE_INVALID_STATE, "dpfhe_debug_ct_mul_trace", "not compiled");
    return check_launch("traced ct_mul kernel launch");
}

extern "C" int dpfhe_relinearize(dpfhe_ctx* c, uint64_t* d_out2, const uint64_t* d_in3, const uint64_t* d_evk, size_t batch, void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_relinearize", "null context");
    if (batch == 0) return DPFHE_SUCCESS;
    if (!d_out2 || !d_in3 || !d_evk || misaligned(d_out2) || misaligned(d_in3) || misaligned(d_evk))
        return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_relinearize", "null or misaligned buffer");
    {   // input items are 3 L N words apart, output items 2 L N: any overlap lets one workgroup overwrite another's unread input
        const size_t poly = (size_t)c->n_limbs << c->log2n;
        if (overlaps(d_out2, batch * 2 * poly, d_in3, batch * 3 * poly)) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_relinearize", "output overlaps the input");
    }
    const size_t blocks = batch * c->n_limbs;
    if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_relinearize", "batch too large for one launch");
    DPFHE_ON_DEVICE(c, "dpfhe_relinearize");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c->log2n > kFusedMaxLog2N) return key_switch_composed(c, d_out2, d_in3, 3, 3, d_evk, batch, s, "dpfhe_relinearize");
    const int rc = relin_launch(c, 0, d_out2, d_in3, d_evk, 0, 1, blocks, s);
    if (rc) return fail(DPFHE_INVALID_STATE, "dpfhe_relinearize", "no kernel geometry for this log2_n");
    return check_launch("relin kernel launch");
}

extern "C" int dpfhe_switch_key(dpfhe_ctx* c, uint64_t* d_out2, const uint64_t* d_in2, const uint64_t* d_key, size_t batch, void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_switch_key", "null context");
    if (batch == 0) return DPFHE_SUCCESS;
    if (!d_out2 || !d_in2 || !d_key || misaligned(d_out2) || misaligned(d_in2) || misaligned(d_key) ||
        overlaps(d_out2, batch * 2 * ((size_t)c->n_limbs << c->log2n), d_in2, batch * 2 * ((size_t)c->n_limbs << c->log2n)))
        return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_switch_key", "null, misaligned or aliased buffer");
    const size_t blocks = batch * c->n_limbs;
    if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_switch_key", "batch too large for one launch");
    DPFHE_ON_DEVICE(c, "dpfhe_switch_key");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c->log2n > kFusedMaxLog2N) return key_switch_composed(c, d_out2, d_in2, 2, 1, d_key, batch, s, "dpfhe_switch_key");
    const int rc = relin_launch(c, 1, d_out2, d_in2, d_key, 0, 1, blocks, s);
    if (rc) return fail(DPFHE_INVALID_STATE, "dpfhe_switch_key", "no kernel geometry for this log2_n");
    return check_launch("switch_key kernel launch");
}

extern "C" int dpfhe_rescale(dpfhe_ctx* c, uint64_t* d_out, const uint64_t* d_in, size_t n_rns_polys, void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_rescale", "null context");
    if (c->n_limbs < 2) return fail(DPFHE_INVALID_STATE, "dpfhe_rescale", "no limb left to drop");
    if (n_rns_polys == 0) return DPFHE_SUCCESS;
    if (!d_out || !d_in || misaligned(d_out) || misaligned(d_in)) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_rescale", "null or misaligned buffer");
    const int n = 1 << c->log2n;
    const int chunks = (n + 511) / 512;
    const size_t blocks = n_rns_polys * (c->n_limbs - 1) * (size_t)chunks;
    if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_rescale", "batch too large for one launch");
    DPFHE_ON_DEVICE(c, "dpfhe_rescale");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c->fold) hipLaunchKernelGGL((rescale_kernel<FoldArith>), dim3((unsigned)blocks), dim3(256), 0, s, d_out, d_in, (const u64*)nullptr, 0, 0, c->foldt.lc, c->d_rescale, (int)c->n_limbs, n, chunks);
    else hipLaunchKernelGGL((rescale_kernel<ShoupArith>), dim3((unsigned)blocks), dim3(256), 0, s, d_out, d_in, (const u64*)nullptr, 0, 0, c->shoup.lc, c->d_rescale, (int)c->n_limbs, n, chunks);
    return check_launch("rescale kernel launch");
}

// Above N = 8192 (no fused key-switch kernel): out[item][2][L][N] (NTT domain over Q P) = sum_j NTT(lift(digit_j of item)) (.) key_j, the digits being the Ld
// limbs of one component (`comp0`: component pointer of item 0, items `item_stride` words apart), keys per item group (key_stride / key_group as in
// launch_relin; 0 / 1 = one key).  Composed from lift_digits_kernel, the batched forward transform and key_inner_product_kernel; scratch from the
// context's pool, batch sliced so that a slice's Ld L N words per item stay below the scratch limit.
static int key_products_composed(dpfhe_ctx* c, uint64_t* d_out_qp, const uint64_t* comp0, size_t item_stride, const uint64_t* d_keys, size_t key_stride, unsigned key_group,
                                 size_t batch, hipStream_t s, const char* what) {
    const size_t L = c->n_limbs, Ld = L - 1;
    const int n = 1 << c->log2n, ch = n / 512;
    const LimbConst* lc = c->fold ? c->foldt.lc : c->shoup.lc;
    const size_t per = slice_items(c, batch, Ld * L * (size_t)n);
    for (size_t i0 = 0; i0 < batch; i0 += per) {
        const size_t m = batch - i0 < per ? batch - i0 : per;
        const size_t lift_grid = m * Ld * L * (size_t)ch;
        if (lift_grid > kMaxGrid || !ntt_grid_fits(c, m * Ld * L)) return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch (lower the scratch limit)");
        StreamScratch ws(c, s);
        if (int rc = ws.alloc(m * Ld * L * n, what)) return rc;
        const u64* comp = comp0 + i0 * item_stride;
        if (c->fold) hipLaunchKernelGGL((lift_digits_kernel<FoldArith>), dim3((unsigned)lift_grid), dim3(256), 0, s, ws.p, comp, item_stride, lc, (int)L, n, ch);
        else hipLaunchKernelGGL((lift_digits_kernel<ShoupArith>), dim3((unsigned)lift_grid), dim3(256), 0, s, ws.p, comp, item_stride, lc, (int)L, n, ch);
        if (int rc = check_launch("digit lift kernel launch")) return rc;
        if (int rc = ntt_launch(c, false, ws.p, ws.p, m * Ld * L, s)) return rc;
        const unsigned grid = (unsigned)(m * L * (size_t)ch);
        u64* o = d_out_qp + i0 * 2 * L * n;
        if (c->fold) hipLaunchKernelGGL((key_inner_product_kernel<FoldArith>), dim3(grid), dim3(256), 0, s, o, ws.p, d_keys, lc, (int)Ld, (int)L, n, ch, key_stride, key_group, (unsigned)i0);
        else hipLaunchKernelGGL((key_inner_product_kernel<ShoupArith>), dim3(grid), dim3(256), 0, s, o, ws.p, d_keys, lc, (int)Ld, (int)L, n, ch, key_stride, key_group, (unsigned)i0);
        if (int rc = check_launch("key inner product kernel launch")) return rc;
    }
    return DPFHE_SUCCESS;
}

// hybrid key switching = inner product over all L limbs (relin_kernel MODE 2/3) + divide by the special prime and add (c0, c1)
static int hybrid_entry(dpfhe_ctx* c, const char* what, int in_comps, uint64_t* d_out2, const uint64_t* d_in, const uint64_t* d_key,
                        uint64_t* d_work, size_t batch, void* stream, size_t key_stride = 0, unsigned key_group = 1) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, what, "null context");
    if (c->n_limbs < 2) return fail(DPFHE_INVALID_STATE, what, "the extended context needs at least one data limb and the special prime");
    if (batch == 0) return DPFHE_SUCCESS;
    if (!d_out2 || !d_in || !d_key || !d_work || misaligned(d_out2) || misaligned(d_in) || misaligned(d_key) || misaligned(d_work))
        return fail(DPFHE_INVALID_ARGUMENT, what, "null or misaligned buffer");
    const size_t L = c->n_limbs, Ld = L - 1;
    const int n = 1 << c->log2n;
    {   // the rescale-add pass reads (c0, c1) of the input while it writes the output; the work buffer is written by pass 1
        const size_t in_words = batch * (size_t)in_comps * Ld * n, out_words = batch * 2 * Ld * n, work_words = batch * 2 * L * n;
        if (overlaps(d_out2, out_words, d_in, in_words) || overlaps(d_work, work_words, d_in, in_words) || overlaps(d_work, work_words, d_out2, out_words))
            return fail(DPFHE_INVALID_ARGUMENT, what, "output, input and work buffers must not overlap");
    }
    const size_t blocks = batch * L;
    // the grouped launch pads the grid to whole XCD rounds (launch_impl.h launch_relin): ceil(blocks / group / 8) * 8 * group workgroups
    const size_t kg = key_group ? key_group : 1, padded = ((blocks / kg + 7) / 8) * 8 * kg;
    if (blocks > kMaxGrid || padded > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch");
    DPFHE_ON_DEVICE(c, "hybrid key switch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int mode = in_comps == 3 ? 2 : 3;
    if (c->log2n > kFusedMaxLog2N) {
        // no fused kernel: digits lifted to Q P -> one batched transform -> inner products with the key(s) -> batched inverse into `work`;
        // in slices whose scratch (Ld L N words per item) stays below the context's limit
        if (int rc = key_products_composed(c, d_work, d_in + (size_t)(in_comps - 1) * Ld * n, (size_t)in_comps * Ld * n, d_key, key_stride, (unsigned)kg, batch, s, what)) return rc;
        if (!ntt_grid_fits(c, batch * 2 * L)) return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch");
        if (int rc = ntt_launch(c, true, d_work, d_work, batch * 2 * L, s)) return rc;
    } else {
        const int rc = relin_launch(c, mode, d_work, d_in, d_key, key_stride, key_group, blocks, s);
        if (rc) return fail(DPFHE_INVALID_STATE, what, "no kernel geometry for this log2_n");
        int e = check_launch("hybrid key-switch kernel launch");
        if (e) return e;
    }
    // divide by P with rounding and add c0 (and c1 for relinearisation): one pass over the 2*batch polynomials of `work`
    const int chunks = (n + 511) / 512;
    const size_t rblocks = batch * 2 * Ld * (size_t)chunks;
    if (rblocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch");
    const int add_mask = in_comps == 3 ? 3 : 1;
    if (c->fold) hipLaunchKernelGGL((rescale_kernel<FoldArith>), dim3((unsigned)rblocks), dim3(256), 0, s, d_out2, d_work, d_in, in_comps, add_mask, c->foldt.lc, c->d_rescale, (int)L, n, chunks);
    else hipLaunchKernelGGL((rescale_kernel<ShoupArith>), dim3((unsigned)rblocks), dim3(256), 0, s, d_out2, d_work, d_in, in_comps, add_mask, c->shoup.lc, c->d_rescale, (int)L, n, chunks);
    return check_launch("hybrid rescale launch");
}

extern "C" int dpfhe_relinearize_hybrid(dpfhe_ctx* c, uint64_t* d_out2, const uint64_t* d_in3, const uint64_t* d_key, uint64_t* d_work, size_t batch,
                                        void* stream) {
    return hybrid_entry(c, "dpfhe_relinearize_hybrid", 3, d_out2, d_in3, d_key, d_work, batch, stream);
}
extern "C" int dpfhe_switch_key_hybrid(dpfhe_ctx* c, uint64_t* d_out2, const uint64_t* d_in2, const uint64_t* d_key, uint64_t* d_work, size_t batch,
                                       void* stream) {
    return hybrid_entry(c, "dpfhe_switch_key_hybrid", 2, d_out2, d_in2, d_key, d_work, batch, stream);
}

// automorphism launch: polynomials up to 64 KiB are staged through LDS, larger ones gathered from global memory
static void launch_galois_multi(hipStream_t s, unsigned grid, u64* out, const u64* in, size_t in_item_stride, const LimbConst* lc, int n_limbs, int n,
                                int polys_per_item, const GaloisInvs& inv, unsigned in_mod, unsigned item0) {
    if (n <= 8192)
        hipLaunchKernelGGL(galois_multi_kernel<true>, dim3(grid), dim3(256), (size_t)n * 8, s, out, in, in_item_stride, lc, n_limbs, n, polys_per_item, inv, in_mod, item0);
    else
        hipLaunchKernelGGL(galois_multi_kernel<false>, dim3(grid), dim3(256), 0, s, out, in, in_item_stride, lc, n_limbs, n, polys_per_item, inv, in_mod, item0);
}

static unsigned galois_inverse(unsigned g, unsigned two_n) {  // g^-1 mod 2N by Newton iteration (g odd): x <- x (2 - g x)
    unsigned inv = 1;
    for (int i = 0; i < 5; ++i) inv *= 2u - g * inv;
    return inv & (two_n - 1u);
}

// N3: `batch` rotations in one pass - item i = key-switched sigma_{g_(i / group)}(input item i, or the single input when n_in == 1);
// `group` consecutive items share an element and its key (several tokens, rotation-major order)
static int rotate_batch_impl(dpfhe_ctx* c, const char* what, uint64_t* d_out2, const uint64_t* d_in2, size_t n_in, const uint32_t* galois_elts, size_t n_elts,
                             size_t group, const uint64_t* d_keys, uint64_t* d_work, uint64_t* d_rotated, size_t batch, void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, what, "null context");
    if (c->n_limbs < 2) return fail(DPFHE_INVALID_STATE, what, "the extended context needs at least one data limb and the special prime");
    if (batch == 0) return DPFHE_SUCCESS;
    if (n_in != 1 && n_in != batch) return fail(DPFHE_INVALID_ARGUMENT, what, "n_in must be 1 (one input, many rotations) or equal to batch");
    if (group == 0 || n_elts * group != batch) return fail(DPFHE_INVALID_ARGUMENT, what, "batch must be n_elts * group");
    if (!d_out2 || !d_in2 || !galois_elts || !d_keys || !d_work || !d_rotated || misaligned(d_out2) || misaligned(d_in2) || misaligned(d_keys) ||
        misaligned(d_work) || misaligned(d_rotated) || d_rotated == d_in2)
        return fail(DPFHE_INVALID_ARGUMENT, what, "null, misaligned or aliased buffer");
    const size_t L = c->n_limbs, Ld = L - 1;
    const int n = 1 << c->log2n;
    const unsigned two_n = 2u << c->log2n;
    for (size_t i = 0; i < n_elts; ++i)
        if (!(galois_elts[i] & 1u) || galois_elts[i] >= two_n) return fail(DPFHE_INVALID_ARGUMENT, what, "galois elements must be odd and < 2N");
    const size_t ct_words = 2 * Ld * (size_t)n, key_words = Ld * 2 * L * (size_t)n;
    if (n_in == batch && overlaps(d_rotated, batch * ct_words, d_in2, batch * ct_words)) return fail(DPFHE_INVALID_ARGUMENT, what, "d_rotated overlaps the input");
    const LimbConst* lc = c->fold ? c->foldt.lc : c->shoup.lc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DPFHE_ON_DEVICE(c, what);
    for (size_t first = 0; first < batch; first += kMaxGaloisBatch) {   // the elements travel as kernel arguments, 64 at a time
        const size_t cnt = batch - first < (size_t)kMaxGaloisBatch ? batch - first : (size_t)kMaxGaloisBatch;
        GaloisInvs inv{};
        for (size_t i = 0; i < cnt; ++i) inv.v[i] = galois_inverse(galois_elts[(first + i) / group], two_n);
        const uint64_t* src = d_in2 + (n_in == 1 ? 0 : first * ct_words);
        launch_galois_multi(s, (unsigned)(cnt * 2 * Ld), d_rotated + first * ct_words, src, n_in == 1 ? (size_t)0 : ct_words, lc, (int)Ld, n, (int)(2 * Ld), inv, 0u, 0u);
        int e = check_launch("galois kernel launch");
        if (e) return e;
    }
    return hybrid_entry(c, what, 2, d_out2, d_rotated, d_keys, d_work, batch, stream, key_words, (unsigned)group);
}

extern "C" int dpfhe_rotate_hybrid_batch(dpfhe_ctx* c, uint64_t* d_out2, const uint64_t* d_in2, size_t n_in, const uint32_t* galois_elts,
                                         const uint64_t* d_keys, uint64_t* d_work, uint64_t* d_rotated, size_t batch, void* stream) {
    return rotate_batch_impl(c, "dpfhe_rotate_hybrid_batch", d_out2, d_in2, n_in, galois_elts, batch, 1, d_keys, d_work, d_rotated, batch, stream);
}

extern "C" int dpfhe_rotate_hybrid_grouped(dpfhe_ctx* c, uint64_t* d_out2, const uint64_t* d_in2, const uint32_t* galois_elts, size_t n_elts, size_t group,
                                           const uint64_t* d_keys, uint64_t* d_work, uint64_t* d_rotated, void* stream) {
    return rotate_batch_impl(c, "dpfhe_rotate_hybrid_grouped", d_out2, d_in2, n_elts * group, galois_elts, n_elts, group, d_keys, d_work, d_rotated, n_elts * group, stream);
}

static int rotate_hoisted_qp_impl(dpfhe_ctx* c, uint64_t* d_out_qp, const uint64_t* d_in2, size_t n_items, const uint32_t* galois_elts, const uint64_t* d_keys,
                                  uint64_t* d_in_ntt, uint64_t* d_digits, size_t batch, void* stream, bool prepared);

// N3, hoisted: `batch` rotations of ONE ciphertext; the digit decomposition of c1 and its Ld*L forward transforms are done once
// (d_digits), every rotation is a permutation of those words in the NTT domain + its key inner product + two inverse transforms
extern "C" int dpfhe_rotate_hybrid_hoisted(dpfhe_ctx* c, uint64_t* d_out2, const uint64_t* d_in2, size_t n_items, const uint32_t* galois_elts, const uint64_t* d_keys,
                                           uint64_t* d_work, uint64_t* d_rotated0, uint64_t* d_digits, size_t batch, void* stream) {
    const char* what = "dpfhe_rotate_hybrid_hoisted";
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, what, "null context");
    if (c->n_limbs < 2) return fail(DPFHE_INVALID_STATE, what, "the extended context needs at least one data limb and the special prime");
    if (c->log2n > 14) return fail(DPFHE_INVALID_STATE, what, "available up to N = 16384");
    if (batch == 0 || n_items == 0) return DPFHE_SUCCESS;
    const bool composed = c->log2n > (uint32_t)kMaxFusedLog2N;   // N = 16384: the deferred-division pipeline below never touches d_work / d_rotated0 - they may be NULL
    if (!d_out2 || !d_in2 || !galois_elts || !d_keys || (!composed && (!d_work || !d_rotated0)) || !d_digits || misaligned(d_out2) || misaligned(d_in2) || misaligned(d_keys) ||
        misaligned(d_work) || misaligned(d_rotated0) || misaligned(d_digits))
        return fail(DPFHE_INVALID_ARGUMENT, what, "null or misaligned buffer");
    const size_t L = c->n_limbs, Ld = L - 1, T = n_items, total = batch * T;
    const int n = 1 << c->log2n;
    const unsigned two_n = 2u << c->log2n;
    for (size_t i = 0; i < batch; ++i)
        if (!(galois_elts[i] & 1u) || galois_elts[i] >= two_n) return fail(DPFHE_INVALID_ARGUMENT, what, "galois elements must be odd and < 2N");
    const size_t in_words = T * 2 * Ld * (size_t)n, out_words = total * 2 * Ld * n, work_words = d_work ? total * 2 * L * n : 0, rot_words = d_rotated0 ? total * Ld * n : 0,
                 dig_words = T * Ld * L * (size_t)n;
    if (overlaps(d_out2, out_words, d_in2, in_words) || overlaps(d_out2, out_words, d_work, work_words) || overlaps(d_out2, out_words, d_rotated0, rot_words) ||
        overlaps(d_work, work_words, d_rotated0, rot_words) || overlaps(d_digits, dig_words, d_work, work_words) || overlaps(d_digits, dig_words, d_out2, out_words) ||
        overlaps(d_digits, dig_words, d_rotated0, rot_words) || overlaps(d_digits, dig_words, d_in2, in_words) || overlaps(d_rotated0, rot_words, d_in2, in_words))
        return fail(DPFHE_INVALID_ARGUMENT, what, "buffers must not overlap");
    const size_t key_words = Ld * 2 * L * (size_t)n;
    const int chunks = (n + 511) / 512;
    if (composed) {
        DPFHE_ON_DEVICE(c, what);
        // N = 16384 (round 5): no fused hoisted kernel - the deferred-division pipeline instead: dpfhe_rotate_hoisted_qp gives, per rotation and item,
        // P sigma_g(ct) + its key-switching term in the NTT domain over Q P; one inverse transform and the division by P per term finish it (the same words:
        // tests/test_rlwe_semantics.py).  Scratch (the Q P terms + the transformed inputs) from the stream's arena, rotations in slices under the scratch limit;
        // d_work / d_rotated0 are not used on this path.
        const size_t item_qp = T * 2 * L * (size_t)n, fit = c->scratch_limit_words.load(std::memory_order_relaxed) / item_qp;
        const size_t per = fit > 2 ? (fit - 2 < batch ? fit - 2 : batch) : 1;     // (+ block 0 and the transformed inputs)
        if (per * T * 2 * Ld * (size_t)chunks > kMaxGrid || !ntt_grid_fits(c, per * T * 2 * L)) return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch (lower the scratch limit)");
        StreamScratch ws(c, static_cast<hipStream_t>(stream));       // one arena for every slice: the transformed inputs and digits are prepared by the first slice only
        if (int rc = ws.alloc((per + 1) * item_qp + in_words, what)) return rc;
        u64* qp = ws.p;
        u64* in_ntt = qp + (per + 1) * item_qp;
        for (size_t r0 = 0; r0 < batch; r0 += per) {
            const size_t m = batch - r0 < per ? batch - r0 : per;
            if (int rc = rotate_hoisted_qp_impl(c, qp, d_in2, T, galois_elts + r0, d_keys + r0 * key_words, in_ntt, d_digits, m, stream, r0 != 0)) return rc;
            hipStream_t s = static_cast<hipStream_t>(stream);
            if (int rc = ntt_launch(c, true, qp + item_qp, qp + item_qp, m * T * 2 * L, s)) return rc;
            const size_t rblocks = m * T * 2 * Ld * (size_t)chunks;
            u64* o = d_out2 + r0 * T * 2 * Ld * n;
            if (c->fold) hipLaunchKernelGGL((rescale_kernel<FoldArith>), dim3((unsigned)rblocks), dim3(256), 0, s, o, qp + item_qp, (const u64*)nullptr, 0, 0, c->foldt.lc, c->d_rescale, (int)L, n, chunks);
            else hipLaunchKernelGGL((rescale_kernel<ShoupArith>), dim3((unsigned)rblocks), dim3(256), 0, s, o, qp + item_qp, (const u64*)nullptr, 0, 0, c->shoup.lc, c->d_rescale, (int)L, n, chunks);
            if (int rc = check_launch("hoisted rescale launch")) return rc;
        }
        return DPFHE_SUCCESS;
    }
    // (the key-switch launch pads its (rotation, limb[, component]) tiles to a multiple of 8 per token: launch_impl.h launch_hoisted_ks)
    if (total * 2 * Ld * (size_t)chunks > kMaxGrid || (total * L * 2 + 8 * T) > kMaxGrid || T * Ld * L * (size_t)chunks > kMaxGrid)
        return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch");
    const LimbConst* lc = c->fold ? c->foldt.lc : c->shoup.lc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DPFHE_ON_DEVICE(c, what);
    // 1. digits of every item's c1, lifted to every limb, then transformed (T * Ld RNS polynomials on the extended context)
    const unsigned lift_grid = (unsigned)(T * Ld * L * (size_t)chunks);
    if (c->fold) hipLaunchKernelGGL((lift_digits_kernel<FoldArith>), dim3(lift_grid), dim3(256), 0, s, d_digits, d_in2 + Ld * (size_t)n, 2 * Ld * (size_t)n, lc, (int)L, n, chunks);
    else hipLaunchKernelGGL((lift_digits_kernel<ShoupArith>), dim3(lift_grid), dim3(256), 0, s, d_digits, d_in2 + Ld * (size_t)n, 2 * Ld * (size_t)n, lc, (int)L, n, chunks);
    if (int e = check_launch("lift_digits kernel launch")) return e;
    if (int e = ntt_launch_items(c, false, d_digits, d_digits, T * Ld, 0, s)) return e;   // (per limb class where the context has them)
    // 2. sigma_g(c0) of every (rotation, item) for the final addition: [batch][T][Ld][N]; 64 output items per launch
    for (size_t first = 0; first < total; first += kMaxGaloisBatch) {
        const size_t cnt = total - first < (size_t)kMaxGaloisBatch ? total - first : (size_t)kMaxGaloisBatch;
        GaloisInvs inv{};
        for (size_t i = 0; i < cnt; ++i) inv.v[i] = galois_inverse(galois_elts[(first + i) / T], two_n);
        launch_galois_multi(s, (unsigned)(cnt * Ld), d_rotated0 + first * Ld * n, d_in2, 2 * Ld * (size_t)n, lc, (int)Ld, n, (int)Ld, inv, (unsigned)T, (unsigned)(first % T));
        if (int e = check_launch("galois kernel launch")) return e;
    }
    // 3. permuted digits (.) keys, one inverse transform per (rotation, limb, key component, item); 64 rotations per launch
    for (size_t first = 0; first < batch; first += kMaxGaloisBatch) {
        const size_t cnt = batch - first < (size_t)kMaxGaloisBatch ? batch - first : (size_t)kMaxGaloisBatch;
        const int rc = with_policy_or_classes(c, [&](const auto& tb) {
            return launch_hoisted_ks((int)c->log2n, d_work + first * T * 2 * L * n, d_digits, d_keys + first * key_words, key_words, galois_elts + first, cnt, T, tb, s);
        });
        if (rc) return fail(DPFHE_INVALID_STATE, what, "no kernel geometry for this log2_n");
        if (int e = check_launch("hoisted key-switch kernel launch")) return e;
    }
    // 4. divide by P with rounding; component 0 gets sigma_g(c0) added ([total][1][Ld][N] addend)
    const size_t rblocks = total * 2 * Ld * (size_t)chunks;
    if (c->fold) hipLaunchKernelGGL((rescale_kernel<FoldArith>), dim3((unsigned)rblocks), dim3(256), 0, s, d_out2, d_work, d_rotated0, 1, 1, c->foldt.lc, c->d_rescale, (int)L, n, chunks);
    else hipLaunchKernelGGL((rescale_kernel<ShoupArith>), dim3((unsigned)rblocks), dim3(256), 0, s, d_out2, d_work, d_rotated0, 1, 1, c->shoup.lc, c->d_rescale, (int)L, n, chunks);
    return check_launch("hoisted rescale launch");
}

// ------------------------------------------------------------------------------------------------
// N3, round 3: baby-step / giant-step sums with the division by P DEFERRED (the rotated terms stay in the NTT domain over Q P)
// ------------------------------------------------------------------------------------------------
// (prepared = true: d_in_ntt, d_digits and item block 0 already hold what steps 1-3 write - a later slice of dpfhe_rotate_hybrid_hoisted at N = 16384)
static int rotate_hoisted_qp_impl(dpfhe_ctx* c, uint64_t* d_out_qp, const uint64_t* d_in2, size_t n_items, const uint32_t* galois_elts, const uint64_t* d_keys,
                                  uint64_t* d_in_ntt, uint64_t* d_digits, size_t batch, void* stream, bool prepared) {
    const char* what = "dpfhe_rotate_hoisted_qp";
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, what, "null context");
    if (c->n_limbs < 2) return fail(DPFHE_INVALID_STATE, what, "the extended context needs at least one data limb and the special prime");
    if (c->log2n > 14) return fail(DPFHE_INVALID_STATE, what, "available up to N = 16384 (the stream kernels and the single-kernel transforms)");
    if (n_items == 0) return DPFHE_SUCCESS;
    if (!d_out_qp || !d_in2 || (batch && (!galois_elts || !d_keys)) || !d_in_ntt || !d_digits || misaligned(d_out_qp) || misaligned(d_in2) || misaligned(d_keys) ||
        misaligned(d_in_ntt) || misaligned(d_digits))
        return fail(DPFHE_INVALID_ARGUMENT, what, "null or misaligned buffer");
    const size_t L = c->n_limbs, Ld = L - 1, T = n_items;
    const int n = 1 << c->log2n;
    const unsigned two_n = 2u << c->log2n;
    for (size_t i = 0; i < batch; ++i)
        if (!(galois_elts[i] & 1u) || galois_elts[i] >= two_n) return fail(DPFHE_INVALID_ARGUMENT, what, "galois elements must be odd and < 2N");
    const size_t in_words = T * 2 * Ld * (size_t)n, out_words = (batch + 1) * T * 2 * L * n, dig_words = T * Ld * L * (size_t)n;
    if (overlaps(d_out_qp, out_words, d_in2, in_words) || overlaps(d_out_qp, out_words, d_in_ntt, in_words) || overlaps(d_out_qp, out_words, d_digits, dig_words) ||
        overlaps(d_digits, dig_words, d_in2, in_words) || overlaps(d_digits, dig_words, d_in_ntt, in_words) || overlaps(d_in_ntt, in_words, d_in2, in_words))
        return fail(DPFHE_INVALID_ARGUMENT, what, "buffers must not overlap");
    const size_t key_words = Ld * 2 * L * (size_t)n;
    const int chunks = (n + 511) / 512;
    if ((batch + 1) * T * L * 8 > kMaxGrid || T * Ld * L * (size_t)chunks > kMaxGrid || T * 2 * L * (size_t)chunks > kMaxGrid)
        return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch");
    const LimbConst* lc = c->fold ? c->foldt.lc : c->shoup.lc;
    const u64 p_special = c->p_special;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DPFHE_ON_DEVICE(c, what);
    if (!prepared) {
        // 1. NTT of the inputs on the data limbs (the same tables, seen as an Ld-limb context)
        if (int e = ntt_launch_items(c, false, d_in_ntt, d_in2, T * 2, Ld, s)) return e;
        // 2. digits of every item's c1, lifted to every limb and transformed
        const unsigned lift_grid = (unsigned)(T * Ld * L * (size_t)chunks);
        if (c->fold) hipLaunchKernelGGL((lift_digits_kernel<FoldArith>), dim3(lift_grid), dim3(256), 0, s, d_digits, d_in2 + Ld * (size_t)n, 2 * Ld * (size_t)n, lc, (int)L, n, chunks);
        else hipLaunchKernelGGL((lift_digits_kernel<ShoupArith>), dim3(lift_grid), dim3(256), 0, s, d_digits, d_in2 + Ld * (size_t)n, 2 * Ld * (size_t)n, lc, (int)L, n, chunks);
        if (int e = check_launch("lift_digits kernel launch")) return e;
        if (int e = ntt_launch_items(c, false, d_digits, d_digits, T * Ld, 0, s)) return e;   // (per limb class where the context has them)
        // 3. item block 0: the inputs themselves as P * ct over the extended basis
        const unsigned idg = (unsigned)(T * 2 * L * (size_t)chunks);
        if (c->fold) hipLaunchKernelGGL((lift_qp_kernel<FoldArith>), dim3(idg), dim3(256), 0, s, d_out_qp, d_in_ntt, lc, p_special, (int)L, n, chunks);
        else hipLaunchKernelGGL((lift_qp_kernel<ShoupArith>), dim3(idg), dim3(256), 0, s, d_out_qp, d_in_ntt, lc, p_special, (int)L, n, chunks);
        if (int e = check_launch("lift_qp kernel launch")) return e;
    }
    // 4. the rotations: permuted digit segments x key segments as a stream (kernels_misc.h hoisted_qp_stream_kernel), 64 rotations per launch
    for (size_t first = 0; first < batch; first += kMaxGaloisBatch) {
        const size_t cnt = batch - first < (size_t)kMaxGaloisBatch ? batch - first : (size_t)kMaxGaloisBatch;
        uint64_t* dst = d_out_qp + (1 + first) * T * 2 * L * n;
        {
            QpElts ge{};
            for (size_t i = 0; i < cnt; ++i) ge.v[i] = galois_elts[first + i];
            // (pairs per thread: 2 measured best at 8 tokens - 291 us against 300 with 1 and 329 with 4 -, 1 is 5 % ahead at one token: profiles/r04_ab_baby_steps.txt)
            if (c->fold && Ld >= 1 && Ld <= 6) {   // every segment of the workgroup requested up front (one pair of words per thread)
                const size_t grid1 = qp_stream_grid((int)c->log2n, (int)L, cnt, T, 1);
                if (grid1 > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch");
#define QPU(LD) case LD: hipLaunchKernelGGL((hoisted_qp_upfront_kernel<LD>), dim3((unsigned)grid1), dim3(256), 0, s, dst, d_digits, d_in_ntt, d_keys + first * key_words, key_words, ge, \
                                            (unsigned)cnt, (unsigned)T, p_special, lc, (int)c->log2n); break
                switch (Ld) { QPU(1); QPU(2); QPU(3); QPU(4); QPU(5); QPU(6); }
#undef QPU
                if (int e = check_launch("hoisted_qp kernel launch")) return e;
                continue;
            }
            const size_t grid = qp_stream_grid((int)c->log2n, (int)L, cnt, T);
            if (grid > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch");
            if (c->fold) hipLaunchKernelGGL((hoisted_qp_stream_kernel<FoldArith, kQpPairs>), dim3((unsigned)grid), dim3(256), 0, s, dst, d_digits, d_in_ntt, d_keys + first * key_words, key_words, ge,
                                            (unsigned)cnt, (unsigned)T, p_special, lc, (int)L, (int)c->log2n);
            else hipLaunchKernelGGL((hoisted_qp_stream_kernel<ShoupArith, kQpPairs>), dim3((unsigned)grid), dim3(256), 0, s, dst, d_digits, d_in_ntt, d_keys + first * key_words, key_words, ge,
                                    (unsigned)cnt, (unsigned)T, p_special, lc, (int)L, (int)c->log2n);
        }
        if (int e = check_launch("hoisted_qp kernel launch")) return e;
    }
    return DPFHE_SUCCESS;
}

extern "C" int dpfhe_rotate_hoisted_qp(dpfhe_ctx* c, uint64_t* d_out_qp, const uint64_t* d_in2, size_t n_items, const uint32_t* galois_elts, const uint64_t* d_keys,
                                       uint64_t* d_in_ntt, uint64_t* d_digits, size_t batch, void* stream) {
    return rotate_hoisted_qp_impl(c, d_out_qp, d_in2, n_items, galois_elts, d_keys, d_in_ntt, d_digits, batch, stream, false);
}

extern "C" int dpfhe_ntt_inv_galois(dpfhe_ctx* c, uint64_t* d_out, const uint64_t* d_in, size_t rns_polys_per_elt, const uint32_t* galois_elts, size_t n_elts, void* stream) {
    const char* what = "dpfhe_ntt_inv_galois";
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, what, "null context");
    if (n_elts == 0 || rns_polys_per_elt == 0) return DPFHE_SUCCESS;
    if (!d_out || !d_in || !galois_elts || misaligned(d_out) || misaligned(d_in)) return fail(DPFHE_INVALID_ARGUMENT, what, "null or misaligned buffer");
    const unsigned two_n = 2u << c->log2n;
    for (size_t i = 0; i < n_elts; ++i)
        if (!(galois_elts[i] & 1u) || galois_elts[i] >= two_n) return fail(DPFHE_INVALID_ARGUMENT, what, "galois elements must be odd and < 2N");
    const size_t per_elt = rns_polys_per_elt * c->n_limbs, words = n_elts * per_elt << c->log2n;
    if (d_out != d_in && overlaps(d_out, words, d_in, words)) return fail(DPFHE_INVALID_ARGUMENT, what, "output must be the input buffer or disjoint from it");
    if (per_elt * (n_elts < (size_t)kMaxGaloisBatch ? n_elts : (size_t)kMaxGaloisBatch) > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch");
    DPFHE_ON_DEVICE(c, what);
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (size_t first = 0; first < n_elts; first += kMaxGaloisBatch) {
        const size_t cnt = n_elts - first < (size_t)kMaxGaloisBatch ? n_elts - first : (size_t)kMaxGaloisBatch;
        const size_t off = first * per_elt << c->log2n;
        const int rc = with_policy_or_classes(c, [&](const auto& tb) { return launch_ntt_inv_galois((int)c->log2n, d_out + off, d_in + off, galois_elts + first, cnt, per_elt, tb, s); });
        if (rc) return fail(DPFHE_INVALID_STATE, what, "no single-kernel transform for this log2_n");
        if (int e = check_launch("ntt_inv_galois kernel launch")) return e;
    }
    return DPFHE_SUCCESS;
}

extern "C" int dpfhe_switch_key_qp(dpfhe_ctx* c, uint64_t* d_out_qp, const uint64_t* d_in2, const uint64_t* d_keys, size_t n_keys, size_t group, void* stream) {
    const char* what = "dpfhe_switch_key_qp";
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, what, "null context");
    if (c->n_limbs < 2) return fail(DPFHE_INVALID_STATE, what, "the extended context needs at least one data limb and the special prime");
    const size_t batch = n_keys * group;
    if (batch == 0) return DPFHE_SUCCESS;
    if (!d_out_qp || !d_in2 || !d_keys || misaligned(d_out_qp) || misaligned(d_in2) || misaligned(d_keys)) return fail(DPFHE_INVALID_ARGUMENT, what, "null or misaligned buffer");
    const size_t L = c->n_limbs, Ld = L - 1;
    const int n = 1 << c->log2n;
    if (overlaps(d_out_qp, batch * 2 * L * n, d_in2, batch * 2 * Ld * n)) return fail(DPFHE_INVALID_ARGUMENT, what, "output overlaps the input");
    const size_t blocks = batch * L;
    if ((blocks / 8 + 1) * 8 + 8 * group > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch");
    DPFHE_ON_DEVICE(c, what);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t key_words = Ld * 2 * L * (size_t)n;
    if (c->log2n > kFusedMaxLog2N)   // composed from the batched transform (round 5): the digits of c1, lifted and transformed, times the group's key
        return key_products_composed(c, d_out_qp, d_in2 + Ld * (size_t)n, 2 * Ld * (size_t)n, d_keys, key_words, (unsigned)group, batch, s, what);
    const int rc = relin_launch(c, 4, d_out_qp, d_in2, d_keys, key_words, (unsigned)group, blocks, s);
    if (rc) return fail(DPFHE_INVALID_STATE, what, "no kernel geometry for this log2_n");
    return check_launch("switch_key_qp kernel launch");
}

extern "C" int dpfhe_rescale_bsgs(dpfhe_ctx* c, uint64_t* d_out2, const uint64_t* d_in_qp, const uint64_t* d_addends, size_t n_add, size_t batch, void* stream) {
    const char* what = "dpfhe_rescale_bsgs";
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, what, "null context");
    if (c->n_limbs < 2) return fail(DPFHE_INVALID_STATE, what, "no limb left to drop");
    if (batch == 0) return DPFHE_SUCCESS;
    if (!d_out2 || !d_in_qp || (n_add && !d_addends) || misaligned(d_out2) || misaligned(d_in_qp) || misaligned(d_addends))
        return fail(DPFHE_INVALID_ARGUMENT, what, "null or misaligned buffer");
    const size_t L = c->n_limbs, Ld = L - 1;
    const int n = 1 << c->log2n;
    const int chunks = (n + 511) / 512;
    if (overlaps(d_out2, batch * 2 * Ld * n, d_in_qp, batch * 2 * L * n) || (n_add && overlaps(d_out2, batch * 2 * Ld * n, d_addends, n_add * batch * 2 * Ld * n)))
        return fail(DPFHE_INVALID_ARGUMENT, what, "output overlaps an input");
    const size_t blocks = batch * 2 * Ld * (size_t)chunks;
    if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch");
    DPFHE_ON_DEVICE(c, what);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c->fold) hipLaunchKernelGGL((rescale_bsgs_kernel<FoldArith>), dim3((unsigned)blocks), dim3(256), 0, s, d_out2, d_in_qp, d_addends, n_add, batch, c->foldt.lc, c->d_rescale, (int)L, n, chunks);
    else hipLaunchKernelGGL((rescale_bsgs_kernel<ShoupArith>), dim3((unsigned)blocks), dim3(256), 0, s, d_out2, d_in_qp, d_addends, n_add, batch, c->shoup.lc, c->d_rescale, (int)L, n, chunks);
    return check_launch("rescale_bsgs kernel launch");
}

extern "C" int dpfhe_apply_galois(dpfhe_ctx* c, uint64_t* d_out, const uint64_t* d_in, size_t n_rns_polys, uint32_t galois_elt, void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_apply_galois", "null context");
    const unsigned two_n = 2u << c->log2n;
    if (!(galois_elt & 1u) || galois_elt >= two_n) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_apply_galois", "galois_elt must be odd and < 2N");
    if (n_rns_polys == 0) return DPFHE_SUCCESS;
    if (!d_out || !d_in || misaligned(d_out) || misaligned(d_in) ||
        overlaps(d_out, n_rns_polys * ((size_t)c->n_limbs << c->log2n), d_in, n_rns_polys * ((size_t)c->n_limbs << c->log2n)))
        return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_apply_galois", "null, misaligned or aliased buffer");
    const size_t npolys = n_rns_polys * c->n_limbs;
    if (npolys > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_apply_galois", "batch too large for one launch");
    DPFHE_ON_DEVICE(c, "dpfhe_apply_galois");
    const unsigned inv = galois_inverse(galois_elt, two_n);
    const LimbConst* lc = c->fold ? c->foldt.lc : c->shoup.lc;
    hipLaunchKernelGGL(galois_kernel, dim3((unsigned)npolys), dim3(256), 0, static_cast<hipStream_t>(stream), d_out, d_in, lc, (int)c->n_limbs,
                       1 << c->log2n, inv);
    return check_launch("galois kernel launch");
}

// ------------------------------------------------------------------------------------------------
extern "C" int dpfhe_matvec_plain(dpfhe_ctx* c, uint64_t* d_y, const uint64_t* d_W, const uint64_t* d_x, size_t rows, size_t cols,
                                  void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_matvec_plain", "null context");
    if (rows == 0) return DPFHE_SUCCESS;
    if (cols == 0) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_matvec_plain", "cols must be > 0");
    if (!d_y || !d_W || !d_x || misaligned(d_y) || misaligned(d_W) || misaligned(d_x))
        return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_matvec_plain", "null or misaligned buffer");
    const int n = 1 << c->log2n;
    constexpr int RT = 4;
    DPFHE_ON_DEVICE(c, "dpfhe_matvec_plain");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c->fold) {   // column accumulators split at bit 30 (kernels_misc.h matvec_fold_kernel): 4 multiply-adds per term
        constexpr int WPT = kMatvecWpt2;
        const int chunks = (n + 256 * WPT - 1) / (256 * WPT);
        const size_t slabs = c->n_limbs * (size_t)chunks, rtiles = (rows + RT - 1) / RT;
        const size_t blocks = ((slabs + 7) / 8) * 8 * rtiles;
        if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_matvec_plain", "too many rows for one launch");
        // a W beyond the 256 MiB Infinity Cache is a read-once stream (every tile goes to exactly one workgroup here): non-temporal loads
        const bool ntw = rows * cols * ((size_t)c->n_limbs << c->log2n) * sizeof(u64) > ((size_t)256 << 20);
        // (the branch-free FULL form of the kernel, which the multi-right-hand-side product takes, measured SLOWER here: 1412 against 1258 us on configs[2] -
        // this launch streams 6 GiB of W from HBM with two right-hand-side polynomials per workgroup and lives on memory-level parallelism, not on issue slots)
        if (ntw) hipLaunchKernelGGL((matvec_fold_kernel<RT, 2, WPT, true>), dim3((unsigned)blocks), dim3(256), 0, s, d_y, d_W, d_x, c->foldt.lc, (int)c->n_limbs, n, chunks, rows,
                                    cols, (size_t)2, 1u, (unsigned)(rtiles * slabs));
        else hipLaunchKernelGGL((matvec_fold_kernel<RT, 2, WPT>), dim3((unsigned)blocks), dim3(256), 0, s, d_y, d_W, d_x, c->foldt.lc, (int)c->n_limbs, n, chunks, rows, cols,
                                (size_t)2, 1u, (unsigned)(rtiles * slabs));
        return check_launch("matvec kernel launch");
    }
    const int chunks = (n + 511) / 512;
    const size_t slabs = c->n_limbs * (size_t)chunks;
    const size_t blocks = ((slabs + 7) / 8) * 8 * ((rows + RT - 1) / RT);   // block ids laid out per XCD: kernels_misc.h matvec_kernel
    if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_matvec_plain", "too many rows for one launch");
    hipLaunchKernelGGL((matvec_kernel<ShoupArith, RT>), dim3((unsigned)blocks), dim3(256), 0, s, d_y, d_W, d_x, c->shoup.lc, (int)c->n_limbs, n, chunks, rows, cols);
    return check_launch("matvec kernel launch");
}

extern "C" int dpfhe_matvec_plain_multi(dpfhe_ctx* c, uint64_t* d_y, const uint64_t* d_W, const uint64_t* d_x, size_t rows, size_t cols, size_t n_rhs,
                                        void* stream) {
    const char* what = "dpfhe_matvec_plain_multi";
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, what, "null context");
    if (rows == 0 || n_rhs == 0) return DPFHE_SUCCESS;
    if (cols == 0) return fail(DPFHE_INVALID_ARGUMENT, what, "cols must be > 0");
    if (!d_y || !d_W || !d_x || misaligned(d_y) || misaligned(d_W) || misaligned(d_x)) return fail(DPFHE_INVALID_ARGUMENT, what, "null or misaligned buffer");
    if (n_rhs == 1) return dpfhe_matvec_plain(c, d_y, d_W, d_x, rows, cols, stream);
    const int n = 1 << c->log2n;
    const int chunks = (n + 511) / 512;
    const size_t poly = (size_t)c->n_limbs << c->log2n;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DPFHE_ON_DEVICE(c, what);
    // 2 right-hand sides x 4 rows per workgroup (32 128-bit accumulators per thread); all groups of 2 in ONE launch whose block ids put
    // the groups of a W tile on the same XCD (kernels_misc.h), an odd last right-hand side in a second launch.  x / y are
    // [cols | rows][n_rhs][2][L][N]: a group is a strided slice, so the kernels take the full stride.  Measured on the 32 x 32 x
    // 1024-diagonal matvec of a packed GPT-2 layer, per token: 86 us single; 8 tokens: 75 us with one launch per group (W re-read from
    // HBM by every group), see MEASUREMENTS.md for the XCD-grouped launch.
    const size_t pairs = n_rhs / 2;
    if (c->fold) {   // split-at-bit-30 column accumulators (kernels_misc.h matvec_fold_kernel), same grouping and block-id layout
#define MVF_LAUNCH(RT, C, WPT, GROUPS, XS, YS)                                                                                                          \
    {                                                                                                                                                    \
        const int chunks = (n + 256 * (WPT) - 1) / (256 * (WPT));                                                                                        \
        const size_t slabs = c->n_limbs * (size_t)chunks, rtiles = (rows + RT - 1) / RT, tiles = rtiles * slabs;                                         \
        const size_t blocks = ((slabs + 7) / 8) * 8 * rtiles * (GROUPS);                                                                                 \
        if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, what, "too many rows for one launch");                                             \
        /* whole row tiles and whole periods (a packed layer's products): the branch-free form, see kernels_misc.h */                                    \
        if (rows % (RT) == 0 && cols % FoldArith::kDot30Period == 0)                                                                \
            hipLaunchKernelGGL((matvec_fold_kernel<RT, C, WPT, false, kMatvecWd, true>), dim3((unsigned)blocks), dim3(256), 0, s, YS, d_W, XS, c->foldt.lc, (int)c->n_limbs, n, chunks, rows, \
                               cols, n_rhs * 2, (unsigned)(GROUPS), (unsigned)tiles);                                                                    \
        else                                                                                                                                             \
            hipLaunchKernelGGL((matvec_fold_kernel<RT, C, WPT, false, kMatvecWd>), dim3((unsigned)blocks), dim3(256), 0, s, YS, d_W, XS, c->foldt.lc, (int)c->n_limbs, n, chunks, rows, cols, \
                               n_rhs * 2, (unsigned)(GROUPS), (unsigned)tiles);                                                                          \
    }
        if (pairs) {
            MVF_LAUNCH(kMatvecRt4, 4, kMatvecWpt4, pairs, d_x, d_y)
            if (int e = check_launch("matvec_multi kernel launch")) return e;
        }
        if (n_rhs & 1) {
            const uint64_t* xs = d_x + (n_rhs - 1) * 2 * poly;
            uint64_t* ys = d_y + (n_rhs - 1) * 2 * poly;
            MVF_LAUNCH(4, 2, kMatvecWpt2, 1, xs, ys)
            if (int e = check_launch("matvec_multi kernel launch")) return e;
        }
#undef MVF_LAUNCH
        return DPFHE_SUCCESS;
    }
#define MV_LAUNCH(ARITH, RT, C, LC, GROUPS, XS, YS)                                                                                                      \
    {                                                                                                                                                    \
        const size_t slabs = c->n_limbs * (size_t)chunks, rtiles = (rows + RT - 1) / RT, tiles = rtiles * slabs;                                         \
        const size_t blocks = ((slabs + 7) / 8) * 8 * rtiles * (GROUPS);                                                                                 \
        if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, what, "too many rows for one launch");                                             \
        hipLaunchKernelGGL((matvec_multi_kernel<ARITH, RT, C>), dim3((unsigned)blocks), dim3(256), 0, s, YS, d_W, XS, LC, (int)c->n_limbs, n, chunks, rows, cols, \
                           n_rhs * 2, (unsigned)(GROUPS), (unsigned)tiles);                                                                              \
    }
    if (pairs) {
        MV_LAUNCH(ShoupArith, 4, 4, c->shoup.lc, pairs, d_x, d_y)
        if (int e = check_launch("matvec_multi kernel launch")) return e;
    }
    if (n_rhs & 1) {
        const uint64_t* xs = d_x + (n_rhs - 1) * 2 * poly;
        uint64_t* ys = d_y + (n_rhs - 1) * 2 * poly;
        MV_LAUNCH(ShoupArith, 4, 2, c->shoup.lc, 1, xs, ys)
        if (int e = check_launch("matvec_multi kernel launch")) return e;
    }
#undef MV_LAUNCH
    return DPFHE_SUCCESS;
}

extern "C" int dpfhe_matvec_scalar(dpfhe_ctx* c, uint64_t* d_y, const uint64_t* d_w, const uint64_t* d_x, size_t rows, size_t cols,
                                   void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_matvec_scalar", "null context");
    if (rows == 0) return DPFHE_SUCCESS;
    if (cols == 0) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_matvec_scalar", "cols must be > 0");
    if (!d_y || !d_w || !d_x || misaligned(d_y) || misaligned(d_x) || (reinterpret_cast<uintptr_t>(d_w) & 7))
        return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_matvec_scalar", "null or misaligned buffer");
    const int n = 1 << c->log2n;
    const int chunks = (n + 511) / 512;
    constexpr int RT = 8;
    const size_t blocks = ((rows + RT - 1) / RT) * c->n_limbs * (size_t)chunks;
    if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_matvec_scalar", "too many rows for one launch");
    DPFHE_ON_DEVICE(c, "dpfhe_matvec_scalar");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c->fold) hipLaunchKernelGGL((matvec_scalar_kernel<FoldArith, RT>), dim3((unsigned)blocks), dim3(256), 0, s, d_y, d_w, d_x, c->foldt.lc, (int)c->n_limbs, n, chunks, rows, cols);
    else hipLaunchKernelGGL((matvec_scalar_kernel<ShoupArith, RT>), dim3((unsigned)blocks), dim3(256), 0, s, d_y, d_w, d_x, c->shoup.lc, (int)c->n_limbs, n, chunks, rows, cols);
    return check_launch("matvec_scalar kernel launch");
}

extern "C" int dpfhe_reduce_sum(dpfhe_ctx* c, uint64_t* d_out, const uint64_t* d_in, size_t count, size_t components, void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_reduce_sum", "null context");
    if (components == 0 || count == 0) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_reduce_sum", "count and components must be > 0");
    if (!d_out || !d_in || misaligned(d_out) || misaligned(d_in)) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_reduce_sum", "null or misaligned buffer");
    const size_t blocks = components * c->n_limbs;
    if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_reduce_sum", "too many components");
    DPFHE_ON_DEVICE(c, "dpfhe_reduce_sum");
    const int n = 1 << c->log2n;
    const LimbConst* lc = c->fold ? c->foldt.lc : c->shoup.lc;
    const int chunks = (n + 511) / 512;
    const size_t words_per_item = components * c->n_limbs * (size_t)n;
    // batch splits (their partial sums are combined with atomics): 15 for the large batches of the sharded multiply; a short batch (the
    // 32 rotated terms of a packed layer) keeps at least 8 items per split, so that every work item has 4 independent loads in flight
    const size_t want = count / 8 ? count / 8 : 1;
    const unsigned splits = (unsigned)(want < (size_t)kReduceSplits ? want : (size_t)kReduceSplits);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (blocks * (size_t)chunks * splits > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_reduce_sum", "too many components for one launch");
    HIP_TRY(hipMemsetAsync(d_out, 0, words_per_item * sizeof(u64), s));   // after every validation: a rejected call leaves d_out untouched
    const unsigned poly_chunks = (unsigned)(blocks * chunks);
    // two workgroups per CU walk the work items: as fast as an uncapped launch when alone (537 vs 551 us for 8192 x 3
    // components at N=4096) and 2 % faster for the multiply it overlaps with in bench.py
    unsigned grid = poly_chunks * splits;
    if (count > 512 && grid > 2u * (unsigned)c->n_cu) grid = 2u * (unsigned)c->n_cu;   // short batches are latency-bound: no cap
    hipLaunchKernelGGL(reduce_partial_kernel, dim3(grid), dim3(256), 0, s, d_out, d_in, lc, (int)c->n_limbs, n, chunks, count, words_per_item,
                       poly_chunks, splits);
    if (int e = check_launch("reduce_sum partial kernel launch")) return e;
    hipLaunchKernelGGL(reduce_final_kernel, dim3((unsigned)(blocks * chunks)), dim3(256), 0, s, d_out, lc, (int)c->n_limbs, n, chunks);
    return check_launch("reduce_sum kernel launch");
}

// words that are sums of at most 15 canonical residues (< 15 q < 2^64) -> canonical, in place: the one pass after an all-reduce of partial ciphertexts
extern "C" int dpfhe_canonicalize_sum(dpfhe_ctx* c, uint64_t* d_io, size_t n_rns_polys, void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_canonicalize_sum", "null context");
    if (n_rns_polys == 0) return DPFHE_SUCCESS;
    if (!d_io || misaligned(d_io)) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_canonicalize_sum", "null or misaligned buffer");
    const int n = 1 << c->log2n, chunks = (n + 511) / 512;
    const size_t blocks = n_rns_polys * c->n_limbs * (size_t)chunks;
    if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_canonicalize_sum", "batch too large for one launch");
    DPFHE_ON_DEVICE(c, "dpfhe_canonicalize_sum");
    hipLaunchKernelGGL(reduce_final_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_io, c->fold ? c->foldt.lc : c->shoup.lc, (int)c->n_limbs, n, chunks);
    return check_launch("canonicalize_sum kernel launch");
}

// ------------------------------------------------------------------------------------------------
// exact base extension / scale-and-round between limb ranges of one context (kernels_misc.h base_extend_kernel)
static int base_extend_common(dpfhe_ctx* c, int mode, uint64_t* d_out, size_t out_stride_limbs, const uint64_t* d_in, size_t in_stride_limbs, uint32_t src0, uint32_t ns,
                              uint32_t dst0, uint32_t nd, uint64_t multiplier, size_t n_polys, void* stream, const char* what) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, what, "null context");
    if (n_polys == 0) return DPFHE_SUCCESS;
    const uint32_t L = c->n_limbs;
    if (ns == 0 || ns > (uint32_t)kBxMaxSrc || nd == 0 || nd > (uint32_t)kBxMaxDst || src0 > L || ns > L - src0 || dst0 > L || nd > L - dst0)   /* no 32-bit wrap-around */
        return fail(DPFHE_INVALID_ARGUMENT, what, "1..10 source limbs and 1..20 destination limbs inside the context");
    if (mode == 1 && !(dst0 >= src0 + ns || dst0 + nd <= src0)) return fail(DPFHE_INVALID_ARGUMENT, what, "the dropped limbs and the kept limbs must be disjoint");
    if (!d_out || !d_in || misaligned(d_out) || misaligned(d_in) || out_stride_limbs < nd || in_stride_limbs < ns) return fail(DPFHE_INVALID_ARGUMENT, what, "null or misaligned buffer, or an item stride shorter than its limbs");
    const size_t n = (size_t)1 << c->log2n;
    const int chunks = (int)((n + 511) / 512);
    if (n_polys * (size_t)chunks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, what, "batch too large for one launch");
    {   // mode 1: d_in points at the first dropped limb of item 0 inside a [n_polys][L][N] buffer that starts src0 limbs earlier
        const uint64_t* in_lo = mode == 1 ? d_in - (size_t)src0 * n : d_in;
        const size_t in_words = mode == 1 ? n_polys * (size_t)L * n : ((n_polys - 1) * in_stride_limbs + ns) * n;
        if (overlaps(d_out, ((n_polys - 1) * out_stride_limbs + nd) * n, in_lo, in_words)) return fail(DPFHE_INVALID_ARGUMENT, what, "output overlaps the input");
    }
    // host constants (a handful of modular inverses; moduli read back from the context's limb constants would need a copy - they are kept on the host)
    const std::vector<uint64_t>& q = c->moduli;
    BaseExtArgs a{};
    a.n_src = (int)ns; a.n_dst = (int)nd;
    for (uint32_t i = 0; i < ns; ++i) a.src_limb[i] = (int)(src0 + i);
    for (uint32_t j = 0; j < nd; ++j) a.dst_limb[j] = (int)(dst0 + j);
    for (uint32_t i = 0; i < ns; ++i)
        for (uint32_t k = i + 1; k < ns; ++k) {
            if (q[src0 + i] == q[src0 + k]) return fail(DPFHE_INVALID_ARGUMENT, what, "source limbs must be distinct primes");
            a.inv[i][k] = h_powmod(q[src0 + i] % q[src0 + k], q[src0 + k] - 2, q[src0 + k]);
        }
    {   // mixed-radix digits of floor(Qs / 2): (Qs - 1) / 2 since Qs is odd; digit k of Qs - 1 is q_k - 1, halve with borrow from the top
        unsigned __int128 carry = 0;   // remainder (0 or 1) carried down, in units of the current radix
        for (int k = (int)ns - 1; k >= 0; --k) {
            const unsigned __int128 cur = carry * q[src0 + k] + (q[src0 + k] - 1);
            a.half[k] = (uint64_t)(cur / 2);
            carry = cur & 1;
        }
    }
    for (uint32_t j = 0; j < nd; ++j) {
        const uint64_t pj = q[dst0 + j];
        uint64_t Qm = 1;
        for (uint32_t i = 0; i < ns; ++i) { a.q_mod[i][j] = q[src0 + i] % pj; Qm = h_mulmod(Qm, a.q_mod[i][j], pj); }
        a.Q_mod[j] = Qm;
        if (mode == 1) {
            if (Qm == 0) return fail(DPFHE_INVALID_ARGUMENT, what, "a kept limb divides the dropped modulus");
            a.Q_inv[j] = h_powmod(Qm, pj - 2, pj);
            a.mul_dst[j] = multiplier % pj;
        }
    }
    for (uint32_t i = 0; i < ns; ++i) a.mul_src[i] = multiplier % q[src0 + i];
    DPFHE_ON_DEVICE(c, what);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)(n_polys * (size_t)chunks);
    const size_t in_dst_off = mode == 1 ? ((size_t)dst0 - (size_t)src0) * n : 0;   // d_in points at the first DROPPED limb of item 0 (may wrap below it: size_t arithmetic, same pointer sum)
    const LimbConst* lc = c->fold ? c->foldt.lc : c->shoup.lc;
    const int rc = c->fold ? launch_base_extend<FoldArith>(mode, d_out, out_stride_limbs * n, d_in, in_stride_limbs * n, in_dst_off, a, lc, (int)n, chunks, grid, s)
                           : launch_base_extend<ShoupArith>(mode, d_out, out_stride_limbs * n, d_in, in_stride_limbs * n, in_dst_off, a, lc, (int)n, chunks, grid, s);
    if (rc) return fail(DPFHE_INVALID_STATE, what, "no kernel for this number of source limbs");
    return check_launch("base_extend kernel launch");
}
extern "C" int dpfhe_base_extend(dpfhe_ctx* c, uint64_t* d_out, size_t out_stride_limbs, const uint64_t* d_in, size_t in_stride_limbs, uint32_t src_limb0, uint32_t n_src,
                                 uint32_t dst_limb0, uint32_t n_dst, size_t n_polys, void* stream) {
    return base_extend_common(c, 0, d_out, out_stride_limbs, d_in, in_stride_limbs, src_limb0, n_src, dst_limb0, n_dst, 1, n_polys, stream, "dpfhe_base_extend");
}
extern "C" int dpfhe_scale_round(dpfhe_ctx* c, uint64_t* d_out, size_t out_stride_limbs, const uint64_t* d_in, uint32_t drop_limb0, uint32_t n_drop, uint32_t keep_limb0,
                                 uint32_t n_keep, uint64_t multiplier, size_t n_polys, void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_scale_round", "null context");
    if (multiplier == 0) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_scale_round", "multiplier must be > 0");
    // d_in: [n_polys][L][N], ALL limbs of the context; the kernel reads the dropped limbs at drop_limb0 and the kept ones at keep_limb0
    const size_t n = (size_t)1 << c->log2n;
    return base_extend_common(c, 1, d_out, out_stride_limbs, d_in ? d_in + (size_t)drop_limb0 * n : nullptr, c->n_limbs, drop_limb0, n_drop, keep_limb0, n_keep, multiplier, n_polys, stream,
                              "dpfhe_scale_round");
}

extern "C" int dpfhe_copy(dpfhe_ctx* c, uint64_t* d_dst, const uint64_t* d_src, size_t n_words, void* stream) {
    if (!c) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_copy", "null context");
    if (n_words == 0) return DPFHE_SUCCESS;
    if (!d_dst || !d_src || misaligned(d_dst) || misaligned(d_src) || (n_words & 1)) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_copy", "null or misaligned buffer, or an odd word count");
    if (overlaps(d_dst, n_words, d_src, n_words)) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_copy", "buffers overlap");
    const size_t n_vec = n_words / 2, blocks = (n_vec + 2047) / 2048;
    if (blocks > kMaxGrid) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_copy", "too many words for one launch");
    DPFHE_ON_DEVICE(c, "dpfhe_copy");
    // streams that cannot live in the 256 MiB Infinity Cache (source + destination) go around it
    if (n_words * 16 > (size_t)256 << 20)
        hipLaunchKernelGGL(copy_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<U64x2*>(d_dst),
                           reinterpret_cast<const U64x2*>(d_src), n_vec);
    else
        hipLaunchKernelGGL(copy_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<U64x2*>(d_dst),
                           reinterpret_cast<const U64x2*>(d_src), n_vec);
    return check_launch("copy kernel launch");
}

// ------------------------------------------------------------------------------------------------
// (e) RCCL all-gather.  librccl is opened lazily so that single-GPU users never depend on it.
// ------------------------------------------------------------------------------------------------
namespace {
struct NcclUniqueId { char internal[128]; };
typedef void* ncclComm_t;
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        x.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!x.h) x.h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!x.h) return x;
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.h, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.h, "ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.h, "ncclCommDestroy"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.h, "ncclAllGather"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.h, "ncclAllReduce"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.h, "ncclGetErrorString"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather && x.AllReduce;
        return x;
    }();
    return r;
}
const int kNcclUint64 = 5;  // ncclUint64 in rccl.h's ncclDataType_t
int rccl_fail(const char* what, int rc) {
    Rccl& r = rccl();
    return fail(DPFHE_RUNTIME_ERROR, what, r.GetErrorString ? r.GetErrorString(rc) : "rccl error");
}
}  // namespace

struct dpfhe_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" int dpfhe_comm_unique_id(uint8_t out_id[128]) {
    if (!out_id) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_comm_unique_id", "null argument");
    Rccl& r = rccl();
    if (!r.ok) return fail(DPFHE_RUNTIME_ERROR, "dpfhe_comm_unique_id", "librccl.so.1 not loadable");
    NcclUniqueId id;
    int rc = r.GetUniqueId(&id);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(out_id, id.internal, 128);
    return DPFHE_SUCCESS;
}

extern "C" int dpfhe_comm_create(dpfhe_comm** out, const uint8_t id[128], int rank, int world_size, int device_id) {
    if (!out || !id || world_size < 1 || rank < 0 || rank >= world_size) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_comm_create", "bad argument");
    Rccl& r = rccl();
    if (!r.ok) return fail(DPFHE_RUNTIME_ERROR, "dpfhe_comm_create", "librccl.so.1 not loadable");
    HIP_TRY(hipSetDevice(device_id));
    dpfhe_comm* c = new (std::nothrow) dpfhe_comm;
    if (!c) return fail(DPFHE_OUT_OF_MEMORY, "dpfhe_comm_create", "host allocation");
    c->rank = rank; c->world = world_size; c->device = device_id;
    NcclUniqueId uid;
    std::memcpy(uid.internal, id, 128);
    int rc = r.CommInitRank(&c->comm, world_size, uid, rank);
    if (rc) { delete c; return rccl_fail("ncclCommInitRank", rc); }
    *out = c;
    return DPFHE_SUCCESS;
}

extern "C" int dpfhe_comm_destroy(dpfhe_comm* c) {
    if (!c) return DPFHE_SUCCESS;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
    return DPFHE_SUCCESS;
}

extern "C" int dpfhe_comm_allgather(dpfhe_comm* c, uint64_t* d_recv, const uint64_t* d_send, size_t words_per_rank, void* stream) {
    if (!c || !d_recv || !d_send) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_comm_allgather", "null argument");
    if (words_per_rank == 0) return DPFHE_SUCCESS;
    DeviceGuard guard(c->device);
    if (guard.err != hipSuccess) return fail(DPFHE_DEVICE_ERROR, "dpfhe_comm_allgather", hipGetErrorString(guard.err));
    int rc = rccl().AllGather(d_send, d_recv, words_per_rank, kNcclUint64, c->comm, static_cast<hipStream_t>(stream));
    if (rc) return rccl_fail("ncclAllGather", rc);
    return DPFHE_SUCCESS;
}

// SURVEY.md section 8(e)'s alternative to the all-gather, ready for the first multi-GPU lease: ncclAllReduce(ncclUint64, ncclSum) of the ranks' partial
// ciphertexts in place, then ONE mod-q pass - safe because world_size * q < 2^64 for world_size <= 15 (q < 2^60).  Same words as all-gather + local sum.
extern "C" int dpfhe_comm_allreduce_sum(dpfhe_comm* c, dpfhe_ctx* ctx, uint64_t* d_io, size_t n_rns_polys, void* stream) {
    if (!c || !ctx || !d_io) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_comm_allreduce_sum", "null argument");
    if (c->world > 15) return fail(DPFHE_INVALID_STATE, "dpfhe_comm_allreduce_sum", "the lazy sum of more than 15 residues below 2^60 does not fit 64 bits");
    if (c->device != ctx->device) return fail(DPFHE_INVALID_ARGUMENT, "dpfhe_comm_allreduce_sum", "communicator and context live on different devices");
    if (n_rns_polys == 0) return DPFHE_SUCCESS;
    {
        DeviceGuard guard(c->device);
        if (guard.err != hipSuccess) return fail(DPFHE_DEVICE_ERROR, "dpfhe_comm_allreduce_sum", hipGetErrorString(guard.err));
        const size_t words = n_rns_polys * ctx->n_limbs << ctx->log2n;
        const int ncclSum = 0;   // rccl.h ncclRedOp_t
        int rc = rccl().AllReduce(d_io, d_io, words, kNcclUint64, ncclSum, c->comm, static_cast<hipStream_t>(stream));
        if (rc) return rccl_fail("ncclAllReduce", rc);
    }
    return dpfhe_canonicalize_sum(ctx, d_io, n_rns_polys, stream);
}

// ------------------------------------------------------------------------------------------------
extern "C" const char* dpfhe_strerror(int code) {
    switch (code) {
        case DPFHE_SUCCESS: return "success";
        case DPFHE_OUT_OF_MEMORY: return "out of memory";
        case DPFHE_DEVICE_ERROR: return "device error";
        case DPFHE_INVALID_ARGUMENT: return "invalid argument";
        case DPFHE_INVALID_STATE: return "invalid state";
        case DPFHE_RUNTIME_ERROR: return "runtime error";
        default: return "unknown error";
    }
}
extern "C" const char* dpfhe_last_error(void) { return g_last_error.c_str(); }
