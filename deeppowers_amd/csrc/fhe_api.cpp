// fhe_api.cpp - deeppowers::fhe facade over the C ABI (include/dpfhe.h).  Plain C++17 (g++); the only
// HIP it touches is the runtime API for buffer ownership, like the reference's HAL device
// (/root/reference/src/core/hal/cuda/cuda_device.cpp:9-16 turns runtime errors into exceptions).
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <sys/random.h>

#include <cerrno>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <istream>
#include <ostream>

#include "deeppowers/fhe.hpp"
#include "dpfhe.h"

namespace deeppowers {
namespace fhe {

namespace {
[[noreturn]] void raise(int code, const char* what) {
    std::string msg = std::string(what) + ": " + dpfhe_last_error();
    if (msg.size() <= std::string(what).size() + 2) msg = std::string(what) + ": " + dpfhe_strerror(code);
    throw Exception(static_cast<ErrorCode>(code), msg);
}
void check(int code, const char* what) {
    if (code != DPFHE_SUCCESS) raise(code, what);
}
void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw Exception(e == hipErrorOutOfMemory ? ErrorCode::OUT_OF_MEMORY : ErrorCode::DEVICE_ERROR, std::string(what) + ": " + hipGetErrorString(e));
}
// the 20 largest primes below 2^60 that are 1 mod 2^14, largest first (deeppowers_amd/params.py ntt_primes(13, 20) generates the same list): {q, smallest
// primitive 8192-th root (N = 4096; 0 = not tabulated), smallest primitive 16384-th root (N = 8192)}
constexpr size_t kChainPrimes = 20;
const uint64_t kPrimes60[kChainPrimes][3] = {
    {1152921504606830593ull, 116777451583545ull, 25959043411404ull}, {1152921504606748673ull, 271802498405390ull, 100406242475323ull},
    {1152921504606683137ull, 134367042585739ull, 45474351589225ull}, {1152921504606601217ull, 276147373136904ull, 92707844590835ull},
    {1152921504606584833ull, 317490233586139ull, 23981819781494ull}, {1152921504606109697ull, 279138086580908ull, 253932030982881ull},
    {1152921504605962241ull, 0ull, 64984728504994ull}, {1152921504605913089ull, 0ull, 27694533958986ull},
    {1152921504605847553ull, 0ull, 105031879276246ull}, {1152921504605618177ull, 0ull, 157253107066567ull},
    {1152921504604979201ull, 0ull, 76334773615457ull}, {1152921504604766209ull, 0ull, 14852029848402ull},
    {1152921504604635137ull, 0ull, 93806574463579ull}, {1152921504602505217ull, 0ull, 217691047434989ull},
    {1152921504601980929ull, 0ull, 112510666220977ull}, {1152921504601915393ull, 0ull, 12114078003698ull},
    {1152921504601784321ull, 0ull, 4580624056246ull}, {1152921504600309761ull, 0ull, 135029094688496ull},
    {1152921504600260609ull, 0ull, 120773065591640ull}, {1152921504600145921ull, 0ull, 1663825873988ull}};
// the 8 largest primes below 2^60 that are 1 mod 2^15 (deeppowers_amd/params.py ntt_primes(14, 8)): {q, smallest primitive 32768-th root}
constexpr size_t kChainPrimes14 = 8;
const uint64_t kPrimes60N14[kChainPrimes14][2] = {
    {1152921504606748673ull, 62213374832584ull}, {1152921504606683137ull, 212089012217363ull}, {1152921504606584833ull, 92166579128688ull},
    {1152921504605962241ull, 74756755228070ull}, {1152921504604979201ull, 52069629205452ull}, {1152921504600260609ull, 27543819356734ull},
    {1152921504599080961ull, 92056553354496ull}, {1152921504598720513ull, 89492317149395ull}};
}  // namespace

FheParams FheParams::drop_last_limb() const {
    if (moduli.size() < 2) throw Exception(ErrorCode::INVALID_STATE, "drop_last_limb: no limb left to drop");
    FheParams p{log2_n, moduli, psi};
    p.moduli.pop_back(); p.psi.pop_back();
    return p;
}
FheParams FheParams::config1() { return FheParams{10, {1073707009ull}, {169871ull}}; }
FheParams FheParams::n4096_l4() {
    FheParams p{12, {}, {}};
    for (int i = 0; i < 4; ++i) { p.moduli.push_back(kPrimes60[i][0]); p.psi.push_back(kPrimes60[i][1]); }
    return p;
}
FheParams FheParams::n8192_l6() { return n8192(6); }
FheParams FheParams::n8192(size_t n_limbs) {
    if (n_limbs == 0 || n_limbs > kChainPrimes) throw Exception(ErrorCode::INVALID_ARGUMENT, "FheParams::n8192: 1..20 limbs");
    FheParams p{13, {}, {}};
    for (size_t i = 0; i < n_limbs; ++i) { p.moduli.push_back(kPrimes60[i][0]); p.psi.push_back(kPrimes60[i][2]); }
    return p;
}

FheParams FheParams::n16384(size_t n_limbs) {
    if (n_limbs == 0 || n_limbs > kChainPrimes14) throw Exception(ErrorCode::INVALID_ARGUMENT, "FheParams::n16384: 1..8 limbs");
    FheParams p{14, {}, {}};
    for (size_t i = 0; i < n_limbs; ++i) { p.moduli.push_back(kPrimes60N14[i][0]); p.psi.push_back(kPrimes60N14[i][1]); }
    return p;
}

// ---- Context ---------------------------------------------------------------------------------------
class Context::Impl {
public:
    FheParams params;
    int device_id = 0;
    dpfhe_ctx* h = nullptr;
};

Context::Context(const FheParams& params, int device_id) : impl_(new Impl) {
    impl_->params = params;
    impl_->device_id = device_id;
    if (params.moduli.empty() || params.moduli.size() != params.psi.size())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "FheParams: moduli and psi must be non-empty and of equal length");
    check(dpfhe_ctx_create(&impl_->h, params.log2_n, (uint32_t)params.moduli.size(), params.moduli.data(), params.psi.data(), device_id), "dpfhe_ctx_create");
}
Context::~Context() {
    if (impl_ && impl_->h) dpfhe_ctx_destroy(impl_->h);
}
const FheParams& Context::params() const { return impl_->params; }
int Context::device_id() const { return impl_->device_id; }
bool Context::uses_fold() const { return dpfhe_ctx_uses_fold(impl_->h) != 0; }
int Context::limb_class(uint32_t limb) const { return dpfhe_ctx_limb_class(impl_->h, limb); }
void Context::release_scratch(void* stream, bool all_streams) { check(dpfhe_ctx_release_scratch(impl_->h, stream, all_streams ? DPFHE_SCRATCH_ALL : 0), "dpfhe_ctx_release_scratch"); }
size_t Context::scratch_bytes() const { return dpfhe_ctx_scratch_bytes(impl_->h); }
void* Context::handle() const { return impl_->h; }
Context::TuneInfo Context::tune_info() const {
    dpfhe_tune_info t{};
    check(dpfhe_ctx_tune_info(impl_->h, &t), "dpfhe_ctx_tune_info");
    static const char* const src[] = {"default", "?", "dpfhe_ctx_autotune", "forced", "cached dpfhe_ctx_autotune of this shape"};   // include/dpfhe.h DPFHE_TUNE_*
    TuneInfo r;
    r.chosen = dpfhe_ct_mul_variant_name(t.chosen);
    r.source = (t.source >= 0 && t.source < 5) ? src[t.source] : "?";
    r.probe_pairs = t.probe_pairs;
    r.probe_reps = t.probe_reps;
    for (int v = 0; v < t.n_variants && v < 8; ++v)
        if (t.probe_us[v] >= 0) r.probe_us.emplace_back(dpfhe_ct_mul_variant_name(v), t.probe_us[v]);
    return r;
}
Context::TuneInfo Context::autotune(PolyBuffer& scratch, unsigned reps) {
    check(dpfhe_ctx_autotune(impl_->h, scratch.data(), scratch.words(), reps, nullptr), "dpfhe_ctx_autotune");
    return tune_info();
}
void Context::synchronize() const {
    hip_check(hipSetDevice(impl_->device_id), "hipSetDevice");
    hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
}

void Context::set_scratch_limit(size_t mib) { check(dpfhe_ctx_set_scratch_limit(impl_->h, mib), "dpfhe_ctx_set_scratch_limit"); }

// ---- PolyBuffer -----------------------------------------------------------------------------------------
class PolyBuffer::Impl {
public:
    const Context* ctx = nullptr;
    uint64_t* d = nullptr;
    size_t batch = 0, comps = 0, words = 0;
    bool ntt = false;
    int device_id = 0;
    ~Impl() {
        if (d) (void)hipFree(d);
    }
};

PolyBuffer::PolyBuffer(const Context& ctx, size_t batch, size_t components, bool is_ntt) : impl_(new Impl) {
    if (batch == 0 || components == 0) throw Exception(ErrorCode::INVALID_ARGUMENT, "PolyBuffer: batch and components must be > 0");
    impl_->ctx = &ctx;
    impl_->batch = batch; impl_->comps = components; impl_->ntt = is_ntt; impl_->device_id = ctx.device_id();
    impl_->words = batch * components * ctx.params().n_limbs() * ctx.params().n();
    hip_check(hipSetDevice(impl_->device_id), "hipSetDevice");
    void* p = nullptr;
    hip_check(hipMalloc(&p, impl_->words * sizeof(uint64_t)), "hipMalloc");
    impl_->d = static_cast<uint64_t*>(p);
}
PolyBuffer::~PolyBuffer() = default;
PolyBuffer::PolyBuffer(PolyBuffer&&) noexcept = default;
PolyBuffer& PolyBuffer::operator=(PolyBuffer&&) noexcept = default;
uint64_t* PolyBuffer::data() { return impl_->d; }
const uint64_t* PolyBuffer::data() const { return impl_->d; }
size_t PolyBuffer::batch() const { return impl_->batch; }
size_t PolyBuffer::size() const { return impl_->comps; }
size_t PolyBuffer::words() const { return impl_->words; }
bool PolyBuffer::is_ntt() const { return impl_->ntt; }
void PolyBuffer::set_ntt(bool v) { impl_->ntt = v; }
void PolyBuffer::copy_from_host(const uint64_t* src) {
    if (!src) throw Exception(ErrorCode::INVALID_ARGUMENT, "copy_from_host: null source");
    hip_check(hipMemcpy(impl_->d, src, impl_->words * sizeof(uint64_t), hipMemcpyHostToDevice), "hipMemcpy H2D");
}
void PolyBuffer::copy_to_host(uint64_t* dst) const {
    if (!dst) throw Exception(ErrorCode::INVALID_ARGUMENT, "copy_to_host: null destination");
    hip_check(hipMemcpy(dst, impl_->d, impl_->words * sizeof(uint64_t), hipMemcpyDeviceToHost), "hipMemcpy D2H");
}

namespace {
const char kMagic[8] = {'D', 'P', 'F', 'H', 'E', 'v', '1', 0};
struct WireHeader {   // all little-endian; x86-64 / gfx950 hosts are little-endian
    char magic[8];
    uint32_t log2_n, n_limbs;
    uint64_t batch, components;
    uint32_t is_ntt, reserved;
};
}  // namespace

void PolyBuffer::save(std::ostream& os) const {
    const FheParams& p = impl_->ctx->params();
    WireHeader h{};
    std::memcpy(h.magic, kMagic, 8);
    h.log2_n = p.log2_n; h.n_limbs = (uint32_t)p.n_limbs(); h.batch = impl_->batch; h.components = impl_->comps;
    h.is_ntt = impl_->ntt ? 1u : 0u; h.reserved = 0;
    std::vector<uint64_t> host(impl_->words);
    copy_to_host(host.data());
    os.write(reinterpret_cast<const char*>(&h), sizeof h);
    os.write(reinterpret_cast<const char*>(p.moduli.data()), (std::streamsize)(p.n_limbs() * sizeof(uint64_t)));
    os.write(reinterpret_cast<const char*>(host.data()), (std::streamsize)(host.size() * sizeof(uint64_t)));
    if (!os) throw Exception(ErrorCode::RUNTIME_ERROR, "save: stream write failed");
}

void PolyBuffer::load(std::istream& is) {
    const FheParams& p = impl_->ctx->params();
    WireHeader h{};
    is.read(reinterpret_cast<char*>(&h), sizeof h);
    if (!is || std::memcmp(h.magic, kMagic, 8) != 0) throw Exception(ErrorCode::INVALID_ARGUMENT, "load: not a DPFHEv1 stream");
    if (h.log2_n != p.log2_n || h.n_limbs != p.n_limbs() || h.batch != impl_->batch || h.components != impl_->comps)
        throw Exception(ErrorCode::INVALID_ARGUMENT, "load: header does not match this buffer (log2_n / limbs / batch / components)");
    std::vector<uint64_t> moduli(h.n_limbs);
    is.read(reinterpret_cast<char*>(moduli.data()), (std::streamsize)(moduli.size() * sizeof(uint64_t)));
    if (!is || moduli != p.moduli) throw Exception(ErrorCode::INVALID_ARGUMENT, "load: moduli differ from this context");
    std::vector<uint64_t> host(impl_->words);
    is.read(reinterpret_cast<char*>(host.data()), (std::streamsize)(host.size() * sizeof(uint64_t)));
    if (!is) throw Exception(ErrorCode::INVALID_ARGUMENT, "load: truncated stream");
    const size_t n = p.n(), L = p.n_limbs();
    for (size_t i = 0; i < host.size(); ++i)
        if (host[i] >= p.moduli[(i / n) % L]) throw Exception(ErrorCode::INVALID_ARGUMENT, "load: non-canonical residue in the payload");
    copy_from_host(host.data());
    impl_->ntt = h.is_ntt != 0;
}

Ciphertext::Ciphertext(const Context& ctx, size_t size, size_t batch, bool is_ntt) : PolyBuffer(ctx, batch, size, is_ntt) {
    if (size != 2 && size != 3) throw Exception(ErrorCode::INVALID_ARGUMENT, "Ciphertext: size must be 2 or 3");
}

// ---- ScalarMatrix -----------------------------------------------------------------------------------------------------------
class ScalarMatrix::Impl {
public:
    const Context* ctx = nullptr;
    size_t rows = 0, cols = 0;
    uint64_t* d = nullptr;
    ~Impl() {
        if (d) (void)hipFree(d);
    }
};
ScalarMatrix::ScalarMatrix(const Context& ctx, size_t rows, size_t cols) : impl_(new Impl) {
    if (rows == 0 || cols == 0) throw Exception(ErrorCode::INVALID_ARGUMENT, "ScalarMatrix: rows and cols must be > 0");
    impl_->ctx = &ctx; impl_->rows = rows; impl_->cols = cols;
    hip_check(hipSetDevice(ctx.device_id()), "hipSetDevice");
    void* p = nullptr;
    hip_check(hipMalloc(&p, rows * cols * ctx.params().n_limbs() * sizeof(uint64_t)), "hipMalloc");
    impl_->d = static_cast<uint64_t*>(p);
}
ScalarMatrix::~ScalarMatrix() = default;
size_t ScalarMatrix::rows() const { return impl_->rows; }
size_t ScalarMatrix::cols() const { return impl_->cols; }
const uint64_t* ScalarMatrix::data() const { return impl_->d; }
void ScalarMatrix::set(const int64_t* w) {
    if (!w) throw Exception(ErrorCode::INVALID_ARGUMENT, "ScalarMatrix::set: null weights");
    const FheParams& p = impl_->ctx->params();
    const size_t L = p.n_limbs(), count = impl_->rows * impl_->cols;
    std::vector<uint64_t> host(count * L);
    for (size_t i = 0; i < count; ++i)
        for (size_t l = 0; l < L; ++l) {
            const uint64_t q = p.moduli[l];
            const uint64_t m = (uint64_t)(w[i] < 0 ? -(w[i] + 1) : w[i]) % q;            // |w| (two's-complement safe), mod q
            host[i * L + l] = w[i] >= 0 ? m : (q - 1 - m);                                 // -(m+1) = q - 1 - m
        }
    hip_check(hipMemcpy(impl_->d, host.data(), host.size() * sizeof(uint64_t), hipMemcpyHostToDevice), "hipMemcpy H2D");
}

RelinKeys::RelinKeys(const Context& ctx) : PolyBuffer(ctx, ctx.params().n_limbs(), 2, /*is_ntt=*/true) {}
GaloisKeys::GaloisKeys(const Context& ctx, uint32_t galois_elt) : PolyBuffer(ctx, ctx.params().n_limbs(), 2, /*is_ntt=*/true), galois_elt_(galois_elt) {
    if (!(galois_elt & 1u) || galois_elt >= 2 * ctx.params().n()) throw Exception(ErrorCode::INVALID_ARGUMENT, "GaloisKeys: galois_elt must be odd and < 2N");
}

// ---- Evaluator --------------------------------------------------------------------------------------------
class Evaluator::Impl {
public:
    const Context* ctx = nullptr;
    dpfhe_ctx* h() const { return static_cast<dpfhe_ctx*>(ctx->handle()); }
    size_t npolys(const PolyBuffer& b) const { return b.batch() * b.size(); }
    static void same(const PolyBuffer& a, const PolyBuffer& b, const char* what) {
        if (a.batch() != b.batch() || a.size() != b.size()) throw Exception(ErrorCode::INVALID_ARGUMENT, std::string(what) + ": operand shapes differ");
        if (a.is_ntt() != b.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, std::string(what) + ": operands are in different domains");
    }
};

Evaluator::Evaluator(const Context& ctx) : impl_(new Impl) { impl_->ctx = &ctx; }
Evaluator::~Evaluator() = default;

void Evaluator::transform_to_ntt_inplace(PolyBuffer& x, Stream* s) const {
    if (x.is_ntt()) return;
    check(dpfhe_ntt_fwd(impl_->h(), x.data(), impl_->npolys(x), s), "dpfhe_ntt_fwd");
    x.set_ntt(true);
}
void Evaluator::transform_from_ntt_inplace(PolyBuffer& x, Stream* s) const {
    if (!x.is_ntt()) return;
    check(dpfhe_ntt_inv(impl_->h(), x.data(), impl_->npolys(x), s), "dpfhe_ntt_inv");
    x.set_ntt(false);
}
void Evaluator::add(const PolyBuffer& a, const PolyBuffer& b, PolyBuffer& out, Stream* s) const {
    Impl::same(a, b, "add");
    if (out.words() != a.words()) throw Exception(ErrorCode::INVALID_ARGUMENT, "add: output shape differs");
    check(dpfhe_add(impl_->h(), out.data(), a.data(), b.data(), impl_->npolys(a), s), "dpfhe_add");
    out.set_ntt(a.is_ntt());
}
void Evaluator::sub(const PolyBuffer& a, const PolyBuffer& b, PolyBuffer& out, Stream* s) const {
    Impl::same(a, b, "sub");
    if (out.words() != a.words()) throw Exception(ErrorCode::INVALID_ARGUMENT, "sub: output shape differs");
    check(dpfhe_sub(impl_->h(), out.data(), a.data(), b.data(), impl_->npolys(a), s), "dpfhe_sub");
    out.set_ntt(a.is_ntt());
}
void Evaluator::negate(const PolyBuffer& a, PolyBuffer& out, Stream* s) const {
    if (out.words() != a.words()) throw Exception(ErrorCode::INVALID_ARGUMENT, "negate: output shape differs");
    check(dpfhe_negate(impl_->h(), out.data(), a.data(), impl_->npolys(a), s), "dpfhe_negate");
    out.set_ntt(a.is_ntt());
}
void Evaluator::dyadic_multiply(const PolyBuffer& a, const PolyBuffer& b, PolyBuffer& out, Stream* s) const {
    Impl::same(a, b, "dyadic_multiply");
    if (out.words() != a.words()) throw Exception(ErrorCode::INVALID_ARGUMENT, "dyadic_multiply: output shape differs");
    check(dpfhe_dyadic_mul(impl_->h(), out.data(), a.data(), b.data(), impl_->npolys(a), s), "dpfhe_dyadic_mul");
    out.set_ntt(a.is_ntt());
}
void Evaluator::dyadic_multiply_add(const PolyBuffer& a, const PolyBuffer& b, PolyBuffer& acc, Stream* s) const {
    Impl::same(a, b, "dyadic_multiply_add");
    Impl::same(a, acc, "dyadic_multiply_add");
    check(dpfhe_dyadic_mul_add(impl_->h(), acc.data(), a.data(), b.data(), impl_->npolys(a), s), "dpfhe_dyadic_mul_add");
}
void Evaluator::multiply(const Ciphertext& a, const Ciphertext& b, Ciphertext& out, Stream* s) const {
    Impl::same(a, b, "multiply");
    if (a.size() != 2) throw Exception(ErrorCode::INVALID_ARGUMENT, "multiply: inputs must be 2-component ciphertexts");
    if (out.size() != 3 || out.batch() != a.batch()) throw Exception(ErrorCode::INVALID_ARGUMENT, "multiply: output must be a 3-component ciphertext of the same batch");
    const uint32_t flags = (a.is_ntt() ? (uint32_t)DPFHE_IN_NTT : 0u) | (out.is_ntt() ? (uint32_t)DPFHE_OUT_NTT : 0u);
    check(dpfhe_ct_mul(impl_->h(), out.data(), a.data(), b.data(), a.batch(), flags, s), "dpfhe_ct_mul");
}
void Evaluator::relinearize(const Ciphertext& in3, const RelinKeys& keys, Ciphertext& out2, Stream* s) const {
    if (in3.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "relinearize: input must be in the coefficient domain");
    if (in3.size() != 3 || out2.size() != 2 || out2.batch() != in3.batch())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "relinearize: 3-component input, 2-component output of the same batch");
    check(dpfhe_relinearize(impl_->h(), out2.data(), in3.data(), keys.data(), in3.batch(), s), "dpfhe_relinearize");
    out2.set_ntt(false);
}
void Evaluator::rescale(const Ciphertext& in, Ciphertext& out, Stream* s) const {
    if (in.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "rescale: input must be in the coefficient domain");
    const size_t L = impl_->ctx->params().n_limbs(), n = impl_->ctx->params().n();
    if (L < 2) throw Exception(ErrorCode::INVALID_STATE, "rescale: no limb left to drop");
    if (out.size() != in.size() || out.batch() != in.batch() || out.words() != in.batch() * in.size() * (L - 1) * n)
        throw Exception(ErrorCode::INVALID_ARGUMENT, "rescale: output must live on the next-level context (L-1 limbs), same size and batch");
    check(dpfhe_rescale(impl_->h(), out.data(), in.data(), in.batch() * in.size(), s), "dpfhe_rescale");
    out.set_ntt(false);
}
// ---- ExactMultiplier ------------------------------------------------------------------------------------------------------------
class ExactMultiplier::Impl {
public:
    const Context* work = nullptr;
    const Context* level = nullptr;
    uint64_t t = 0;
    size_t ll = 0, L = 0, n = 0;
    std::unique_ptr<Ciphertext> A, B, T;     // operands and tensor product on all work limbs
    std::unique_ptr<PolyBuffer> W;           // scaled product on the workspace limbs: [batch * 3][L - ll][N], kept as a 1-component buffer on `work`-sized storage
    size_t cap = 0;
    void ensure(size_t batch) {
        if (batch <= cap) return;
        A.reset(new Ciphertext(*work, 2, batch));
        B.reset(new Ciphertext(*work, 2, batch));
        T.reset(new Ciphertext(*work, 3, batch));
        W.reset(new PolyBuffer(*work, batch, 3, false));   // 3 L N words per item: room for 3 (L - ll) N
        cap = batch;
    }
};
ExactMultiplier::ExactMultiplier(const Context& work_ctx, const Context& level_ctx, uint64_t plain_modulus) : impl_(new Impl) {
    const FheParams &pw = work_ctx.params(), &pl = level_ctx.params();
    impl_->work = &work_ctx; impl_->level = &level_ctx; impl_->t = plain_modulus;
    impl_->ll = pl.n_limbs(); impl_->L = pw.n_limbs(); impl_->n = pw.n();
    if (pl.log2_n != pw.log2_n || impl_->ll == 0 || impl_->ll >= impl_->L || impl_->ll > (work_ctx.uses_fold() ? 9u : 8u) || impl_->L > 20 || impl_->L - impl_->ll > (work_ctx.uses_fold() ? 10u : 8u) || work_ctx.device_id() != level_ctx.device_id() || plain_modulus < 2)
        throw Exception(ErrorCode::INVALID_ARGUMENT, "ExactMultiplier: the level context must hold the first 1..9 limbs (1..8 with generic primes) of the work context, which has at most 20 and at most 10 (generic primes: 8) beyond the level (same ring degree and device)");
    for (size_t i = 0; i < impl_->ll; ++i)
        if (pl.moduli[i] != pw.moduli[i]) throw Exception(ErrorCode::INVALID_ARGUMENT, "ExactMultiplier: the level's moduli must be the first moduli of the work context");
    double lq = 0, lQ = 0, lW = 0;
    for (size_t i = 0; i < impl_->L; ++i) { const double b = std::log2((double)pw.moduli[i]); lQ += b; if (i < impl_->ll) lq += b; else lW += b; }
    const double lnt = (double)pw.log2_n + std::log2((double)plain_modulus);
    if (lQ - 1 <= lnt + 2 * lq + 1 || lW - 1 <= lnt + lq + 2)
        throw Exception(ErrorCode::INVALID_STATE, "ExactMultiplier: the work context is too small for the integer tensor product of this level (needs Q > 2 N t q^2)");
}
ExactMultiplier::~ExactMultiplier() = default;
void ExactMultiplier::multiply(const Ciphertext& a, const Ciphertext& b, Ciphertext& out3, Stream* s) {
    Impl& I = *impl_;
    const size_t batch = a.batch(), lvl_words = I.ll * I.n;
    if (a.is_ntt() || b.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "ExactMultiplier::multiply: operands must be in the coefficient domain");
    if (a.size() != 2 || b.size() != 2 || out3.size() != 3 || b.batch() != batch || out3.batch() != batch || a.words() != batch * 2 * lvl_words || out3.words() != batch * 3 * lvl_words)
        throw Exception(ErrorCode::INVALID_ARGUMENT, "ExactMultiplier::multiply: 2-component operands and a 3-component output of one batch on the level context");
    I.ensure(batch);
    dpfhe_ctx* h = static_cast<dpfhe_ctx*>(I.work->handle());
    const bool square = a.data() == b.data();
    check(dpfhe_base_extend(h, I.A->data(), I.L, a.data(), I.ll, 0, (uint32_t)I.ll, 0, (uint32_t)I.L, batch * 2, s), "dpfhe_base_extend");
    if (!square) check(dpfhe_base_extend(h, I.B->data(), I.L, b.data(), I.ll, 0, (uint32_t)I.ll, 0, (uint32_t)I.L, batch * 2, s), "dpfhe_base_extend");
    check(dpfhe_ct_mul(h, I.T->data(), I.A->data(), square ? I.A->data() : I.B->data(), batch, 0, s), "dpfhe_ct_mul");
    check(dpfhe_scale_round(h, I.W->data(), I.L - I.ll, I.T->data(), 0, (uint32_t)I.ll, (uint32_t)I.ll, (uint32_t)(I.L - I.ll), I.t, batch * 3, s), "dpfhe_scale_round");
    check(dpfhe_base_extend(h, out3.data(), I.ll, I.W->data(), I.L - I.ll, (uint32_t)I.ll, (uint32_t)(I.L - I.ll), 0, (uint32_t)I.ll, batch * 3, s), "dpfhe_base_extend");
    out3.set_ntt(false);
}

void Evaluator::apply_galois(const Ciphertext& in2, const GaloisKeys& keys, Ciphertext& out2, Stream* s) const {
    if (in2.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "apply_galois: input must be in the coefficient domain");
    if (in2.size() != 2 || out2.size() != 2 || out2.batch() != in2.batch())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "apply_galois: 2-component input and output of the same batch");
    Ciphertext rotated(*impl_->ctx, 2, in2.batch());
    check(dpfhe_apply_galois(impl_->h(), rotated.data(), in2.data(), in2.batch() * 2, keys.galois_elt(), s), "dpfhe_apply_galois");
    check(dpfhe_switch_key(impl_->h(), out2.data(), rotated.data(), keys.data(), in2.batch(), s), "dpfhe_switch_key");
    hip_check(hipStreamSynchronize(static_cast<hipStream_t>(s)), "hipStreamSynchronize");  // `rotated` is freed on return
    out2.set_ntt(false);
}
void Evaluator::multiply_plain(const Ciphertext& a, const Plaintext& p, Ciphertext& out, Stream* s) const {
    if (!a.is_ntt() || !p.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "multiply_plain: operands must be in the NTT domain");
    if (p.batch() != 1 || out.words() != a.words()) throw Exception(ErrorCode::INVALID_ARGUMENT, "multiply_plain: one plaintext, output shaped like the input");
    check(dpfhe_multiply_plain(impl_->h(), out.data(), a.data(), p.data(), a.batch() * a.size(), s), "dpfhe_multiply_plain");
    out.set_ntt(true);
}
void Evaluator::matvec_plain(const Plaintext& W, const Ciphertext& x, Ciphertext& y, Stream* s) const {
    if (!W.is_ntt() || !x.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "matvec_plain: operands must be in the NTT domain");
    const size_t cols = x.batch(), rows = y.batch();
    if (x.size() != 2 || y.size() != 2 || cols == 0 || W.batch() != rows * cols) throw Exception(ErrorCode::INVALID_ARGUMENT, "matvec_plain: W batch must be rows*cols, x/y 2-component");
    check(dpfhe_matvec_plain(impl_->h(), y.data(), W.data(), x.data(), rows, cols, s), "dpfhe_matvec_plain");
    y.set_ntt(true);
}
void Evaluator::matvec_plain_multi(const Plaintext& W, const Ciphertext& x, Ciphertext& y, size_t n_rhs, Stream* s) const {
    if (!W.is_ntt() || !x.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "matvec_plain_multi needs NTT-domain operands");
    if (n_rhs == 0 || x.size() != 2 || y.size() != 2 || x.batch() % n_rhs || y.batch() % n_rhs)
        throw Exception(ErrorCode::INVALID_ARGUMENT, "matvec_plain_multi: x [cols][n_rhs], y [rows][n_rhs] 2-component items");
    const size_t cols = x.batch() / n_rhs, rows = y.batch() / n_rhs;
    if (W.batch() != rows * cols) throw Exception(ErrorCode::INVALID_ARGUMENT, "matvec_plain_multi: W must hold rows * cols plaintexts");
    check(dpfhe_matvec_plain_multi(impl_->h(), y.data(), W.data(), x.data(), rows, cols, n_rhs, s), "dpfhe_matvec_plain_multi");
    y.set_ntt(true);
}

void Evaluator::matvec_scalar(const ScalarMatrix& W, const Ciphertext& x, Ciphertext& y, Stream* s) const {
    if (x.size() != 2 || y.size() != 2 || x.batch() != W.cols() || y.batch() != W.rows())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "matvec_scalar: x batch = cols, y batch = rows, 2-component ciphertexts");
    check(dpfhe_matvec_scalar(impl_->h(), y.data(), W.data(), x.data(), W.rows(), W.cols(), s), "dpfhe_matvec_scalar");
    y.set_ntt(x.is_ntt());
}
void Evaluator::reduce_sum(const PolyBuffer& in, PolyBuffer& out, Stream* s) const {
    if (out.batch() != 1 || out.size() != in.size()) throw Exception(ErrorCode::INVALID_ARGUMENT, "reduce_sum: output must be one item of the same size");
    check(dpfhe_reduce_sum(impl_->h(), out.data(), in.data(), in.batch(), in.size(), s), "dpfhe_reduce_sum");
    out.set_ntt(in.is_ntt());
}

// =====================================================================================================================
// (e) Communicator: dpfhe_comm_* behind the facade
// =====================================================================================================================
class Communicator::Impl {
public:
    dpfhe_comm* h = nullptr;
    int rank = 0, world = 1;
};
std::vector<uint8_t> Communicator::unique_id() {
    std::vector<uint8_t> id(128);
    check(dpfhe_comm_unique_id(id.data()), "dpfhe_comm_unique_id");
    return id;
}
Communicator::Communicator(const std::vector<uint8_t>& id, int rank, int world_size, int device_id) : impl_(new Impl) {
    if (id.size() != 128) throw Exception(ErrorCode::INVALID_ARGUMENT, "Communicator: the id is 128 bytes");
    check(dpfhe_comm_create(&impl_->h, id.data(), rank, world_size, device_id), "dpfhe_comm_create");
    impl_->rank = rank; impl_->world = world_size;
}
Communicator::~Communicator() { if (impl_ && impl_->h) dpfhe_comm_destroy(impl_->h); }
int Communicator::rank() const { return impl_->rank; }
int Communicator::world_size() const { return impl_->world; }
void Communicator::all_gather(const PolyBuffer& send, PolyBuffer& recv, Stream* stream) const {
    if (recv.batch() != send.batch() * (size_t)impl_->world || recv.size() != send.size())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "all_gather: recv must hold world_size x send items of the same size");
    check(dpfhe_comm_allgather(impl_->h, recv.data(), send.data(), send.words(), stream), "dpfhe_comm_allgather");
    recv.set_ntt(send.is_ntt());
}

// =====================================================================================================================
// N2: keys, encryption, decryption (host side)
// =====================================================================================================================
namespace {
typedef unsigned __int128 u128;

// Randomness of keys and encryptions.  Default: ChaCha20 keyed with 48 bytes from the operating system's CSPRNG
// (getrandom(2), /dev/urandom as fallback) - uniform values by rejection sampling, ternary secrets, centred-binomial errors
// (eta = 21: sigma = 3.24, |e| <= 21).  The TestSeed constructors of the public classes switch to SplitMix64 so that tests
// and examples are reproducible; that generator is invertible with 64 bits of state and must never protect real data.
struct Sampler {
    bool secure = true;
    uint64_t sm = 0;          // SplitMix64 state (testing)
    uint32_t st[16] = {};     // ChaCha20 state: constants | key | counter | nonce
    uint32_t blk[16] = {};
    int used = 16;            // 32-bit words of blk already handed out

    Sampler() { key_from_os(); }
    explicit Sampler(TestSeed seed) : secure(false), sm(seed.value) {}

    void key_from_os() {
        unsigned char buf[48];
        size_t got = 0;
        while (got < sizeof(buf)) {
            const ssize_t r = getrandom(buf + got, sizeof(buf) - got, 0);
            if (r > 0) { got += (size_t)r; continue; }
            if (r < 0 && errno == EINTR) continue;
            break;
        }
        if (got < sizeof(buf)) {   // kernels without getrandom(2)
            FILE* f = std::fopen("/dev/urandom", "rb");
            if (f) { got += std::fread(buf + got, 1, sizeof(buf) - got, f); std::fclose(f); }
        }
        if (got < sizeof(buf)) throw Exception(ErrorCode::RUNTIME_ERROR, "no operating-system randomness available (getrandom, /dev/urandom)");
        static const uint32_t sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};   // "expand 32-byte k"
        std::memcpy(st, sigma, 16);
        std::memcpy(st + 4, buf, 32);          // key
        st[12] = 0; st[13] = 0;                // 64-bit block counter
        std::memcpy(st + 14, buf + 32, 8);     // nonce
        // the remaining 8 bytes perturb the counter start so that equal (key, nonce) - impossible in practice - still differ
        uint32_t c[2]; std::memcpy(c, buf + 40, 8); st[12] = c[0]; st[13] = c[1];
        volatile unsigned char* wipe = buf;
        for (size_t i = 0; i < sizeof(buf); ++i) wipe[i] = 0;
    }
    static uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
    void refill() {   // one ChaCha20 block (RFC 8439 section 2.3), 64-bit counter
        uint32_t x[16];
        std::memcpy(x, st, 64);
        auto qr = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        };
        for (int i = 0; i < 10; ++i) {
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; ++i) blk[i] = x[i] + st[i];
        if (++st[12] == 0) ++st[13];
        used = 0;
    }
    uint64_t next() {
        if (!secure) {   // SplitMix64 (same generator as the synthetic-data spec)
            sm += 0x9E3779B97F4A7C15ull;
            uint64_t z = sm;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        }
        if (used > 14) refill();
        const uint64_t v = (uint64_t)blk[used] | ((uint64_t)blk[used + 1] << 32);
        used += 2;
        return v;
    }
    // uniform in [0, bound): rejection sampling on the smallest covering power of two (no modulo bias)
    uint64_t below(uint64_t bound) {
        if (bound <= 1) return 0;
        const uint64_t mask = ~0ull >> __builtin_clzll(bound - 1);
        for (;;) {
            const uint64_t v = next() & mask;
            if (v < bound) return v;
        }
    }
    int ternary() { return (int)below(3) - 1; }
    // centred binomial, eta = 21: popcount(21 bits) - popcount(21 bits); variance 10.5 (sigma 3.24)
    int64_t error() {
        const uint64_t v = next();
        return (int64_t)__builtin_popcountll(v & 0x1fffffull) - (int64_t)__builtin_popcountll((v >> 21) & 0x1fffffull);
    }
};

uint64_t lift_signed(int64_t v, uint64_t q) { return v >= 0 ? (uint64_t)v % q : q - ((uint64_t)(-v) % q == 0 ? q : (uint64_t)(-v) % q); }
uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) {
    uint64_t r = 1;
    for (b %= q; e; e >>= 1) { if (e & 1) r = (uint64_t)((u128)r * b % q); b = (uint64_t)((u128)b * b % q); }
    return r;
}

// little-endian multiword unsigned integers, just enough for CRT composition of <= 1024 limbs
typedef std::vector<uint64_t> Big;
void big_mul_small(Big& a, uint64_t m) {
    u128 carry = 0;
    for (auto& w : a) { u128 t = (u128)w * m + carry; w = (uint64_t)t; carry = t >> 64; }
    if (carry) a.push_back((uint64_t)carry);
}
void big_add(Big& a, const Big& b) {
    if (a.size() < b.size()) a.resize(b.size(), 0);
    u128 carry = 0;
    for (size_t i = 0; i < a.size(); ++i) { u128 t = (u128)a[i] + (i < b.size() ? b[i] : 0) + carry; a[i] = (uint64_t)t; carry = t >> 64; }
    if (carry) a.push_back((uint64_t)carry);
}
int big_cmp(const Big& a, const Big& b) {
    size_t n = a.size() > b.size() ? a.size() : b.size();
    for (size_t i = n; i-- > 0;) {
        uint64_t x = i < a.size() ? a[i] : 0, y = i < b.size() ? b[i] : 0;
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}
Big big_sub(const Big& a, const Big& b) {  // a >= b
    Big r(a.size(), 0);
    uint64_t borrow = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        u128 t = (u128)a[i] - (i < b.size() ? b[i] : 0) - borrow;
        r[i] = (uint64_t)t; borrow = (uint64_t)(t >> 64) & 1;
    }
    return r;
}
uint64_t big_divmod_small(Big& a, uint64_t m) {  // a = floor(a / m), returns a mod m
    u128 rem = 0;
    for (size_t i = a.size(); i-- > 0;) { u128 cur = (rem << 64) | a[i]; a[i] = (uint64_t)(cur / m); rem = cur % m; }
    while (a.size() > 1 && a.back() == 0) a.pop_back();
    return (uint64_t)rem;
}
uint64_t big_mod_small(const Big& a, uint64_t m) { Big t(a); return big_divmod_small(t, m); }
void big_shr_round(Big& a, unsigned sh) {  // a = floor((a + 2^(sh-1)) / 2^sh)
    if (sh) {
        Big half((sh - 1) / 64 + 1, 0);
        half[(sh - 1) / 64] = 1ull << ((sh - 1) % 64);
        big_add(a, half);
    }
    const size_t ws = sh / 64, bs = sh % 64;
    Big r(a.size() > ws ? a.size() - ws : 1, 0);
    for (size_t i = 0; i + ws < a.size(); ++i) {
        r[i] = a[i + ws] >> bs;
        if (bs && i + ws + 1 < a.size()) r[i] |= a[i + ws + 1] << (64 - bs);
    }
    a = r;
}
}  // namespace

// ---- SecretKey ----------------------------------------------------------------------------------------------------------
class SecretKey::Impl {
public:
    const Context* ctx = nullptr;
    std::vector<int8_t> s;
    std::unique_ptr<PolyBuffer> s_hat, s2_hat;
};

namespace {
std::vector<int8_t> sample_ternary(size_t n, Sampler rng) {
    std::vector<int8_t> s(n);
    for (auto& v : s) v = (int8_t)rng.ternary();
    return s;
}
}  // namespace

SecretKey::SecretKey(const Context& ctx) : SecretKey(ctx, sample_ternary(ctx.params().n(), Sampler())) {}
SecretKey::SecretKey(const Context& ctx, TestSeed seed) : SecretKey(ctx, sample_ternary(ctx.params().n(), Sampler(seed))) {}

SecretKey::SecretKey(const Context& ctx, const std::vector<int8_t>& coeffs) : impl_(new Impl) {
    impl_->ctx = &ctx;
    const FheParams& p = ctx.params();
    const size_t n = p.n(), L = p.n_limbs();
    if (coeffs.size() != n) throw Exception(ErrorCode::INVALID_ARGUMENT, "SecretKey: need N ternary coefficients");
    for (int8_t v : coeffs)
        if (v < -1 || v > 1) throw Exception(ErrorCode::INVALID_ARGUMENT, "SecretKey: coefficients must be in {-1, 0, 1}");
    impl_->s = coeffs;
    std::vector<uint64_t> host(L * n);
    for (size_t l = 0; l < L; ++l)
        for (size_t k = 0; k < n; ++k) host[l * n + k] = lift_signed(impl_->s[k], p.moduli[l]);
    impl_->s_hat.reset(new PolyBuffer(ctx, 1, 1, false));
    impl_->s2_hat.reset(new PolyBuffer(ctx, 1, 1, true));
    impl_->s_hat->copy_from_host(host.data());
    Evaluator ev(ctx);
    ev.transform_to_ntt_inplace(*impl_->s_hat);
    ev.dyadic_multiply(*impl_->s_hat, *impl_->s_hat, *impl_->s2_hat);
    ctx.synchronize();
}
SecretKey::~SecretKey() = default;
const std::vector<int8_t>& SecretKey::coefficients() const { return impl_->s; }
const uint64_t* SecretKey::ntt() const { return impl_->s_hat->data(); }
const uint64_t* SecretKey::ntt_squared() const { return impl_->s2_hat->data(); }

// ---- KeyGenerator ----------------------------------------------------------------------------------------------------------
class KeyGenerator::Impl {
public:
    const Context* ctx = nullptr;
    std::unique_ptr<SecretKey> sk;
    Sampler rng;
};

KeyGenerator::KeyGenerator(const Context& ctx) : impl_(new Impl) {
    impl_->ctx = &ctx;
    impl_->sk.reset(new SecretKey(ctx));
}
KeyGenerator::KeyGenerator(const Context& ctx, TestSeed seed) : impl_(new Impl) {
    impl_->ctx = &ctx;
    impl_->sk.reset(new SecretKey(ctx, seed));
    impl_->rng = Sampler(TestSeed{seed.value ^ 0xD1B54A32D192ED03ull});
}
KeyGenerator::~KeyGenerator() = default;
const SecretKey& KeyGenerator::secret_key() const { return *impl_->sk; }

namespace {
// shared by relinearisation and Galois keys: key_j = (-(a_j s) + e_j + g_j * target, a_j), everything in the NTT domain
template <class Rng>
void make_switch_key(const Context& ctx, const SecretKey& sk, Rng& rng, const uint64_t* d_target_ntt, PolyBuffer& out,
                     size_t n_digits = 0 /* 0 = all limbs */, const uint64_t* d_target_scaled_ntt = nullptr);
}  // namespace

void KeyGenerator::create_galois_keys(GaloisKeys& out) {
    const Context& ctx = *impl_->ctx;
    const FheParams& p = ctx.params();
    const size_t n = p.n(), L = p.n_limbs();
    const std::vector<int8_t>& s = impl_->sk->coefficients();
    std::vector<int64_t> sg(n, 0);
    for (size_t i = 0; i < n; ++i) {   // sigma_g(s): coefficient i -> index i g mod 2N, negated past N
        const size_t idx = (i * (size_t)out.galois_elt()) & (2 * n - 1);
        if (idx < n) sg[idx] = s[i]; else sg[idx - n] = -s[i];
    }
    std::vector<uint64_t> host(L * n);
    for (size_t l = 0; l < L; ++l)
        for (size_t k = 0; k < n; ++k) host[l * n + k] = lift_signed(sg[k], p.moduli[l]);
    PolyBuffer target(ctx, 1, 1, false);
    target.copy_from_host(host.data());
    Evaluator ev(ctx);
    ev.transform_to_ntt_inplace(target);
    ctx.synchronize();
    make_switch_key(ctx, *impl_->sk, impl_->rng, target.data(), out);
}

void KeyGenerator::create_public_key(PublicKey& out) {
    const Context& ctx = *impl_->ctx;
    const FheParams& p = ctx.params();
    const size_t n = p.n(), L = p.n_limbs(), poly = L * n;
    dpfhe_ctx* h = static_cast<dpfhe_ctx*>(ctx.handle());
    std::vector<uint64_t> ha(poly), he(poly);
    for (size_t l = 0; l < L; ++l)
        for (size_t k = 0; k < n; ++k) ha[l * n + k] = impl_->rng.below(p.moduli[l]);   // uniform: any domain
    for (size_t k = 0; k < n; ++k) {
        const int64_t ev = impl_->rng.error();
        for (size_t l = 0; l < L; ++l) he[l * n + k] = lift_signed(ev, p.moduli[l]);
    }
    PolyBuffer a(ctx, 1, 1, true), e(ctx, 1, 1, false), t(ctx, 1, 1, true);
    a.copy_from_host(ha.data());
    e.copy_from_host(he.data());
    check(dpfhe_ntt_fwd(h, e.data(), 1, nullptr), "dpfhe_ntt_fwd");
    check(dpfhe_dyadic_mul(h, t.data(), a.data(), impl_->sk->ntt(), 1, nullptr), "dpfhe_dyadic_mul");   // a s
    check(dpfhe_sub(h, out.data(), e.data(), t.data(), 1, nullptr), "dpfhe_sub");                       // pk0 = e - a s
    hip_check(hipMemcpyAsync(out.data() + poly, a.data(), poly * sizeof(uint64_t), hipMemcpyDeviceToDevice, nullptr), "hipMemcpyAsync");
    ctx.synchronize();
    out.set_ntt(true);
}

void KeyGenerator::create_relin_keys(RelinKeys& out) { make_switch_key(*impl_->ctx, *impl_->sk, impl_->rng, impl_->sk->ntt_squared(), out); }

namespace {
// n_digits < L (hybrid): only the data limbs are digits and d_target_ntt must already carry the factor P.
template <class Rng>
void make_switch_key(const Context& ctx, const SecretKey& sk_ref, Rng& rng_ref, const uint64_t* d_target_ntt, PolyBuffer& out, size_t n_digits,
                     const uint64_t*) {
    struct { const SecretKey* sk; Rng* rng; } impl_s{&sk_ref, &rng_ref};
    auto* impl_ = &impl_s;
    const FheParams& p = ctx.params();
    const size_t n = p.n(), L = p.n_limbs(), poly = L * n;
    dpfhe_ctx* h = static_cast<dpfhe_ctx*>(ctx.handle());
    PolyBuffer a(ctx, 1, 1, true), e(ctx, 1, 1, false), t(ctx, 1, 1, true);
    std::vector<uint64_t> ha(poly), he(poly);
    const size_t digits = n_digits ? n_digits : L;
    for (size_t j = 0; j < digits; ++j) {
        for (size_t l = 0; l < L; ++l)
            for (size_t k = 0; k < n; ++k) ha[l * n + k] = impl_->rng->below(p.moduli[l]);      // uniform: any domain
        for (size_t k = 0; k < n; ++k) {
            const int64_t ev = impl_->rng->error();
            for (size_t l = 0; l < L; ++l) he[l * n + k] = lift_signed(ev, p.moduli[l]);
        }
        a.copy_from_host(ha.data());
        e.copy_from_host(he.data());
        e.set_ntt(false);
        uint64_t* b = out.data() + (j * 2 + 0) * poly;   // evk_j[0]
        uint64_t* a_out = out.data() + (j * 2 + 1) * poly;  // evk_j[1] = a_j
        check(dpfhe_ntt_fwd(h, e.data(), 1, nullptr), "dpfhe_ntt_fwd");                                   // NTT(e_j)
        check(dpfhe_dyadic_mul(h, t.data(), a.data(), impl_->sk->ntt(), 1, nullptr), "dpfhe_dyadic_mul");  // a_j s
        check(dpfhe_sub(h, b, e.data(), t.data(), 1, nullptr), "dpfhe_sub");                               // e_j - a_j s
        // + g_j * target : the target polynomial in limb j only (g_j = 1 mod q_j, 0 mod the other primes)
        check(dpfhe_add(h, t.data(), b, d_target_ntt, 1, nullptr), "dpfhe_add");
        hip_check(hipMemcpyAsync(b + j * n, t.data() + j * n, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, nullptr), "hipMemcpyAsync");
        hip_check(hipMemcpyAsync(a_out, a.data(), poly * sizeof(uint64_t), hipMemcpyDeviceToDevice, nullptr), "hipMemcpyAsync");
        ctx.synchronize();
    }
    out.set_ntt(true);
}
}  // namespace

// ---- Encryptor --------------------------------------------------------------------------------------------------------------
class Encryptor::Impl {
public:
    const Context* ctx = nullptr;
    const SecretKey* sk = nullptr;     // symmetric mode
    const PublicKey* pk = nullptr;     // public-key mode
    Sampler rng;
};
Encryptor::Encryptor(const Context& ctx, const SecretKey& sk) : impl_(new Impl) { impl_->ctx = &ctx; impl_->sk = &sk; }
Encryptor::Encryptor(const Context& ctx, const SecretKey& sk, TestSeed seed) : impl_(new Impl) {
    impl_->ctx = &ctx; impl_->sk = &sk; impl_->rng = Sampler(seed);
}
static void check_public_key(const PublicKey& pk) {
    if (!pk.is_ntt() || pk.size() != 2 || pk.batch() != 1) throw Exception(ErrorCode::INVALID_ARGUMENT, "Encryptor: public key must be one 2-component NTT-domain item");
}
Encryptor::Encryptor(const Context& ctx, const PublicKey& pk) : impl_(new Impl) {
    check_public_key(pk);
    impl_->ctx = &ctx; impl_->pk = &pk;
}
Encryptor::Encryptor(const Context& ctx, const PublicKey& pk, TestSeed seed) : impl_(new Impl) {
    check_public_key(pk);
    impl_->ctx = &ctx; impl_->pk = &pk; impl_->rng = Sampler(seed);
}
Encryptor::~Encryptor() = default;

namespace {
// c0 = -(a s) + e + scale * m, c1 = a with scale given per limb (2^log2_scale for the approximate flavour, floor(Q/t) for
// the exact one)
template <class Rng>
void encrypt_scaled(const Context& ctx, const SecretKey& sk, Rng& rng, const int64_t* messages, const std::vector<uint64_t>& scale, Ciphertext& out) {
    const FheParams& p = ctx.params();
    const size_t n = p.n(), L = p.n_limbs(), poly = L * n;
    dpfhe_ctx* h = static_cast<dpfhe_ctx*>(ctx.handle());
    PolyBuffer a(ctx, 1, 1, false), t(ctx, 1, 1, false);
    std::vector<uint64_t> ha(poly), hm(poly);
    for (size_t item = 0; item < out.batch(); ++item) {
        for (size_t l = 0; l < L; ++l) {
            const uint64_t q = p.moduli[l];
            for (size_t k = 0; k < n; ++k) ha[l * n + k] = rng.below(q);
            for (size_t k = 0; k < n; ++k) hm[l * n + k] = (uint64_t)((u128)lift_signed(messages[item * n + k], q) * scale[l] % q);
        }
        for (size_t k = 0; k < n; ++k) {   // + e, the same small integer in every limb
            const int64_t ev = rng.error();
            for (size_t l = 0; l < L; ++l) { const uint64_t q = p.moduli[l]; uint64_t v = hm[l * n + k] + lift_signed(ev, q); hm[l * n + k] = v >= q ? v - q : v; }
        }
        uint64_t* c0 = out.data() + (item * 2 + 0) * poly;
        uint64_t* c1 = out.data() + (item * 2 + 1) * poly;
        a.copy_from_host(ha.data());                       // c1 = a (coefficient domain)
        hip_check(hipMemcpy(c1, a.data(), poly * sizeof(uint64_t), hipMemcpyDeviceToDevice), "hipMemcpy D2D");
        t.copy_from_host(hm.data());                       // e + scale m
        check(dpfhe_ntt_fwd(h, a.data(), 1, nullptr), "dpfhe_ntt_fwd");
        check(dpfhe_dyadic_mul(h, a.data(), a.data(), sk.ntt(), 1, nullptr), "dpfhe_dyadic_mul");
        check(dpfhe_ntt_inv(h, a.data(), 1, nullptr), "dpfhe_ntt_inv");          // a s
        check(dpfhe_sub(h, c0, t.data(), a.data(), 1, nullptr), "dpfhe_sub");     // c0 = e + scale m - a s
        ctx.synchronize();
    }
    out.set_ntt(false);
}
// (c0, c1) = (u pk0 + e1 + scale m, u pk1 + e2), u ternary, e1 / e2 uniform in [-8, 8]
template <class Rng>
void encrypt_scaled_pk(const Context& ctx, const PublicKey& pk, Rng& rng, const int64_t* messages, const std::vector<uint64_t>& scale, Ciphertext& out) {
    const FheParams& p = ctx.params();
    const size_t n = p.n(), L = p.n_limbs(), poly = L * n;
    dpfhe_ctx* h = static_cast<dpfhe_ctx*>(ctx.handle());
    PolyBuffer u(ctx, 1, 1, false), t(ctx, 1, 2, false);
    std::vector<uint64_t> hu(poly), ht(2 * poly);
    for (size_t item = 0; item < out.batch(); ++item) {
        for (size_t k = 0; k < n; ++k) {
            const int64_t uv = rng.ternary(), e1 = rng.error(), e2 = rng.error();
            for (size_t l = 0; l < L; ++l) {
                const uint64_t q = p.moduli[l];
                hu[l * n + k] = lift_signed(uv, q);
                const uint64_t m = (uint64_t)((u128)lift_signed(messages[item * n + k], q) * scale[l] % q), v = m + lift_signed(e1, q);
                ht[l * n + k] = v >= q ? v - q : v;                  // e1 + scale m
                ht[poly + l * n + k] = lift_signed(e2, q);           // e2
            }
        }
        uint64_t* c = out.data() + item * 2 * poly;
        u.copy_from_host(hu.data());
        t.copy_from_host(ht.data());
        check(dpfhe_ntt_fwd(h, u.data(), 1, nullptr), "dpfhe_ntt_fwd");
        check(dpfhe_dyadic_mul(h, c, u.data(), pk.data(), 1, nullptr), "dpfhe_dyadic_mul");                  // u pk0
        check(dpfhe_dyadic_mul(h, c + poly, u.data(), pk.data() + poly, 1, nullptr), "dpfhe_dyadic_mul");    // u pk1
        check(dpfhe_ntt_inv(h, c, 2, nullptr), "dpfhe_ntt_inv");
        check(dpfhe_add(h, c, c, t.data(), 2, nullptr), "dpfhe_add");
        ctx.synchronize();
    }
    out.set_ntt(false);
}
Big modulus_product(const FheParams& p) {
    Big Q{1};
    for (uint64_t q : p.moduli) big_mul_small(Q, q);
    return Q;
}
}  // namespace

void Encryptor::encrypt(const int64_t* messages, unsigned log2_scale, Ciphertext& out) {
    if (!messages) throw Exception(ErrorCode::INVALID_ARGUMENT, "encrypt: null messages");
    if (out.size() != 2) throw Exception(ErrorCode::INVALID_ARGUMENT, "encrypt: output must be a 2-component ciphertext");
    if (log2_scale > 200) throw Exception(ErrorCode::INVALID_ARGUMENT, "encrypt: log2_scale too large");
    const FheParams& p = impl_->ctx->params();
    std::vector<uint64_t> scale(p.n_limbs());
    for (size_t l = 0; l < p.n_limbs(); ++l) scale[l] = powmod(2, log2_scale, p.moduli[l]);
    if (impl_->pk) encrypt_scaled_pk(*impl_->ctx, *impl_->pk, impl_->rng, messages, scale, out);
    else encrypt_scaled(*impl_->ctx, *impl_->sk, impl_->rng, messages, scale, out);
}

void Encryptor::encrypt_exact(const int64_t* messages, uint64_t t, Ciphertext& out) {
    if (!messages) throw Exception(ErrorCode::INVALID_ARGUMENT, "encrypt_exact: null messages");
    if (out.size() != 2) throw Exception(ErrorCode::INVALID_ARGUMENT, "encrypt_exact: output must be a 2-component ciphertext");
    if (t < 2) throw Exception(ErrorCode::INVALID_ARGUMENT, "encrypt_exact: plaintext modulus must be >= 2");
    const FheParams& p = impl_->ctx->params();
    Big delta = modulus_product(p);
    big_divmod_small(delta, t);   // floor(Q / t)
    std::vector<uint64_t> scale(p.n_limbs());
    for (size_t l = 0; l < p.n_limbs(); ++l) scale[l] = big_mod_small(delta, p.moduli[l]);
    if (impl_->pk) encrypt_scaled_pk(*impl_->ctx, *impl_->pk, impl_->rng, messages, scale, out);
    else encrypt_scaled(*impl_->ctx, *impl_->sk, impl_->rng, messages, scale, out);
}

// ---- Decryptor --------------------------------------------------------------------------------------------------------------
class Decryptor::Impl {
public:
    const Context* ctx = nullptr;
    const SecretKey* sk = nullptr;
    std::vector<uint64_t> garner_inv;  // [i][j<i]: (q_j)^-1 mod q_i  (mixed-radix conversion)
    Big Q, halfQ;
};
Decryptor::Decryptor(const Context& ctx, const SecretKey& sk) : impl_(new Impl) {
    impl_->ctx = &ctx; impl_->sk = &sk;
    const FheParams& p = ctx.params();
    const size_t L = p.n_limbs();
    impl_->garner_inv.assign(L * L, 0);
    for (size_t i = 0; i < L; ++i)
        for (size_t j = 0; j < i; ++j) impl_->garner_inv[i * L + j] = powmod(p.moduli[j] % p.moduli[i], p.moduli[i] - 2, p.moduli[i]);
    impl_->Q = Big{1};
    for (size_t i = 0; i < L; ++i) big_mul_small(impl_->Q, p.moduli[i]);
    impl_->halfQ = impl_->Q;
    for (size_t i = 0; i < impl_->halfQ.size(); ++i) {  // >> 1
        impl_->halfQ[i] >>= 1;
        if (i + 1 < impl_->halfQ.size()) impl_->halfQ[i] |= impl_->Q[i + 1] << 63;
    }
}
Decryptor::~Decryptor() = default;

namespace {
// phase = c0 + c1 s (+ c2 s^2) per item on the device (NTT domain), then per coefficient the Garner mixed-radix digits
// x = v0 + v1 q0 + v2 q0 q1 + ... of its CRT composition in [0, Q); f(item, k, digits) consumes them
template <class F>
void for_each_phase(const Context& ctx, const SecretKey& sk, const std::vector<uint64_t>& garner_inv, const Ciphertext& ct, F f) {
    const FheParams& p = ctx.params();
    const size_t n = p.n(), L = p.n_limbs(), poly = L * n;
    dpfhe_ctx* h = static_cast<dpfhe_ctx*>(ctx.handle());
    PolyBuffer acc(ctx, 1, 1, true), t(ctx, 1, 1, true);
    std::vector<uint64_t> ph(poly), digit(L);
    for (size_t item = 0; item < ct.batch(); ++item) {
        const uint64_t* c = ct.data() + item * ct.size() * poly;
        check(dpfhe_ntt_fwd_oop(h, t.data(), c + poly, 1, nullptr), "dpfhe_ntt_fwd_oop");
        check(dpfhe_dyadic_mul(h, acc.data(), t.data(), sk.ntt(), 1, nullptr), "dpfhe_dyadic_mul");
        if (ct.size() == 3) {
            check(dpfhe_ntt_fwd_oop(h, t.data(), c + 2 * poly, 1, nullptr), "dpfhe_ntt_fwd_oop");
            check(dpfhe_dyadic_mul_add(h, acc.data(), t.data(), sk.ntt_squared(), 1, nullptr), "dpfhe_dyadic_mul_add");
        }
        check(dpfhe_ntt_inv(h, acc.data(), 1, nullptr), "dpfhe_ntt_inv");
        check(dpfhe_add(h, acc.data(), acc.data(), c, 1, nullptr), "dpfhe_add");
        ctx.synchronize();
        hip_check(hipMemcpy(ph.data(), acc.data(), poly * sizeof(uint64_t), hipMemcpyDeviceToHost), "hipMemcpy D2H");
        for (size_t k = 0; k < n; ++k) {
            for (size_t i = 0; i < L; ++i) {
                const uint64_t qi = p.moduli[i];
                uint64_t v = ph[i * n + k] % qi;
                for (size_t j = 0; j < i; ++j) {
                    const uint64_t dj = digit[j] % qi;
                    v = v >= dj ? v - dj : v + qi - dj;
                    v = (uint64_t)((u128)v * garner_inv[i * L + j] % qi);
                }
                digit[i] = v;
            }
            f(item, k, digit);
        }
    }
}
}  // namespace

void Decryptor::decrypt(const Ciphertext& ct, unsigned log2_scale, int64_t* out) {
    if (!out) throw Exception(ErrorCode::INVALID_ARGUMENT, "decrypt: null output");
    if (ct.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "decrypt: ciphertext must be in the coefficient domain");
    const FheParams& p = impl_->ctx->params();
    const size_t n = p.n(), L = p.n_limbs();
    for_each_phase(*impl_->ctx, *impl_->sk, impl_->garner_inv, ct, [&](size_t item, size_t k, const std::vector<uint64_t>& digit) {
        Big x{0};
        for (size_t i = L; i-- > 0;) { big_mul_small(x, p.moduli[i]); big_add(x, Big{digit[i]}); }
        const bool neg = big_cmp(x, impl_->halfQ) > 0;
        if (neg) x = big_sub(impl_->Q, x);
        big_shr_round(x, log2_scale);
        for (size_t i = 1; i < x.size(); ++i)
            if (x[i]) throw Exception(ErrorCode::RUNTIME_ERROR, "decrypt: value does not fit 62 bits (scale or noise overflow)");
        if (x[0] >> 62) throw Exception(ErrorCode::RUNTIME_ERROR, "decrypt: value does not fit 62 bits (scale or noise overflow)");
        out[item * n + k] = neg ? -(int64_t)x[0] : (int64_t)x[0];
    });
}

void Decryptor::decrypt_exact(const Ciphertext& ct, uint64_t t, uint64_t* out) {
    if (!out) throw Exception(ErrorCode::INVALID_ARGUMENT, "decrypt_exact: null output");
    if (ct.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "decrypt_exact: ciphertext must be in the coefficient domain");
    if (t < 2 || t >> 32) throw Exception(ErrorCode::INVALID_ARGUMENT, "decrypt_exact: plaintext modulus must be in [2, 2^32)");
    const FheParams& p = impl_->ctx->params();
    const size_t n = p.n(), L = p.n_limbs();
    for_each_phase(*impl_->ctx, *impl_->sk, impl_->garner_inv, ct, [&](size_t item, size_t k, const std::vector<uint64_t>& digit) {
        // x / Q = (v0 + q0 (v1 + q1 (...))) / (q0 q1 ...) evaluated from the lowest digit: f <- (v_i + f) / q_i.  The phase is
        // floor(Q/t) m + small noise, so t x / Q sits within ~2^-200 of an integer and 64-bit long double rounding is exact.
        long double f = 0.0L;
        for (size_t i = 0; i < L; ++i) f = ((long double)digit[i] + f) / (long double)p.moduli[i];
        const long double r = f * (long double)t;
        uint64_t m = (uint64_t)(r + 0.5L);
        out[item * n + k] = m >= t ? m - t : m;
    });
}

double Decryptor::noise_budget_bits(const Ciphertext& ct, uint64_t t) {
    if (ct.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "noise_budget_bits: ciphertext must be in the coefficient domain");
    if (t < 2 || t >> 32) throw Exception(ErrorCode::INVALID_ARGUMENT, "noise_budget_bits: plaintext modulus must be in [2, 2^32)");
    const FheParams& p = impl_->ctx->params();
    const size_t L = p.n_limbs();
    auto bits = [](const Big& a) {
        for (size_t i = a.size(); i-- > 0;)
            if (a[i]) return (double)(i * 64) + (double)(64 - __builtin_clzll(a[i]));
        return 0.0;
    };
    double worst = 0.0;   // most noise bits seen
    for_each_phase(*impl_->ctx, *impl_->sk, impl_->garner_inv, ct, [&](size_t, size_t, const std::vector<uint64_t>& digit) {
        // m = round(t x / Q) from the mixed-radix digits (exact: the noise is far below the long double's resolution of 1/2), then the
        // noise  e = t x - m Q  in exact integers
        long double f = 0.0L;
        for (size_t i = 0; i < L; ++i) f = ((long double)digit[i] + f) / (long double)p.moduli[i];
        const uint64_t m = (uint64_t)(f * (long double)t + 0.5L);
        Big x{0};
        for (size_t i = L; i-- > 0;) { big_mul_small(x, p.moduli[i]); big_add(x, Big{digit[i]}); }
        big_mul_small(x, t);
        Big mq(impl_->Q);
        big_mul_small(mq, m);
        const Big e = big_cmp(x, mq) >= 0 ? big_sub(x, mq) : big_sub(mq, x);
        const double b = bits(e);
        if (b > worst) worst = b;
    });
    return bits(impl_->Q) - 1.0 - worst;
}

// ---- HybridKeySwitcher ---------------------------------------------------------------------------------------------------
class HybridKeySwitcher::Impl {
public:
    const Context* data_ctx = nullptr;
    std::unique_ptr<Context> ext;
    std::unique_ptr<SecretKey> sk_ext;
    std::unique_ptr<PolyBuffer> relin;                       // [Ld][2][L][N]
    std::vector<std::pair<uint32_t, std::unique_ptr<PolyBuffer>>> galois;
    std::vector<std::pair<std::vector<uint32_t>, std::unique_ptr<PolyBuffer>>> packed;   // element list -> its keys back to back
    std::unique_ptr<PolyBuffer> scratch_work;      // batched rotations: reused across calls (one caller at a time per switcher)
    std::unique_ptr<Ciphertext> scratch_rotated;
    std::unique_ptr<PolyBuffer> scratch_digits;    // hoisted rotations: NTT of the lifted digits, [Ld][L][N]
    std::unique_ptr<Ciphertext> scratch_in_ntt;    // rotate_hoisted_qp: NTT of the inputs on the data limbs
    void init(const Context& data_ctx, const SecretKey& sk, uint64_t special_prime, uint64_t special_psi);
    // the keys of an element list, packed back to back once and cached
    const PolyBuffer* packed_keys(const std::vector<uint32_t>& elts, Stream* s = nullptr) {
        for (auto& kv : packed) if (kv.first == elts) return kv.second.get();
        const FheParams& pe = ext->params();
        const size_t L = pe.n_limbs(), Ld = L - 1, key_words = Ld * 2 * L * pe.n();
        std::unique_ptr<PolyBuffer> buf(new PolyBuffer(*ext, elts.size() * Ld, 2, true));
        for (size_t i = 0; i < elts.size(); ++i) {
            const PolyBuffer* key = nullptr;
            for (auto& kv : galois) if (kv.first == elts[i]) key = kv.second.get();
            if (!key) throw Exception(ErrorCode::INVALID_STATE, "HybridKeySwitcher: no key for an element (add_galois_element first)");
            // on the caller's stream: a blocking null-stream copy would not order against work on a non-blocking stream
            hip_check(hipMemcpyAsync(buf->data() + i * key_words, key->data(), key_words * sizeof(uint64_t), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(s)),
                      "hipMemcpyAsync D2D");
        }
        // The pack is cached and handed to LATER callers on ANY stream: it happens once per element list, so the copies are simply waited for
        // here - nothing orders another stream's kernels against an asynchronous copy they never saw being enqueued.
        hip_check(hipStreamSynchronize(static_cast<hipStream_t>(s)), "hipStreamSynchronize (key pack)");
        packed.emplace_back(elts, std::move(buf));
        return packed.back().second.get();
    }
    void ensure_scratch(size_t k) {
        if (!scratch_work || scratch_work->batch() < k) {
            scratch_work.reset(new PolyBuffer(*ext, k, 2, false));
            scratch_rotated.reset(new Ciphertext(*data_ctx, 2, k));
        }
    }
    Sampler rng;
    uint64_t p_special = 0;

    // target (NTT domain on ext, all limbs) scaled by P limb-wise: P mod q_i for data limbs, 0 for the P limb
    void make_key(const uint64_t* d_target_ntt_ext, PolyBuffer& out) {
        const FheParams& pe = ext->params();
        const size_t n = pe.n(), L = pe.n_limbs();
        std::vector<uint64_t> host(L * n);
        hip_check(hipMemcpy(host.data(), d_target_ntt_ext, L * n * sizeof(uint64_t), hipMemcpyDeviceToHost), "hipMemcpy D2H");
        for (size_t l = 0; l < L; ++l) {
            const uint64_t q = pe.moduli[l], f = p_special % q;   // 0 on the special limb itself
            for (size_t k = 0; k < n; ++k) host[l * n + k] = (uint64_t)((u128)host[l * n + k] * f % q);
        }
        PolyBuffer scaled(*ext, 1, 1, true);
        scaled.copy_from_host(host.data());
        make_switch_key(*ext, *sk_ext, rng, scaled.data(), out, L - 1, nullptr);
    }
};

HybridKeySwitcher::HybridKeySwitcher(const Context& data_ctx, const SecretKey& sk, uint64_t special_prime, uint64_t special_psi)
    : impl_(new Impl) {
    impl_->init(data_ctx, sk, special_prime, special_psi);
}
HybridKeySwitcher::HybridKeySwitcher(const Context& data_ctx, const SecretKey& sk, uint64_t special_prime, uint64_t special_psi, TestSeed seed)
    : impl_(new Impl) {
    impl_->rng = Sampler(seed);
    impl_->init(data_ctx, sk, special_prime, special_psi);
}
void HybridKeySwitcher::Impl::init(const Context& data_ctx, const SecretKey& sk, uint64_t special_prime, uint64_t special_psi) {
    Impl* impl_ = this;
    impl_->data_ctx = &data_ctx;
    FheParams pe = data_ctx.params();
    pe.moduli.push_back(special_prime);
    pe.psi.push_back(special_psi);
    impl_->ext.reset(new Context(pe, data_ctx.device_id()));
    impl_->sk_ext.reset(new SecretKey(*impl_->ext, sk.coefficients()));
    impl_->p_special = special_prime;
    impl_->relin.reset(new PolyBuffer(*impl_->ext, pe.n_limbs() - 1, 2, true));
    impl_->make_key(impl_->sk_ext->ntt_squared(), *impl_->relin);
}
HybridKeySwitcher::~HybridKeySwitcher() = default;

void HybridKeySwitcher::add_galois_element(uint32_t g) {
    const FheParams& pe = impl_->ext->params();
    const size_t n = pe.n(), L = pe.n_limbs();
    if (!(g & 1u) || g >= 2 * n) throw Exception(ErrorCode::INVALID_ARGUMENT, "add_galois_element: element must be odd and < 2N");
    for (auto& kv : impl_->galois) if (kv.first == g) return;
    const std::vector<int8_t>& s = impl_->sk_ext->coefficients();
    std::vector<uint64_t> host(L * n, 0);
    for (size_t i = 0; i < n; ++i) {
        const size_t idx = (i * (size_t)g) & (2 * n - 1);
        const int64_t v = idx < n ? s[i] : -s[i];
        for (size_t l = 0; l < L; ++l) host[l * n + (idx & (n - 1))] = lift_signed(v, pe.moduli[l]);
    }
    PolyBuffer target(*impl_->ext, 1, 1, false);
    target.copy_from_host(host.data());
    Evaluator ev(*impl_->ext);
    ev.transform_to_ntt_inplace(target);
    impl_->ext->synchronize();
    std::unique_ptr<PolyBuffer> key(new PolyBuffer(*impl_->ext, L - 1, 2, true));
    impl_->make_key(target.data(), *key);
    impl_->galois.emplace_back(g, std::move(key));
}

void HybridKeySwitcher::relinearize(const Ciphertext& in3, Ciphertext& out2, Stream* s) const {
    if (in3.is_ntt() || in3.size() != 3 || out2.size() != 2 || out2.batch() != in3.batch())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "HybridKeySwitcher::relinearize: 3-component coefficient-domain input, 2-component output");
    PolyBuffer work(*impl_->ext, in3.batch(), 2, false);
    check(dpfhe_relinearize_hybrid(static_cast<dpfhe_ctx*>(impl_->ext->handle()), out2.data(), in3.data(), impl_->relin->data(), work.data(), in3.batch(), s),
          "dpfhe_relinearize_hybrid");
    hip_check(hipStreamSynchronize(static_cast<hipStream_t>(s)), "hipStreamSynchronize");   // `work` is freed on return
    out2.set_ntt(false);
}

void HybridKeySwitcher::apply_galois(const Ciphertext& in2, uint32_t g, Ciphertext& out2, Stream* s) const {
    if (in2.is_ntt() || in2.size() != 2 || out2.size() != 2 || out2.batch() != in2.batch())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "HybridKeySwitcher::apply_galois: 2-component coefficient-domain input and output");
    const PolyBuffer* key = nullptr;
    for (auto& kv : impl_->galois) if (kv.first == g) key = kv.second.get();
    if (!key) throw Exception(ErrorCode::INVALID_STATE, "HybridKeySwitcher::apply_galois: no key for this element (add_galois_element first)");
    Ciphertext rotated(*impl_->data_ctx, 2, in2.batch());
    PolyBuffer work(*impl_->ext, in2.batch(), 2, false);
    check(dpfhe_apply_galois(static_cast<dpfhe_ctx*>(impl_->data_ctx->handle()), rotated.data(), in2.data(), in2.batch() * 2, g, s), "dpfhe_apply_galois");
    check(dpfhe_switch_key_hybrid(static_cast<dpfhe_ctx*>(impl_->ext->handle()), out2.data(), rotated.data(), key->data(), work.data(), in2.batch(), s),
          "dpfhe_switch_key_hybrid");
    hip_check(hipStreamSynchronize(static_cast<hipStream_t>(s)), "hipStreamSynchronize");
    out2.set_ntt(false);
}

void HybridKeySwitcher::apply_galois_many(const Ciphertext& in2, const std::vector<uint32_t>& elts, Ciphertext& out2, size_t out_first, Stream* s) const {
    if (in2.batch() != 1 && in2.batch() != elts.size())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "HybridKeySwitcher::apply_galois_many: input of 1 or k items, output with room for k items");
    apply_galois_range(in2, 0, in2.batch() == 1 && elts.size() != 1, elts, out2, out_first, s);
}

void HybridKeySwitcher::apply_galois_range(const Ciphertext& in2, size_t in_first, bool broadcast, const std::vector<uint32_t>& elts, Ciphertext& out2,
                                           size_t out_first, Stream* s) const {
    const size_t k = elts.size();
    if (k == 0) return;
    if (in2.is_ntt() || in2.size() != 2 || out2.size() != 2 || in_first + (broadcast ? 1 : k) > in2.batch() || out_first + k > out2.batch())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "HybridKeySwitcher::apply_galois_range: input range of 1 (broadcast) or k items, output with room for k items");
    const FheParams& pe = impl_->ext->params();
    const size_t L = pe.n_limbs(), Ld = L - 1, n = pe.n(), ct_words = 2 * Ld * n;
    const PolyBuffer* keys = impl_->packed_keys(elts, s);
    impl_->ensure_scratch(k);   // kept for the next call: no allocation and no host synchronisation on the steady path
    check(dpfhe_rotate_hybrid_batch(static_cast<dpfhe_ctx*>(impl_->ext->handle()), out2.data() + out_first * ct_words, in2.data() + in_first * ct_words,
                                    broadcast ? 1 : k, elts.data(), keys->data(), impl_->scratch_work->data(), impl_->scratch_rotated->data(), k, s),
          "dpfhe_rotate_hybrid_batch");
    out2.set_ntt(false);
}

void HybridKeySwitcher::apply_galois_hoisted(const Ciphertext& in2, size_t in_first, size_t n_items, const std::vector<uint32_t>& elts, Ciphertext& out2,
                                             size_t out_first, Stream* s) const {
    const size_t k = elts.size();
    if (k == 0 || n_items == 0) return;
    if (in2.is_ntt() || in2.size() != 2 || out2.size() != 2 || in_first + n_items > in2.batch() || out_first + k * n_items > out2.batch())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "HybridKeySwitcher::apply_galois_hoisted: n_items input items, output with room for k * n_items items");
    const FheParams& pe = impl_->ext->params();
    const size_t L = pe.n_limbs(), Ld = L - 1, n = pe.n(), ct_words = 2 * Ld * n;
    const PolyBuffer* keys = impl_->packed_keys(elts, s);
    impl_->ensure_scratch(k * n_items);
    if (!impl_->scratch_digits || impl_->scratch_digits->batch() < n_items * Ld) impl_->scratch_digits.reset(new PolyBuffer(*impl_->ext, n_items * Ld, 1, true));
    check(dpfhe_rotate_hybrid_hoisted(static_cast<dpfhe_ctx*>(impl_->ext->handle()), out2.data() + out_first * ct_words, in2.data() + in_first * ct_words, n_items,
                                      elts.data(), keys->data(), impl_->scratch_work->data(), impl_->scratch_rotated->data(), impl_->scratch_digits->data(), k, s),
          "dpfhe_rotate_hybrid_hoisted");
    out2.set_ntt(false);
}

void HybridKeySwitcher::apply_galois_grouped(const Ciphertext& in2, size_t in_first, const std::vector<uint32_t>& elts, size_t group, Ciphertext& out2,
                                             size_t out_first, Stream* s) const {
    const size_t k = elts.size(), batch = k * group;
    if (batch == 0) return;
    if (in2.is_ntt() || in2.size() != 2 || out2.size() != 2 || in_first + batch > in2.batch() || out_first + batch > out2.batch())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "HybridKeySwitcher::apply_galois_grouped: k * group items in and out");
    const FheParams& pe = impl_->ext->params();
    const size_t Ld = pe.n_limbs() - 1, n = pe.n(), ct_words = 2 * Ld * n;
    const PolyBuffer* keys = impl_->packed_keys(elts, s);
    impl_->ensure_scratch(batch);
    check(dpfhe_rotate_hybrid_grouped(static_cast<dpfhe_ctx*>(impl_->ext->handle()), out2.data() + out_first * ct_words, in2.data() + in_first * ct_words, elts.data(), k,
                                      group, keys->data(), impl_->scratch_work->data(), impl_->scratch_rotated->data(), s),
          "dpfhe_rotate_hybrid_grouped");
    out2.set_ntt(false);
}

const Context& HybridKeySwitcher::extended_context() const { return *impl_->ext; }

void HybridKeySwitcher::rotate_hoisted_qp(const Ciphertext& in2, size_t in_first, size_t n_items, const std::vector<uint32_t>& elts, PolyBuffer& out_qp,
                                          size_t out_first, Stream* s) const {
    const size_t k = elts.size();
    if (n_items == 0) return;
    const FheParams& pe = impl_->ext->params();
    const size_t L = pe.n_limbs(), Ld = L - 1, n = pe.n();
    if (in2.is_ntt() || in2.size() != 2 || out_qp.size() != 2 || in_first + n_items > in2.batch() || out_first + (k + 1) * n_items > out_qp.batch() ||
        out_qp.words() != out_qp.batch() * 2 * L * n)
        throw Exception(ErrorCode::INVALID_ARGUMENT, "HybridKeySwitcher::rotate_hoisted_qp: n_items coefficient-domain inputs, (k + 1) * n_items items on the extended context out");
    const PolyBuffer* keys = k ? impl_->packed_keys(elts, s) : nullptr;
    if (!impl_->scratch_digits || impl_->scratch_digits->batch() < n_items * Ld) impl_->scratch_digits.reset(new PolyBuffer(*impl_->ext, n_items * Ld, 1, true));
    if (!impl_->scratch_in_ntt || impl_->scratch_in_ntt->batch() < n_items) impl_->scratch_in_ntt.reset(new Ciphertext(*impl_->data_ctx, 2, n_items, true));
    check(dpfhe_rotate_hoisted_qp(static_cast<dpfhe_ctx*>(impl_->ext->handle()), out_qp.data() + out_first * 2 * L * n, in2.data() + in_first * 2 * Ld * n, n_items,
                                  elts.data(), keys ? keys->data() : nullptr, impl_->scratch_in_ntt->data(), impl_->scratch_digits->data(), k, s),
          "dpfhe_rotate_hoisted_qp");
    out_qp.set_ntt(true);
}

void HybridKeySwitcher::switch_key_qp(const Ciphertext& in2, size_t in_first, const std::vector<uint32_t>& elts, size_t group, PolyBuffer& out_qp,
                                      size_t out_first, Stream* s) const {
    const size_t k = elts.size(), batch = k * group;
    if (batch == 0) return;
    const FheParams& pe = impl_->ext->params();
    const size_t L = pe.n_limbs(), Ld = L - 1, n = pe.n();
    if (in2.is_ntt() || in2.size() != 2 || out_qp.size() != 2 || in_first + batch > in2.batch() || out_first + batch > out_qp.batch() ||
        out_qp.words() != out_qp.batch() * 2 * L * n)
        throw Exception(ErrorCode::INVALID_ARGUMENT, "HybridKeySwitcher::switch_key_qp: k * group coefficient-domain items in, as many items on the extended context out");
    const PolyBuffer* keys = impl_->packed_keys(elts, s);
    check(dpfhe_switch_key_qp(static_cast<dpfhe_ctx*>(impl_->ext->handle()), out_qp.data() + out_first * 2 * L * n, in2.data() + in_first * 2 * Ld * n, keys->data(), k,
                              group, s),
          "dpfhe_switch_key_qp");
    out_qp.set_ntt(true);
}

// ---- N3: slot packing --------------------------------------------------------------------------------------------------------
class BatchEncoder::Impl {
public:
    uint64_t t = 0;
    size_t n = 0;
    int logn = 0;
    uint64_t n_inv = 0;
    std::vector<uint64_t> rp, irp;       // zeta^brv(i), zeta^-brv(i) mod t  (the library's NTT convention, over Z_t)
    std::vector<uint32_t> idx;           // slot -> NTT index: row 0 slots, then row 1 slots

    static uint32_t brv(uint32_t x, int bits) { uint32_t r = 0; for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; } return r; }
    uint64_t mul(uint64_t a, uint64_t b) const { return (uint64_t)((u128)a * b % t); }
    void ntt_fwd(std::vector<uint64_t>& a) const {   // natural in -> bit-reversed out: a^[k] = a(zeta^(2 brv(k) + 1))
        for (size_t m = 1, len = n / 2; m < n; m <<= 1, len >>= 1)
            for (size_t i = 0; i < m; ++i) {
                const uint64_t w = rp[m + i];
                for (size_t j = 2 * i * len; j < 2 * i * len + len; ++j) {
                    const uint64_t u = a[j], v = mul(a[j + len], w);
                    a[j] = u + v >= t ? u + v - t : u + v;
                    a[j + len] = u >= v ? u - v : u + t - v;
                }
            }
    }
    void ntt_inv(std::vector<uint64_t>& a) const {
        for (size_t m = n / 2, len = 1; m >= 1; m >>= 1, len <<= 1)
            for (size_t i = 0; i < m; ++i) {
                const uint64_t w = irp[m + i];
                for (size_t j = 2 * i * len; j < 2 * i * len + len; ++j) {
                    const uint64_t u = a[j], v = a[j + len];
                    a[j] = u + v >= t ? u + v - t : u + v;
                    a[j + len] = mul(u >= v ? u - v : u + t - v, w);
                }
            }
        for (auto& v : a) v = mul(v, n_inv);
    }
};

BatchEncoder::BatchEncoder(const Context& ctx, uint64_t t) : impl_(new Impl) {
    const size_t n = ctx.params().n();
    if (t < 3 || (t >> 32) || (t - 1) % (2 * n) != 0) throw Exception(ErrorCode::INVALID_ARGUMENT, "BatchEncoder: plaintext modulus must be a prime = 1 mod 2N below 2^32");
    impl_->t = t; impl_->n = n; impl_->logn = (int)ctx.params().log2_n;
    uint64_t zeta = 0;
    for (uint64_t g = 2; g < t && !zeta; ++g) {   // zeta = g^((t-1)/2N) has order exactly 2N iff zeta^N = -1
        const uint64_t z = powmod(g, (t - 1) / (2 * n), t);
        if (powmod(z, n, t) == t - 1) zeta = z;
    }
    if (!zeta) throw Exception(ErrorCode::INVALID_ARGUMENT, "BatchEncoder: no primitive 2N-th root of unity mod t (t not prime?)");
    const uint64_t izeta = powmod(zeta, t - 2, t);
    impl_->rp.assign(n, 0); impl_->irp.assign(n, 0);
    uint64_t pw = 1, ipw = 1;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t r = Impl::brv((uint32_t)i, impl_->logn);
        impl_->rp[r] = pw; impl_->irp[r] = ipw;
        pw = impl_->mul(pw, zeta); ipw = impl_->mul(ipw, izeta);
    }
    impl_->n_inv = powmod(n % t, t - 2, t);
    impl_->idx.assign(n, 0);
    uint64_t e = 1;
    for (size_t i = 0; i < n / 2; ++i) {
        impl_->idx[i] = Impl::brv((uint32_t)((e - 1) / 2), impl_->logn);                  // zeta^(3^i)
        impl_->idx[n / 2 + i] = Impl::brv((uint32_t)((2 * n - e - 1) / 2), impl_->logn);  // zeta^(-3^i)
        e = e * 3 % (2 * n);
    }
}
BatchEncoder::~BatchEncoder() = default;
uint64_t BatchEncoder::plain_modulus() const { return impl_->t; }
size_t BatchEncoder::slot_count() const { return impl_->n; }
size_t BatchEncoder::row_size() const { return impl_->n / 2; }

void BatchEncoder::encode(const uint64_t* slots, int64_t* coeffs) const {
    if (!slots || !coeffs) throw Exception(ErrorCode::INVALID_ARGUMENT, "BatchEncoder::encode: null argument");
    const size_t n = impl_->n;
    std::vector<uint64_t> a(n);
    for (size_t i = 0; i < n; ++i) {
        if (slots[i] >= impl_->t) throw Exception(ErrorCode::INVALID_ARGUMENT, "BatchEncoder::encode: slot value >= plaintext modulus");
        a[impl_->idx[i]] = slots[i];
    }
    impl_->ntt_inv(a);
    for (size_t i = 0; i < n; ++i) coeffs[i] = a[i] > impl_->t / 2 ? (int64_t)a[i] - (int64_t)impl_->t : (int64_t)a[i];
}
void BatchEncoder::decode(const uint64_t* coeffs, uint64_t* slots) const {
    if (!slots || !coeffs) throw Exception(ErrorCode::INVALID_ARGUMENT, "BatchEncoder::decode: null argument");
    const size_t n = impl_->n;
    std::vector<uint64_t> a(coeffs, coeffs + n);
    for (auto& v : a) v %= impl_->t;
    impl_->ntt_fwd(a);
    for (size_t i = 0; i < n; ++i) slots[i] = a[impl_->idx[i]];
}
uint32_t BatchEncoder::galois_element(int left_rotation) const {
    const long long row = (long long)impl_->n / 2;
    const uint64_t s = (uint64_t)(((left_rotation % row) + row) % row);
    return (uint32_t)powmod(3, s, 2 * impl_->n);
}

// ---- N3: packed matrix-vector product (diagonal method, baby-step / giant-step) -------------------------------------------
constexpr int kBabyShiftDefault = 1;

class PackedLinear::Impl {
public:
    const Context* ctx = nullptr;
    const BatchEncoder* enc = nullptr;
    HybridKeySwitcher* ks = nullptr;
    size_t out_dim = 0, in_dim = 0;
    size_t n = 0;        // input period: the input vector repeats every n slots of a row (power of two >= in_dim)
    size_t m = 0;        // diagonals per pass = output period (n, or the padded out_dim of a wide-input layer)
    size_t tpc = 1;      // tokens per ciphertext: 2 = the two slot rows carry two tokens (the windows of ONE row share the output blocks)
    size_t copies = 0;   // independent n-slot windows that share the output blocks = N / n (tpc = 2: of one row, N / 2 / n)
    size_t blocks = 0;   // output row blocks of m rows
    size_t passes = 0;   // output ciphertexts
    bool replicate = false;   // one block: every window computes it (the output is again a periodic vector)
    size_t n1 = 0, n2 = 0;
    std::unique_ptr<Plaintext> diag;   // [passes][n2][n1] pre-rotated diagonals, NTT domain
    std::vector<uint32_t> baby_elts, giant_elts, fold_elts;
    // per-layer scratch, reused by every apply() (one caller at a time).  Terms over Q P live on the key switcher's extended context.
    std::unique_ptr<PolyBuffer> babies_qp, inner_qp, terms_qp, ksum_qp;
    std::unique_ptr<Ciphertext> rot, fold;
    std::vector<uint32_t> inner_elts;                            // element of inner sum (pass, i): 1 for i = 0, the giant step's otherwise
    size_t tokens = 0;                                           // scratch capacity in tokens
    void ensure_tokens(size_t T) {
        if (T <= tokens) return;
        const Context& ext = ks->extended_context();
        babies_qp.reset(new PolyBuffer(ext, n1 * T, 2, true));                      // [n1][T]: P rot_j(x_t) + key-switching terms, NTT domain
        inner_qp.reset(new PolyBuffer(ext, passes * n2 * T, 2, true));              // [passes * n2][T]
        rot.reset(new Ciphertext(*ctx, 2, passes * n2 * T));                        // the inner sums, rotated by their giant step, divided by P
        if (n2 > 1) {
            terms_qp.reset(new PolyBuffer(ext, (n2 - 1) * T, 2, true));             // key inner products of one output ciphertext's giant steps
            ksum_qp.reset(new PolyBuffer(ext, T, 2, true));
        }
        if (!fold_elts.empty()) fold.reset(new Ciphertext(*ctx, 2, 2 * T));         // [2][T]: running sums | their rotation
        tokens = T;
    }

    // which output row a slot of pass `pass` holds (or npos)
    size_t row_of_slot(size_t pass, size_t slot) const {
        const size_t row = enc->row_size(), r = slot % row, rho = slot / row;
        const size_t c = r / n + (tpc == 2 ? 0 : rho * (row / n));
        const size_t b = replicate || m < n ? 0 : pass * copies + c;
        const size_t R = b * m + r % m;
        return R < out_dim ? R : (size_t)-1;
    }
};

PackedLinear::PackedLinear(const Context& ctx, const BatchEncoder& enc, HybridKeySwitcher& ks, const uint64_t* W, size_t d)
    : PackedLinear(ctx, enc, ks, W, d, d) {
    if (d < 2 || (d & (d - 1))) throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedLinear: d must be a power of two dividing N/2");
}

PackedLinear::PackedLinear(const Context& ctx, const BatchEncoder& enc, HybridKeySwitcher& ks, const uint64_t* W, size_t out_dim, size_t in_dim, size_t tokens_per_ciphertext)
    : impl_(new Impl) {
    if (tokens_per_ciphertext != 1 && tokens_per_ciphertext != 2) throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedLinear: one or two tokens per ciphertext");
    const FheParams& p = ctx.params();
    const size_t N = p.n(), L = p.n_limbs(), row = N / 2;
    if (!W || out_dim == 0 || in_dim == 0) throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedLinear: empty matrix");
    if (enc.slot_count() != N) throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedLinear: encoder and context disagree on N");
    auto pow2 = [](size_t v) { size_t x = 1; while (x < v) x <<= 1; return x; };
    Impl& I = *impl_;
    I.ctx = &ctx; I.enc = &enc; I.ks = &ks; I.out_dim = out_dim; I.in_dim = in_dim;
    I.n = pow2(in_dim) < 2 ? 2 : pow2(in_dim);
    if (I.n > row) throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedLinear: in_dim (padded to a power of two) must be <= N/2");
    I.tpc = tokens_per_ciphertext;
    I.copies = (I.tpc == 2 ? row : N) / I.n;
    const size_t mo = pow2(out_dim) < 2 ? 2 : pow2(out_dim);
    I.m = mo < I.n ? mo : I.n;                                  // wide-input layer: only m wrapped diagonals, folded afterwards
    I.blocks = (out_dim + I.m - 1) / I.m;
    I.replicate = I.blocks == 1;
    I.passes = I.replicate ? 1 : (I.blocks + I.copies - 1) / I.copies;
    size_t n1 = 1;
    while (n1 * n1 < I.m) n1 <<= 1;
    // A hoisted baby step (gathers + key inner products, no transform) costs about a third of a giant step (Ld transforms per limb + its
    // share of the inverse transform and the division by P), so the split leans towards baby steps: n1 = 2 sqrt(m) when m allows.
    int shift = kBabyShiftDefault;
    // (the split sweep behind this default: profiles/r03_bsgs_split_sweep.txt)
    for (; shift > 0 && n1 * 2 < I.m; --shift) n1 <<= 1;
    for (; shift < 0 && n1 > 2; ++shift) n1 >>= 1;
    I.n1 = n1; I.n2 = I.m / n1;
    const uint64_t t = enc.plain_modulus();
    for (size_t j = 1; j < I.n1; ++j) I.baby_elts.push_back(enc.galois_element((int)j));
    for (size_t i = 1; i < I.n2; ++i) I.giant_elts.push_back(enc.galois_element((int)(i * n1)));
    for (size_t sft = I.m; sft < I.n; sft <<= 1) I.fold_elts.push_back(enc.galois_element((int)sft));
    for (uint32_t g : I.baby_elts) ks.add_galois_element(g);
    for (uint32_t g : I.giant_elts) ks.add_galois_element(g);
    for (uint32_t g : I.fold_elts) ks.add_galois_element(g);
    for (size_t i = 0; i < out_dim * in_dim; ++i)
        if (W[i] >= t) throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedLinear: weight >= plaintext modulus");

    // Pre-rotated diagonals.  The product of giant step i lands on output slot r = r' - i n1 (row rotation), so position r' of
    // diagonal (i, j) carries the weight of the output row that slot r holds and of input index (r + k) mod n, k = i n1 + j.
    // They are multiplied with terms over Q P (the division by P comes after the sum), so they are encoded over all limbs of the
    // key switcher's extended context.
    const Context& ext = ks.extended_context();
    const FheParams& pe = ext.params();
    if (pe.log2_n != p.log2_n || pe.n_limbs() != L + 1 || !std::equal(p.moduli.begin(), p.moduli.end(), pe.moduli.begin()))
        throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedLinear: the key switcher was built for another context");
    const size_t Le = L + 1;
    I.diag.reset(new Plaintext(ext, I.passes * I.n2 * n1, /*is_ntt=*/false));
    std::vector<uint64_t> slots(N), host(n1 * Le * N);
    std::vector<int64_t> coeffs(N);
    for (size_t pass = 0; pass < I.passes; ++pass) {
        for (size_t i = 0; i < I.n2; ++i) {
            for (size_t j = 0; j < n1; ++j) {
                const size_t k = i * n1 + j;
                for (size_t rho = 0; rho < 2; ++rho)
                    for (size_t rp = 0; rp < row; ++rp) {
                        const size_t r = (rp + row - (i * n1) % row) % row;
                        const size_t R = I.row_of_slot(pass, rho * row + r), col = (r + k) % I.n;
                        slots[rho * row + rp] = (R != (size_t)-1 && col < in_dim) ? W[R * in_dim + col] : 0;
                    }
                enc.encode(slots.data(), coeffs.data());
                for (size_t l = 0; l < Le; ++l)
                    for (size_t c = 0; c < N; ++c) host[(j * Le + l) * N + c] = lift_signed(coeffs[c], pe.moduli[l]);
            }
            hip_check(hipMemcpy(I.diag->data() + ((pass * I.n2 + i) * n1) * Le * N, host.data(), host.size() * sizeof(uint64_t), hipMemcpyHostToDevice), "hipMemcpy H2D");
        }
    }
    Evaluator ev(ext);
    ev.transform_to_ntt_inplace(*I.diag);
    for (size_t pass = 0; pass < I.passes; ++pass) {
        I.inner_elts.push_back(1u);
        I.inner_elts.insert(I.inner_elts.end(), I.giant_elts.begin(), I.giant_elts.end());
    }
    I.ensure_tokens(1);
    ext.synchronize();
}
PackedLinear::~PackedLinear() = default;
size_t PackedLinear::dim() const { return impl_->m; }
size_t PackedLinear::in_dim() const { return impl_->in_dim; }
size_t PackedLinear::out_dim() const { return impl_->out_dim; }
size_t PackedLinear::input_period() const { return impl_->n; }
size_t PackedLinear::output_ciphertexts() const { return impl_->passes; }
size_t PackedLinear::baby_steps() const { return impl_->n1; }
size_t PackedLinear::giant_steps() const { return impl_->n2; }
size_t PackedLinear::key_switches_per_apply() const {
    return impl_->baby_elts.size() + impl_->passes * impl_->giant_elts.size() + impl_->fold_elts.size();
}

void PackedLinear::pack_input(const uint64_t* x, uint64_t* slots) const {
    const size_t N = impl_->enc->slot_count();
    for (size_t s = 0; s < N; ++s) {
        const size_t c = (s % (N / 2)) % impl_->n;
        slots[s] = c < impl_->in_dim ? x[c] : 0;
    }
}
size_t PackedLinear::tokens_per_ciphertext() const { return impl_->tpc; }
void PackedLinear::pack_input_rows(const uint64_t* x0, const uint64_t* x1, uint64_t* slots) const {
    const size_t N = impl_->enc->slot_count(), row = N / 2;
    for (size_t s = 0; s < N; ++s) {
        const size_t c = (s % row) % impl_->n;
        slots[s] = c < impl_->in_dim ? (s < row ? x0[c] : x1[c]) : 0;
    }
}
void PackedLinear::unpack_output_rows(const uint64_t* slots, uint64_t* y0, uint64_t* y1) const {
    const size_t N = impl_->enc->slot_count(), row = N / 2;
    for (size_t rho = 0; rho < 2; ++rho) {
        std::vector<char> seen(impl_->out_dim, 0);
        uint64_t* y = rho ? y1 : y0;
        for (size_t pass = 0; pass < impl_->passes; ++pass)
            for (size_t s = rho * row; s < (rho + 1) * row; ++s) {
                const size_t R = impl_->row_of_slot(pass, s);
                if (R != (size_t)-1 && !seen[R]) { y[R] = slots[pass * N + s]; seen[R] = 1; }
            }
    }
}
void PackedLinear::unpack_output(const uint64_t* slots, uint64_t* y) const {
    const size_t N = impl_->enc->slot_count();
    std::vector<char> seen(impl_->out_dim, 0);
    for (size_t pass = 0; pass < impl_->passes; ++pass)
        for (size_t s = 0; s < N; ++s) {
            const size_t R = impl_->row_of_slot(pass, s);
            if (R != (size_t)-1 && !seen[R]) { y[R] = slots[pass * N + s]; seen[R] = 1; }
        }
}

void PackedLinear::apply(const Ciphertext& x, Ciphertext& y, Stream* s) const {
    Impl& I = *impl_;
    const size_t T = x.batch();
    if (x.is_ntt() || x.size() != 2 || T == 0 || y.size() != 2 || y.batch() != I.passes * T)
        throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedLinear::apply: T 2-component coefficient-domain ciphertexts in, output_ciphertexts() * T out");
    I.ensure_tokens(T);   // (re)allocates only when a larger batch than ever before arrives
    const Context& ctx = *I.ctx;
    const FheParams& p = ctx.params();
    const size_t ct_words = 2 * p.n_limbs() * p.n(), n1 = I.n1, n2 = I.n2;
    hipStream_t hs = static_cast<hipStream_t>(s);
    dpfhe_ctx* h = static_cast<dpfhe_ctx*>(ctx.handle());
    dpfhe_ctx* he = static_cast<dpfhe_ctx*>(I.ks->extended_context().handle());
    // Layout of every intermediate: [rotation or diagonal index][token][component] - the token index sits where the plaintext
    // matvec sees "more components", so keys and diagonals are read once for all tokens.
    // baby steps: P rot_j(x_t) + key-switching term, j < n1, all tokens, ONE hoisted pass; they stay in the NTT domain over Q P
    I.ks->rotate_hoisted_qp(x, 0, T, I.baby_elts, *I.babies_qp, 0, s);
    // inner sums of all giant steps of all output ciphertexts of all tokens: ONE matrix-vector product over the pre-rotated diagonals
    check(dpfhe_matvec_plain_multi(he, I.inner_qp->data(), I.diag->data(), I.babies_qp->data(), I.passes * n2, n1, T, s), "dpfhe_matvec_plain_multi");
    // back to the coefficient domain, the giant step's automorphism applied by the transform's loads; then the ONE division by P
    // the baby steps and the plaintext products share
    check(dpfhe_ntt_inv_galois(he, I.inner_qp->data(), I.inner_qp->data(), T * 2, I.inner_elts.data(), I.passes * n2, s), "dpfhe_ntt_inv_galois");
    Ciphertext& rot = *I.rot;
    check(dpfhe_rescale(he, rot.data(), I.inner_qp->data(), I.passes * n2 * T * 2, s), "dpfhe_rescale");
    rot.set_ntt(false);
    // giant steps: key inner products of the rotated inner sums (i >= 1), summed over Q P; one inverse transform and one
    // division by P per output ciphertext, which also adds the c0 parts and the un-rotated inner sum
    uint64_t* sums = I.fold_elts.empty() ? y.data() : I.fold->data();   // wide-input layer: the block sum is folded below before it becomes y
    if (n2 > 1) {
        for (size_t pass = 0; pass < I.passes; ++pass) {
            const size_t base = pass * n2 * T;
            I.ks->switch_key_qp(rot, base + T, I.giant_elts, T, *I.terms_qp, 0, s);
            check(dpfhe_reduce_sum(he, I.ksum_qp->data(), I.terms_qp->data(), n2 - 1, 2 * T, s), "dpfhe_reduce_sum");
            check(dpfhe_ntt_inv(he, I.ksum_qp->data(), T * 2, s), "dpfhe_ntt_inv");
            check(dpfhe_rescale_bsgs(he, sums + pass * T * ct_words, I.ksum_qp->data(), rot.data() + base * ct_words, n2, T, s), "dpfhe_rescale_bsgs");
        }
    } else {
        hip_check(hipMemcpyAsync(sums, rot.data(), I.passes * T * ct_words * sizeof(uint64_t), hipMemcpyDeviceToDevice, hs), "hipMemcpyAsync");
    }
    // wide input (m < n): slot r holds the partial sum over input indices congruent to r + k; fold the n/m windows together
    if (!I.fold_elts.empty()) {
        Ciphertext& f = *I.fold;   // items [0, T): running sums, [T, 2T): their rotation
        f.set_ntt(false);
        for (size_t e = 0; e < I.fold_elts.size(); ++e) {
            const std::vector<uint32_t> one(1, I.fold_elts[e]);
            I.ks->apply_galois_grouped(f, 0, one, T, f, T, s);
            const bool last = e + 1 == I.fold_elts.size();
            check(dpfhe_add(h, last ? y.data() : f.data(), f.data(), f.data() + T * ct_words, 2 * T, s), "dpfhe_add");
        }
    }
    y.set_ntt(false);
    // enqueue only: the scratch belongs to the layer, the caller synchronises (Context::synchronize) before reading y
}

// ---- N3: hand-over between packed layers and the transformer block's linear skeleton ------------------------------------------
class PackedSelect::Impl {
public:
    const Context* ctx = nullptr;
    HybridKeySwitcher* ks = nullptr;
    size_t offset = 0, tpc = 1;
    uint32_t shift_elt = 0, swap_elt = 0;
    std::vector<uint32_t> spread_elts;          // right rotations by period, 2 period, ... up to half a slot row
    std::unique_ptr<Plaintext> mask;            // NTT domain: 1 on slots [0, length) of row 0, 0 elsewhere
    std::unique_ptr<Ciphertext> a, b;           // scratch, T items each
    size_t tokens = 0;
    void ensure(size_t T) {
        if (T <= tokens) return;
        a.reset(new Ciphertext(*ctx, 2, T));
        b.reset(new Ciphertext(*ctx, 2, T));
        tokens = T;
    }
};

PackedSelect::PackedSelect(const Context& ctx, const BatchEncoder& enc, HybridKeySwitcher& ks, size_t offset, size_t length, size_t period, size_t tokens_per_ciphertext) : impl_(new Impl) {
    if (tokens_per_ciphertext != 1 && tokens_per_ciphertext != 2) throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedSelect: one or two tokens per ciphertext");
    const FheParams& p = ctx.params();
    const size_t N = p.n(), row = N / 2, L = p.n_limbs();
    if (enc.slot_count() != N || length == 0 || period < length || (period & (period - 1)) || period > row || offset + length > row)
        throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedSelect: slice of slot row 0, period a power of two in [length, N/2]");
    Impl& I = *impl_;
    I.ctx = &ctx; I.ks = &ks; I.offset = offset; I.tpc = tokens_per_ciphertext;
    if (offset) { I.shift_elt = enc.galois_element((int)offset); ks.add_galois_element(I.shift_elt); }
    for (size_t sft = period; sft < row; sft <<= 1) { I.spread_elts.push_back(enc.galois_element(-(int)sft)); ks.add_galois_element(I.spread_elts.back()); }
    I.swap_elt = (uint32_t)(2 * N - 1);
    if (I.tpc == 1) ks.add_galois_element(I.swap_elt);
    std::vector<uint64_t> slots(N, 0), host(L * N);
    std::vector<int64_t> coeffs(N);
    for (size_t i = 0; i < length; ++i) { slots[i] = 1; if (I.tpc == 2) slots[row + i] = 1; }   // (two tokens: the same slice of row 1)
    enc.encode(slots.data(), coeffs.data());
    for (size_t l = 0; l < L; ++l)
        for (size_t c = 0; c < N; ++c) host[l * N + c] = lift_signed(coeffs[c], p.moduli[l]);
    I.mask.reset(new Plaintext(ctx, 1, false));
    I.mask->copy_from_host(host.data());
    Evaluator ev(ctx);
    ev.transform_to_ntt_inplace(*I.mask);
    I.ensure(1);
    ctx.synchronize();
}
PackedSelect::~PackedSelect() = default;
size_t PackedSelect::key_switches_per_apply() const { return (impl_->offset ? 1 : 0) + impl_->spread_elts.size() + (impl_->tpc == 1 ? 1 : 0); }

void PackedSelect::apply(const Ciphertext& x, Ciphertext& y, Stream* s) const {
    Impl& I = *impl_;
    const size_t T = x.batch();
    if (x.is_ntt() || x.size() != 2 || y.size() != 2 || y.batch() != T || T == 0)
        throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedSelect::apply: T 2-component coefficient-domain ciphertexts in and out");
    I.ensure(T);
    dpfhe_ctx* h = static_cast<dpfhe_ctx*>(I.ctx->handle());
    Ciphertext &a = *I.a, &b = *I.b;
    const Ciphertext* cur = &x;
    if (I.offset) {   // slot offset + i -> slot i
        I.ks->apply_galois_grouped(x, 0, std::vector<uint32_t>(1, I.shift_elt), T, a, 0, s);
        cur = &a;
    }
    // mask: NTT, every polynomial times the (broadcast) mask in one launch, back
    check(dpfhe_ntt_fwd_oop(h, b.data(), cur->data(), T * 2, s), "dpfhe_ntt_fwd_oop");
    check(dpfhe_multiply_plain(h, b.data(), b.data(), I.mask->data(), T * 2, s), "dpfhe_multiply_plain");
    check(dpfhe_ntt_inv(h, b.data(), T * 2, s), "dpfhe_ntt_inv");
    b.set_ntt(false);
    // spread along the row: b += rot(b, -period), then -2 period, ...; then the other row
    Ciphertext* have = &b;
    Ciphertext* tmp = &a;
    auto rotate_add = [&](uint32_t g, Ciphertext& out) {
        I.ks->apply_galois_grouped(*have, 0, std::vector<uint32_t>(1, g), T, *tmp, 0, s);
        check(dpfhe_add(h, out.data(), have->data(), tmp->data(), T * 2, s), "dpfhe_add");
        out.set_ntt(false);
    };
    if (I.tpc == 1) {
        for (uint32_t g : I.spread_elts) rotate_add(g, *have);
        rotate_add(I.swap_elt, y);
    } else {   // two tokens per ciphertext: every row keeps its own token - spread inside the rows only, the last step writes y
        const FheParams& p = I.ctx->params();
        if (I.spread_elts.empty()) {
            hip_check(hipMemcpyAsync(y.data(), have->data(), T * 2 * p.n_limbs() * p.n() * sizeof(uint64_t), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(s)), "hipMemcpyAsync");
            y.set_ntt(false);
        } else {
            for (size_t i = 0; i + 1 < I.spread_elts.size(); ++i) rotate_add(I.spread_elts[i], *have);
            rotate_add(I.spread_elts.back(), y);
        }
    }
}

class PackedTransformerBlock::Impl {
public:
    const Context* ctx = nullptr;
    HybridKeySwitcher* ks = nullptr;
    size_t d = 0, h = 0;
    std::unique_ptr<PackedLinear> qkv, proj, up, down;
    std::unique_ptr<PackedSelect> take_v;
    std::unique_ptr<Ciphertext> st[5], o, u, us, dn;   // stages + scratch, T items each
    size_t tokens = 0;
    void ensure(size_t T) {
        if (T <= tokens) return;
        for (auto& c : st) c.reset(new Ciphertext(*ctx, 2, T));
        o.reset(new Ciphertext(*ctx, 2, T)); u.reset(new Ciphertext(*ctx, 2, T)); us.reset(new Ciphertext(*ctx, 2, T)); dn.reset(new Ciphertext(*ctx, 2, T));
        tokens = T;
    }
};

PackedTransformerBlock::PackedTransformerBlock(const Context& ctx, const BatchEncoder& enc, HybridKeySwitcher& ks, const uint64_t* W_qkv, const uint64_t* W_o,
                                               const uint64_t* W_up, const uint64_t* W_down, size_t d, size_t h) : impl_(new Impl) {
    if (!W_qkv || !W_o || !W_up || !W_down || d == 0 || h == 0) throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedTransformerBlock: null or empty matrix");
    Impl& I = *impl_;
    I.ctx = &ctx; I.ks = &ks; I.d = d; I.h = h;
    I.qkv.reset(new PackedLinear(ctx, enc, ks, W_qkv, 3 * d, d));
    I.proj.reset(new PackedLinear(ctx, enc, ks, W_o, d, d));
    I.up.reset(new PackedLinear(ctx, enc, ks, W_up, h, d));
    I.down.reset(new PackedLinear(ctx, enc, ks, W_down, d, h));
    const size_t row = ctx.params().n() / 2;
    // the hand-overs below rely on: one output ciphertext per layer, outputs of the wide layers at slot r of row 0 (out >= period),
    // and W_down consuming a vector that fills a whole slot row
    if (I.qkv->output_ciphertexts() != 1 || I.up->output_ciphertexts() != 1 || I.down->output_ciphertexts() != 1 || I.proj->output_ciphertexts() != 1 ||
        3 * d < I.qkv->input_period() || 3 * d > row || h < I.up->input_period() || I.down->input_period() != row || I.proj->input_period() != I.qkv->input_period())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedTransformerBlock: needs 3 d <= N/2, h >= the padded d and the padded h = N/2 (GPT-2 small at N = 8192: d = 768, h = 3072)");
    I.take_v.reset(new PackedSelect(ctx, enc, ks, 2 * d, d, I.proj->input_period()));
    ks.add_galois_element((uint32_t)(2 * ctx.params().n() - 1));
    I.ensure(1);
}
PackedTransformerBlock::~PackedTransformerBlock() = default;
size_t PackedTransformerBlock::hidden() const { return impl_->d; }
size_t PackedTransformerBlock::inner() const { return impl_->h; }
size_t PackedTransformerBlock::key_switches_per_token() const {
    const Impl& I = *impl_;
    return I.qkv->key_switches_per_apply() + I.take_v->key_switches_per_apply() + I.proj->key_switches_per_apply() + I.up->key_switches_per_apply() + 1 +
           I.down->key_switches_per_apply();
}
void PackedTransformerBlock::pack_input(const uint64_t* x, uint64_t* slots) const { impl_->qkv->pack_input(x, slots); }
void PackedTransformerBlock::unpack_output(const uint64_t* slots, uint64_t* y) const { impl_->down->unpack_output(slots, y); }
const Ciphertext& PackedTransformerBlock::stage(int index) const {
    if (index < 0 || index > 4 || !impl_->st[index]) throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedTransformerBlock::stage: index in [0, 4]");
    return *impl_->st[index];
}

void PackedTransformerBlock::apply(const Ciphertext& x, Ciphertext& y, Stream* s) const {
    Impl& I = *impl_;
    const size_t T = x.batch();
    if (x.is_ntt() || x.size() != 2 || y.size() != 2 || y.batch() != T || T == 0)
        throw Exception(ErrorCode::INVALID_ARGUMENT, "PackedTransformerBlock::apply: T 2-component coefficient-domain ciphertexts in and out");
    I.ensure(T);
    dpfhe_ctx* h = static_cast<dpfhe_ctx*>(I.ctx->handle());
    const FheParams& p = I.ctx->params();
    const size_t words = T * 2 * p.n_limbs() * p.n();
    Ciphertext &qkv = *I.st[0], &a = *I.st[1], &h1 = *I.st[2], &u2 = *I.st[3], &h2 = *I.st[4];
    I.qkv->apply(x, qkv, s);                                   // q | k | v at slots 0 .. 3d-1 of row 0 (gpt_model.cpp:793)
    I.take_v->apply(qkv, a, s);                                // attention over one position: the output is v; re-packed as a layer input
    I.proj->apply(a, *I.o, s);                                 // attention-output projection
    check(dpfhe_add(h, h1.data(), x.data(), I.o->data(), T * 2, s), "dpfhe_add");   // residual
    h1.set_ntt(false);
    I.up->apply(h1, *I.u, s);                                  // FFN up (gpt_model.cpp:848): outputs at slot r of row 0
    I.ks->apply_galois_grouped(*I.u, 0, std::vector<uint32_t>(1, (uint32_t)(2 * p.n() - 1)), T, *I.us, 0, s);   // row swap
    check(dpfhe_add(h, u2.data(), I.u->data(), I.us->data(), T * 2, s), "dpfhe_add");                          // both rows: W_down's input packing
    u2.set_ntt(false);
    I.down->apply(u2, *I.dn, s);                               // FFN down
    check(dpfhe_add(h, h2.data(), h1.data(), I.dn->data(), T * 2, s), "dpfhe_add");   // residual
    h2.set_ntt(false);
    hip_check(hipMemcpyAsync(y.data(), h2.data(), words * sizeof(uint64_t), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(s)), "hipMemcpyAsync");
    y.set_ntt(false);
}

}  // namespace fhe
}  // namespace deeppowers
