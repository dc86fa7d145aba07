// fhe_api.cpp - deeppowers::fhe facade over the C ABI (include/dpfhe.h).  Plain C++17 (g++); the only
// HIP it touches is the runtime API for buffer ownership, like the reference's HAL device
// (/root/reference/src/core/hal/cuda/cuda_device.cpp:9-16 turns runtime errors into exceptions).
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

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
const uint64_t kPrimes60[6][3] = {
    {1152921504606830593ull, 116777451583545ull, 25959043411404ull},  {1152921504606748673ull, 271802498405390ull, 100406242475323ull},
    {1152921504606683137ull, 134367042585739ull, 45474351589225ull},  {1152921504606601217ull, 276147373136904ull, 92707844590835ull},
    {1152921504606584833ull, 317490233586139ull, 23981819781494ull},  {1152921504606109697ull, 279138086580908ull, 253932030982881ull}};
}  // namespace

FheParams FheParams::config1() { return FheParams{10, {1073707009ull}, {169871ull}}; }
FheParams FheParams::n4096_l4() {
    FheParams p{12, {}, {}};
    for (int i = 0; i < 4; ++i) { p.moduli.push_back(kPrimes60[i][0]); p.psi.push_back(kPrimes60[i][1]); }
    return p;
}
FheParams FheParams::n8192_l6() {
    FheParams p{13, {}, {}};
    for (int i = 0; i < 6; ++i) { p.moduli.push_back(kPrimes60[i][0]); p.psi.push_back(kPrimes60[i][2]); }
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
void* Context::handle() const { return impl_->h; }
void Context::synchronize() const {
    hip_check(hipSetDevice(impl_->device_id), "hipSetDevice");
    hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
}

// ---- PolyBuffer -----------------------------------------------------------------------------------------
class PolyBuffer::Impl {
public:
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

Ciphertext::Ciphertext(const Context& ctx, size_t size, size_t batch, bool is_ntt) : PolyBuffer(ctx, batch, size, is_ntt) {
    if (size != 2 && size != 3) throw Exception(ErrorCode::INVALID_ARGUMENT, "Ciphertext: size must be 2 or 3");
}

RelinKeys::RelinKeys(const Context& ctx) : PolyBuffer(ctx, ctx.params().n_limbs(), 2, /*is_ntt=*/true) {}

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
    const uint32_t flags = (a.is_ntt() ? DPFHE_IN_NTT : 0u) | (out.is_ntt() ? DPFHE_OUT_NTT : 0u);
    check(dpfhe_ct_mul(impl_->h(), out.data(), a.data(), b.data(), a.batch(), flags, s), "dpfhe_ct_mul");
}
void Evaluator::relinearize(const Ciphertext& in3, const RelinKeys& keys, Ciphertext& out2, Stream* s) const {
    if (in3.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "relinearize: input must be in the coefficient domain");
    if (in3.size() != 3 || out2.size() != 2 || out2.batch() != in3.batch())
        throw Exception(ErrorCode::INVALID_ARGUMENT, "relinearize: 3-component input, 2-component output of the same batch");
    check(dpfhe_relinearize(impl_->h(), out2.data(), in3.data(), keys.data(), in3.batch(), s), "dpfhe_relinearize");
    out2.set_ntt(false);
}
void Evaluator::multiply_plain(const Ciphertext& a, const Plaintext& p, Ciphertext& out, Stream* s) const {
    if (!a.is_ntt() || !p.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "multiply_plain: operands must be in the NTT domain");
    if (p.batch() != 1 || out.words() != a.words()) throw Exception(ErrorCode::INVALID_ARGUMENT, "multiply_plain: one plaintext, output shaped like the input");
    const size_t per_poly = p.words();  // L*N
    for (size_t i = 0; i < a.batch() * a.size(); ++i)
        check(dpfhe_dyadic_mul(impl_->h(), out.data() + i * per_poly, a.data() + i * per_poly, p.data(), 1, s), "dpfhe_dyadic_mul");
    out.set_ntt(true);
}
void Evaluator::matvec_plain(const Plaintext& W, const Ciphertext& x, Ciphertext& y, Stream* s) const {
    if (!W.is_ntt() || !x.is_ntt()) throw Exception(ErrorCode::INVALID_STATE, "matvec_plain: operands must be in the NTT domain");
    const size_t cols = x.batch(), rows = y.batch();
    if (x.size() != 2 || y.size() != 2 || cols == 0 || W.batch() != rows * cols) throw Exception(ErrorCode::INVALID_ARGUMENT, "matvec_plain: W batch must be rows*cols, x/y 2-component");
    check(dpfhe_matvec_plain(impl_->h(), y.data(), W.data(), x.data(), rows, cols, s), "dpfhe_matvec_plain");
    y.set_ntt(true);
}
void Evaluator::reduce_sum(const PolyBuffer& in, PolyBuffer& out, Stream* s) const {
    if (out.batch() != 1 || out.size() != in.size()) throw Exception(ErrorCode::INVALID_ARGUMENT, "reduce_sum: output must be one item of the same size");
    check(dpfhe_reduce_sum(impl_->h(), out.data(), in.data(), in.batch(), in.size(), s), "dpfhe_reduce_sum");
    out.set_ntt(in.is_ntt());
}

}  // namespace fhe
}  // namespace deeppowers
