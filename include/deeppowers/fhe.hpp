// deeppowers/fhe.hpp - C++ operator API of the FHE ciphertext-arithmetic hot path.
//
// The reference has no Ciphertext/Evaluator to keep, so this is the BUILD-SPEC
// API of SURVEY.md section 8(a)/(b), written in the reference's house style so that it reads like the
// rest of deeppowers:
//   - namespace deeppowers                     (/root/reference/src/core/hal/hal.hpp:8)
//   - pimpl with std::unique_ptr<Impl>         (/root/reference/src/api/cpp/include/deeppowers.hpp:73-75)
//   - heavy objects are non-copyable           (/root/reference/src/core/execution/model.hpp:88-89)
//   - errors are exceptions carrying ErrorCode (/root/reference/src/common/error.hpp:42-53)
//   - device chosen by id                      (/root/reference/src/api/cpp/src/deeppowers.cpp:15)
//   - optional stream as the last argument     (/root/reference/src/core/hal/hal.hpp:95)
// Host code stays C++; every operation is one call through the C ABI of include/dpfhe.h into HIP kernels.
// An encrypted linear layer at the reference's matmul sites
// (/root/reference/src/core/execution/models/gpt_model.cpp:793,848,883) is Evaluator::matvec_plain.
#pragma once

#include <cstddef>
#include <cstdint>
#include <iosfwd>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace deeppowers {
namespace fhe {

// numbers of deeppowers::common::ErrorCode (/root/reference/src/common/error.hpp:10-40)
enum class ErrorCode { SUCCESS = 0, OUT_OF_MEMORY = 1001, DEVICE_ERROR = 1002, INVALID_ARGUMENT = 2000, INVALID_STATE = 2002, RUNTIME_ERROR = 3000 };

class Exception : public std::runtime_error {
public:
    Exception(ErrorCode code, const std::string& message) : std::runtime_error(message), code_(code) {}
    ErrorCode code() const { return code_; }

private:
    ErrorCode code_;
};

using Stream = void;  // a hipStream_t; nullptr = the null stream

struct FheParams {
    uint32_t log2_n = 0;
    std::vector<uint64_t> moduli;  // primes < 2^60, q = 1 (mod 2N)
    std::vector<uint64_t> psi;     // primitive 2N-th roots of unity
    size_t n() const { return size_t(1) << log2_n; }
    size_t n_limbs() const { return moduli.size(); }
    FheParams drop_last_limb() const;  // the level after a rescale
    static FheParams config1();     // N=1024, one 30-bit limb       (BASELINE.json configs[0])
    static FheParams n4096_l4();    // N=4096, 4 x 60-bit limbs      (configs[1..3], the metric)
    static FheParams n8192_l6();    // N=8192, 6 x 60-bit limbs      (configs[4] sizes)
    static FheParams n16384(size_t n_limbs);  // N=16384, the first n_limbs (<= 8) primes below 2^60 that are 1 mod 2^15: a ring with room for a security margin
    static FheParams n8192(size_t n_limbs);   // N=8192, the first n_limbs (<= 20) primes of the same descending chain: deeper levels, multiply workspaces
};

class PolyBuffer;
class Context {
public:
    explicit Context(const FheParams& params, int device_id = 0);
    ~Context();
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;

    const FheParams& params() const;
    int device_id() const;
    bool uses_fold() const;
    // arithmetic of limb i in the batched transforms and the fused multiply (dpfhe_ctx_limb_class): 0 Shoup, 1 fold, 2 f64, 3 fold-scaled
    int limb_class(uint32_t limb) const;
    // scratch arenas of the composed large-ring operations, one per stream (dpfhe_ctx_release_scratch / dpfhe_ctx_scratch_bytes)
    void release_scratch(void* stream = nullptr, bool all_streams = false);
    size_t scratch_bytes() const;
    void* handle() const;  // dpfhe_ctx*
    void synchronize() const;
    // set-up call: slice size (MiB of scratch) of the operations composed from the batched transforms at N >= 16384 (dpfhe_ctx_set_scratch_limit; default 1024)
    void set_scratch_limit(size_t mib);
    // Which form of the fused multiply Evaluator::multiply launches (include/dpfhe.h "A0, continued": the ring degree's default; autotune()
    // is the explicit opt-in measurement, in the spirit of the reference's AutoTuner, src/core/inference/auto_tuner.hpp:26-64).  Both forms give the same words.
    struct TuneInfo {
        std::string chosen, source;                                // e.g. "quad", "default"
        std::vector<std::pair<std::string, float>> probe_us;       // microseconds per probe launch of each measured form
        unsigned probe_pairs = 0, probe_reps = 0;
    };
    TuneInfo tune_info() const;
    TuneInfo autotune(PolyBuffer& scratch, unsigned reps = 3);    // re-measure on the caller's buffer (overwritten; >= 7 RNS polynomials)

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

// Device-resident words [batch][components][L][N]; owner of its buffer.
class PolyBuffer {
public:
    PolyBuffer(const Context& ctx, size_t batch, size_t components, bool is_ntt);
    ~PolyBuffer();
    PolyBuffer(PolyBuffer&&) noexcept;
    PolyBuffer& operator=(PolyBuffer&&) noexcept;
    PolyBuffer(const PolyBuffer&) = delete;
    PolyBuffer& operator=(const PolyBuffer&) = delete;

    uint64_t* data();
    const uint64_t* data() const;
    size_t batch() const;
    size_t size() const;   // components per item
    size_t words() const;  // batch * size * L * N
    bool is_ntt() const;
    void set_ntt(bool v);
    void copy_from_host(const uint64_t* src);  // words() canonical residues
    void copy_to_host(uint64_t* dst) const;
    // N4: wire format.  Little-endian: "DPFHEv1\0", u32 log2_n, u32 n_limbs, u64 batch, u64 components,
    // u32 is_ntt, u32 reserved, u64 moduli[n_limbs], then batch*components*n_limbs*N u64 words.  load() checks that the
    // header matches this buffer's context and shape (throws INVALID_ARGUMENT otherwise) and sets the domain flag.
    void save(std::ostream& os) const;
    void load(std::istream& is);

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

class Plaintext : public PolyBuffer {  // A5: one polynomial per item
public:
    explicit Plaintext(const Context& ctx, size_t batch = 1, bool is_ntt = false) : PolyBuffer(ctx, batch, 1, is_ntt) {}
};

class Ciphertext : public PolyBuffer {  // A4: size in {2, 3}
public:
    explicit Ciphertext(const Context& ctx, size_t size = 2, size_t batch = 1, bool is_ntt = false);
};

// Plaintext scalar weights of a linear layer: rows x cols integers, stored as one residue per limb ([rows][cols][L]).
class ScalarMatrix {
public:
    ScalarMatrix(const Context& ctx, size_t rows, size_t cols);
    ~ScalarMatrix();
    ScalarMatrix(const ScalarMatrix&) = delete;
    ScalarMatrix& operator=(const ScalarMatrix&) = delete;
    void set(const int64_t* weights);  // rows*cols signed integers, row-major
    size_t rows() const;
    size_t cols() const;
    const uint64_t* data() const;      // device

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

// N1: evaluation keys for relinearisation: L digits x 2 polynomials, NTT domain
// (evk_j = (-(a_j s) + e_j + g_j s^2, a_j), g_j = CRT basis element of limb j; layout [L][2][L][N]).
class RelinKeys : public PolyBuffer {
public:
    explicit RelinKeys(const Context& ctx);
};

// N3: switching key from sigma_g(s) to s for one Galois element g (odd, < 2N); same layout as RelinKeys.
class GaloisKeys : public PolyBuffer {
public:
    GaloisKeys(const Context& ctx, uint32_t galois_elt);
    uint32_t galois_elt() const { return galois_elt_; }

private:
    uint32_t galois_elt_;
};

class Evaluator {
public:
    explicit Evaluator(const Context& ctx);
    ~Evaluator();
    Evaluator(const Evaluator&) = delete;
    Evaluator& operator=(const Evaluator&) = delete;

    // A1 / A2 (in place; flips is_ntt)
    void transform_to_ntt_inplace(PolyBuffer& x, Stream* stream = nullptr) const;
    void transform_from_ntt_inplace(PolyBuffer& x, Stream* stream = nullptr) const;
    // A3 / A8 on whole buffers (same shape and domain)
    void add(const PolyBuffer& a, const PolyBuffer& b, PolyBuffer& out, Stream* stream = nullptr) const;
    void sub(const PolyBuffer& a, const PolyBuffer& b, PolyBuffer& out, Stream* stream = nullptr) const;
    void negate(const PolyBuffer& a, PolyBuffer& out, Stream* stream = nullptr) const;
    void dyadic_multiply(const PolyBuffer& a, const PolyBuffer& b, PolyBuffer& out, Stream* stream = nullptr) const;
    void dyadic_multiply_add(const PolyBuffer& a, const PolyBuffer& b, PolyBuffer& acc, Stream* stream = nullptr) const;
    // A6: the metric op.  out must be a 3-component ciphertext of the same batch; its domain flag selects OUT_NTT.
    void multiply(const Ciphertext& a, const Ciphertext& b, Ciphertext& out, Stream* stream = nullptr) const;
    // N1: 3 -> 2 components, coefficient domain in and out
    void relinearize(const Ciphertext& in3, const RelinKeys& keys, Ciphertext& out2, Stream* stream = nullptr) const;
    // N1, second half: out = round(in / q_last) at the next level; `out` must have been created on a Context of
    // params().drop_last_limb() with the same size and batch (coefficient domain)
    void rescale(const Ciphertext& in, Ciphertext& out_next_level, Stream* stream = nullptr) const;
    // N3: ciphertext of m(X) -> ciphertext of m(X^g) under the same key (automorphism + key switch), coefficient domain
    void apply_galois(const Ciphertext& in2, const GaloisKeys& keys, Ciphertext& out2, Stream* stream = nullptr) const;
    // A7: ct (.) pt per component (NTT domain) and the matrix-vector product y_i = sum_j W_ij (.) x_j
    void multiply_plain(const Ciphertext& a, const Plaintext& p, Ciphertext& out, Stream* stream = nullptr) const;
    void matvec_plain(const Plaintext& W /* batch = rows*cols */, const Ciphertext& x /* batch = cols */, Ciphertext& y /* batch = rows */,
                      Stream* stream = nullptr) const;
    // A7 with n_rhs right-hand sides: x batch = cols * n_rhs laid out [cols][n_rhs], y batch = rows * n_rhs laid out [rows][n_rhs]
    void matvec_plain_multi(const Plaintext& W, const Ciphertext& x, Ciphertext& y, size_t n_rhs, Stream* stream = nullptr) const;
    // A7, scalar weights: y_i = sum_j w_ij * x_j (either domain; y takes x's domain).  x: batch = cols, y: batch = rows.
    // With one ciphertext per input feature and one sample per coefficient this IS the encrypted linear layer that
    // replaces the plaintext matvec sites of the reference (gpt_model.cpp:793,848,883).
    void matvec_scalar(const ScalarMatrix& W, const Ciphertext& x, Ciphertext& y, Stream* stream = nullptr) const;
    // A8: modular sum over the batch -> one item
    void reduce_sum(const PolyBuffer& in, PolyBuffer& out, Stream* stream = nullptr) const;

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

// ---- round 4: EXACT ciphertext x ciphertext multiply (BFV-style) around the fused tensor-product kernel ---------------------------------
// For ciphertexts that carry their message as floor(q / t) * m (Encryptor::encrypt_exact: slot-packed exact arithmetic mod t) a tensor
// product modulo q is not enough: the product carries floor(q/t)^2 and has to be scaled by t / q over the INTEGERS.  ExactMultiplier does
// that with the level's modulus q = the first `level` limbs of a larger work context (the remaining limbs hold the integer product):
//     extend both operands exactly to all work limbs (dpfhe_base_extend)  ->  Evaluator::multiply on the work context (the fused
//     ct x ct kernel: the metric op)  ->  x t / q with rounding on the workspace limbs (dpfhe_scale_round)  ->  back to the level's limbs.
// Needs prod(work limbs) > 2 N t q^2 (N = 8192, t = 65537, a two-limb level: five 60-bit limbs).  The result has 3 components on the LEVEL
// context (relinearise there).  This is what lets a packed layer's output go through a polynomial activation (x -> x^2 between W_up and
// W_down: /root/reference/src/core/execution/models/gpt_model.cpp:842-859 with the square standing in for GELU).
class ExactMultiplier {
public:
    // level_ctx's moduli must be the first level_ctx.params().n_limbs() moduli of work_ctx, same ring degree, same device
    ExactMultiplier(const Context& work_ctx, const Context& level_ctx, uint64_t plain_modulus);
    ~ExactMultiplier();
    ExactMultiplier(const ExactMultiplier&) = delete;
    ExactMultiplier& operator=(const ExactMultiplier&) = delete;
    // a, b: 2 components on level_ctx (coefficient domain; may be the same object: squaring); out3: 3 components on level_ctx, same batch.
    // Enqueues on `stream`; scratch belongs to the object and grows to the largest batch seen (one multiply() at a time per object).
    void multiply(const Ciphertext& a, const Ciphertext& b, Ciphertext& out3, Stream* stream = nullptr);

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

// ---- (e) SURVEY.md section 8e: the one collective of the sharded path -------------------------------------------------
// One process per GPU.  Rank 0 obtains an id, ships its 128 bytes to the other ranks by the host program's own rendezvous
// (a file, a pipe, MPI ...), every rank constructs a Communicator.  all_gather is RCCL's all-gather over xGMI on the caller's
// stream: rank r's `send` lands at item range [r * send.batch(), (r + 1) * send.batch()) of `recv` on every rank.  Stands in for
// /root/reference/src/core/distributed/distributed_context.cpp:97-122 (ncclAllGather + stream create + synchronise per call).
class Communicator {
public:
    static std::vector<uint8_t> unique_id();   // 128 bytes
    Communicator(const std::vector<uint8_t>& id, int rank, int world_size, int device_id);
    ~Communicator();
    Communicator(const Communicator&) = delete;
    Communicator& operator=(const Communicator&) = delete;
    int rank() const;
    int world_size() const;
    // recv.batch() == world_size * send.batch(), same components; the domain flag is copied
    void all_gather(const PolyBuffer& send, PolyBuffer& recv, Stream* stream = nullptr) const;

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

// Randomness.  Keys, encryption randomness and errors are drawn from ChaCha20 keyed by the operating system's CSPRNG
// (getrandom(2)): uniform values by rejection sampling, ternary secrets, centred-binomial errors (sigma = 3.24).
// TestSeed selects a DETERMINISTIC, NON-CRYPTOGRAPHIC generator (SplitMix64) instead - for reproducible tests, examples
// and benchmarks ONLY: anyone who knows the seed (64 bits, invertible) knows the secret key and every encryption.
struct TestSeed {
    uint64_t value;
};

// ---- N2: secret key, encryption, decryption, key generation -------------------------------
// Host-side (client-side in the reference's story, /root/reference/README.md:57-60): sampling and CRT decoding run on
// the CPU, polynomial arithmetic goes through the same C ABI as everything else.  Symmetric RLWE:
//   ct = (c0, c1) = (-(a s) + e + 2^log2_scale * m,  a),   s ternary, e centred binomial (sigma = 3.24, |e| <= 21).
// Messages are integer polynomials (N signed coefficients per item); decryption returns round(phase / 2^log2_scale),
// where phase = c0 + c1 s (+ c2 s^2) is CRT-composed over all limbs and centred mod Q = prod q_i.
class SecretKey {
public:
    explicit SecretKey(const Context& ctx);            // fresh ternary secret from the OS CSPRNG
    SecretKey(const Context& ctx, TestSeed seed);      // deterministic - tests only
    SecretKey(const Context& ctx, const std::vector<int8_t>& ternary_coefficients);  // the same secret on another context
    ~SecretKey();
    SecretKey(const SecretKey&) = delete;
    SecretKey& operator=(const SecretKey&) = delete;
    const std::vector<int8_t>& coefficients() const;  // ternary s
    const uint64_t* ntt() const;                        // device, [L][N]: NTT(s)
    const uint64_t* ntt_squared() const;                // device, [L][N]: NTT(s)^2 = NTT(s^2)

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

// Public key: pk = (-(a s) + e, a), stored in the NTT domain ([1][2][L][N]).  Whoever holds it can encrypt; only the
// secret key decrypts (the reference's story keeps the secret key with the client, README.md:57-60).
class PublicKey : public PolyBuffer {
public:
    explicit PublicKey(const Context& ctx) : PolyBuffer(ctx, 1, 2, /*is_ntt=*/true) {}
};

class KeyGenerator {
public:
    explicit KeyGenerator(const Context& ctx);          // secret key and all key randomness from the OS CSPRNG
    KeyGenerator(const Context& ctx, TestSeed seed);    // deterministic - tests only
    ~KeyGenerator();
    KeyGenerator(const KeyGenerator&) = delete;
    KeyGenerator& operator=(const KeyGenerator&) = delete;
    const SecretKey& secret_key() const;
    void create_relin_keys(RelinKeys& out);  // evk_j = (-(a_j s) + e_j + g_j s^2, a_j), NTT domain
    void create_galois_keys(GaloisKeys& out);  // key_j = (-(a_j s) + e_j + g_j sigma_g(s), a_j) for out.galois_elt()
    void create_public_key(PublicKey& out);    // (-(a s) + e, a), NTT domain

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

class Encryptor {
public:
    Encryptor(const Context& ctx, const SecretKey& sk);   // symmetric: c1 uniform, c0 = -(c1 s) + e + scale m
    Encryptor(const Context& ctx, const SecretKey& sk, TestSeed seed);   // deterministic - tests only
    // public-key: (c0, c1) = (u pk0 + e1 + scale m, u pk1 + e2) with u ternary, e1, e2 small; decrypts under the same secret
    Encryptor(const Context& ctx, const PublicKey& pk);
    Encryptor(const Context& ctx, const PublicKey& pk, TestSeed seed);   // deterministic - tests only
    ~Encryptor();
    Encryptor(const Encryptor&) = delete;
    Encryptor& operator=(const Encryptor&) = delete;
    // messages: out.batch() * N signed coefficients; out: 2-component ciphertext(s), coefficient domain
    void encrypt(const int64_t* messages, unsigned log2_scale, Ciphertext& out);
    // exact integer arithmetic mod a plaintext modulus t (BFV-style scaling): c0 = -(a s) + e + floor(Q/t) * m
    void encrypt_exact(const int64_t* messages, uint64_t plain_modulus, Ciphertext& out);

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

class Decryptor {
public:
    Decryptor(const Context& ctx, const SecretKey& sk);
    ~Decryptor();
    Decryptor(const Decryptor&) = delete;
    Decryptor& operator=(const Decryptor&) = delete;
    // ct: 2 or 3 components, coefficient domain.  messages_out: ct.batch() * N values round(phase / 2^log2_scale);
    // throws RUNTIME_ERROR if a value does not fit 62 bits (wrong scale or noise overflow).
    void decrypt(const Ciphertext& ct, unsigned log2_scale, int64_t* messages_out);
    // inverse of encrypt_exact: messages_out[k] = round(t * phase_k / Q) mod t, in [0, t)
    void decrypt_exact(const Ciphertext& ct, uint64_t plain_modulus, uint64_t* messages_out);
    // Remaining noise budget of encrypt_exact-style ciphertexts in bits (the smallest over all items and coefficients):
    // log2(Q / 2) - log2 |t * phase - Q * round(t * phase / Q)|, exact big-integer arithmetic.  Decryption is correct while it is > 0;
    // a fresh ciphertext at N = 8192 with five 60-bit limbs and t = 65537 has ~ 270 bits.  Host-side (whoever holds the secret key).
    double noise_budget_bits(const Ciphertext& ct, uint64_t plain_modulus);

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

// Hybrid key switching with one special prime P (SURVEY.md 8f N1 "base extension"): keys live on the extended
// context (data moduli + P) and the switching noise drops from ~ L N q sigma to ~ L N q sigma / P.
class HybridKeySwitcher {
public:
    // data_ctx: the level the ciphertexts live on; (special_prime, special_psi): one more NTT prime and its 2N-th root
    HybridKeySwitcher(const Context& data_ctx, const SecretKey& sk, uint64_t special_prime, uint64_t special_psi);
    HybridKeySwitcher(const Context& data_ctx, const SecretKey& sk, uint64_t special_prime, uint64_t special_psi, TestSeed seed);   // tests only
    ~HybridKeySwitcher();
    HybridKeySwitcher(const HybridKeySwitcher&) = delete;
    HybridKeySwitcher& operator=(const HybridKeySwitcher&) = delete;
    void add_galois_element(uint32_t galois_elt);  // generates and keeps the key for sigma_g
    // 3 -> 2 components, coefficient domain, on data_ctx
    void relinearize(const Ciphertext& in3, Ciphertext& out2, Stream* stream = nullptr) const;
    // ciphertext of m(X) -> ciphertext of m(X^g); the element must have been added
    void apply_galois(const Ciphertext& in2, uint32_t galois_elt, Ciphertext& out2, Stream* stream = nullptr) const;
    // many rotations in one pass (dpfhe_rotate_hybrid_batch): out item out_first + i = sigma_{elts[i]} of in item i, or of THE
    // item of `in` when it holds one.  All elements must have been added; the keys of a given element list are packed
    // back to back once and cached.
    void apply_galois_many(const Ciphertext& in2, const std::vector<uint32_t>& galois_elts, Ciphertext& out2, size_t out_first = 0,
                           Stream* stream = nullptr) const;
    // same on item ranges: out item out_first + i = sigma_{elts[i]} of in item in_first + i (or of in item in_first when `broadcast`).
    // `in2` and `out2` may be the same buffer as long as the two ranges do not overlap.
    void apply_galois_range(const Ciphertext& in2, size_t in_first, bool broadcast, const std::vector<uint32_t>& galois_elts, Ciphertext& out2,
                            size_t out_first, Stream* stream = nullptr) const;

    // HOISTED: many rotations of ONE item (in2 item in_item): the digit decomposition of c1 and its forward transforms are done
    // once, each rotation is a permutation in the NTT domain + its key inner product (dpfhe_rotate_hybrid_hoisted).  A valid key
    // switch of sigma_g(ct), not word-identical to apply_galois_many (the automorphism acts after the lift, not before it).
    // n_items inputs (in2 items in_first ...) share the rotations; the output is ROTATION-MAJOR: item out_first + r * n_items + t.
    void apply_galois_hoisted(const Ciphertext& in2, size_t in_first, size_t n_items, const std::vector<uint32_t>& galois_elts, Ciphertext& out2,
                              size_t out_first, Stream* stream = nullptr) const;
    // k * group items, item i rotated by galois_elts[i / group] (the giant steps of `group` tokens; keys read once per element)
    void apply_galois_grouped(const Ciphertext& in2, size_t in_first, const std::vector<uint32_t>& galois_elts, size_t group, Ciphertext& out2,
                              size_t out_first, Stream* stream = nullptr) const;

    // ---- division by P deferred ("double hoisting"; include/dpfhe.h "N3, round 3") -------------------------------------
    // The context of the data moduli + the special prime: buffers holding terms over Q P live on it.
    const Context& extended_context() const;
    // n_items inputs (in2 items in_first ...) -> out_qp items out_first + r * n_items + t, r = 0 .. elts.size(): r = 0 is P * ct_t, r >= 1 is
    // P * sigma_{elts[r-1]}(ct_t) + its key-switching term; NTT domain over Q P, NOT divided by P (dpfhe_rotate_hoisted_qp).
    // `out_qp`: 2-component buffer on extended_context().
    void rotate_hoisted_qp(const Ciphertext& in2, size_t in_first, size_t n_items, const std::vector<uint32_t>& galois_elts, PolyBuffer& out_qp,
                           size_t out_first, Stream* stream = nullptr) const;
    // elts.size() * group items (in2 items in_first ...), item i with the key of elts[i / group]: out_qp item out_first + i = the key
    // inner product of item i's c1 alone, NTT domain over Q P (dpfhe_switch_key_qp); the caller sums such terms, transforms back
    // once and finishes with dpfhe_rescale_bsgs.  The items must already carry the automorphism (coefficient domain).
    void switch_key_qp(const Ciphertext& in2, size_t in_first, const std::vector<uint32_t>& galois_elts, size_t group, PolyBuffer& out_qp,
                       size_t out_first, Stream* stream = nullptr) const;
    // One HybridKeySwitcher serves ONE stream at a time: its scratch buffers and its cache of packed keys are shared by every call
    // (and by every PackedLinear built on it); calls on different streams must be ordered by the caller.

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

// ---- N3: slot packing and the packed matrix-vector product ------------------------------------
// BatchEncoder: N slots of Z_t (t prime, t = 1 mod 2N) arranged as 2 rows x N/2.  Slot i of row 0 is the value of the
// message polynomial at zeta^(3^i), row 1 at zeta^(-3^i) (zeta a primitive 2N-th root mod t), so the automorphism
// X -> X^(3^s) rotates both rows LEFT by s slots and X -> X^(2N-1) swaps the rows.  Host-side (client-side) code.
class BatchEncoder {
public:
    explicit BatchEncoder(const Context& ctx, uint64_t plain_modulus = 65537);
    ~BatchEncoder();
    BatchEncoder(const BatchEncoder&) = delete;
    BatchEncoder& operator=(const BatchEncoder&) = delete;
    uint64_t plain_modulus() const;
    size_t slot_count() const;   // N
    size_t row_size() const;     // N / 2
    void encode(const uint64_t* slots /* N values < t: row 0 then row 1 */, int64_t* coeffs_out /* N, centred mod t */) const;
    void decode(const uint64_t* coeffs_mod_t /* N */, uint64_t* slots_out /* N */) const;
    uint32_t galois_element(int left_rotation) const;   // 3^s mod 2N (negative s rotates right)

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

// PackedLinear: encrypted y = W x over Z_t for an out_dim x in_dim matrix in slot packing - the encrypted replacement of the
// dense layers at the reference's matmul sites for a single token (/root/reference/src/core/execution/models/gpt_model.cpp:793
// QKV 768 -> 2304, :848 FFN 768 -> 3072 -> 768, :883 LM head 768 -> 50257; dims /root/reference/src/core/execution/model.hpp:47-50).
//
// Input packing: the vector repeats with period n = 2^ceil(log2 in_dim) along both slot rows (pack_input), so a ciphertext
// holds N / n identical windows.  Diagonal method with baby-step / giant-step rotations over m diagonals:
//   sum_i rot_{i n1}( sum_j rot_{-i n1}(diag_{i n1 + j}) (.) rot_j(x) ),   n1 n2 = m,
//   * out_dim >= n  (m = n): every window computes a DIFFERENT block of n output rows, so one ciphertext delivers N outputs and
//     ceil(out_dim / N) output ciphertexts share the n1 - 1 baby-step rotations; each costs n2 - 1 giant-step rotations;
//   * out_dim <  n  (m = 2^ceil(log2 out_dim)): only the m wrapped diagonals diag_k[r] = W[r mod m][(r + k) mod n] are used and the
//     n / m partial sums are folded with log2(n / m) more rotations; the result repeats with period m (it is a valid input
//     of the next layer);
//   * one block (out_dim <= m = n): every window computes the same block, the result repeats with period n.
// Per application (round 3: the division by P is deferred - "double hoisting"): ONE hoisted rotation pass for the baby steps whose
// results stay in the NTT domain over Q P (dpfhe_rotate_hoisted_qp: gathers + key inner products, no transform), ONE
// dpfhe_matvec_plain_multi over all pre-transformed diagonals (encoded over Q P) of all output ciphertexts, one inverse transform
// that applies the giant-step automorphisms as it loads (dpfhe_ntt_inv_galois) + one divide-by-P pass over the inner sums, then per
// output ciphertext the giant steps' key inner products (dpfhe_switch_key_qp: Ld transforms each instead of Ld + 2), their sum,
// ONE inverse transform and ONE divide-by-P + add (dpfhe_rescale_bsgs).
class PackedLinear {
public:
    // W: out_dim * in_dim values < t, row-major.  The needed Galois keys are added to `ks`.
    // tokens_per_ciphertext = 2 (round 6): the two slot ROWS of a ciphertext carry two different tokens - row 0 repeats token A's vector with period n, row 1
    // token B's.  Every rotation of the diagonal method is a row rotation and the diagonals are the same for both rows, so the layer computes W x_A in row 0 and
    // W x_B in row 1 with the very same kernels: half the cost per token.  It needs the layer's output blocks to fit the windows of ONE row
    // (ceil(out_dim / n) <= N / (2 n) for one output ciphertext: GPT-2 small's layers do at N = 8192 and N = 16384); pack_input_rows / unpack_output_rows.
    PackedLinear(const Context& data_ctx, const BatchEncoder& enc, HybridKeySwitcher& ks, const uint64_t* W, size_t out_dim, size_t in_dim, size_t tokens_per_ciphertext = 1);
    // square d x d (d a power of two dividing N/2)
    PackedLinear(const Context& data_ctx, const BatchEncoder& enc, HybridKeySwitcher& ks, const uint64_t* W, size_t d);
    ~PackedLinear();
    PackedLinear(const PackedLinear&) = delete;
    PackedLinear& operator=(const PackedLinear&) = delete;
    size_t dim() const;                 // m: diagonals per output ciphertext
    size_t in_dim() const;
    size_t out_dim() const;
    size_t input_period() const;        // n
    size_t output_ciphertexts() const;  // batch of y
    size_t baby_steps() const;
    size_t giant_steps() const;
    size_t key_switches_per_apply() const;
    // slot vectors (N values each) <-> plain vectors: x (in_dim values) -> the N slots to encode and encrypt;
    // output_ciphertexts() * N decoded slots -> y (out_dim values)
    void pack_input(const uint64_t* x, uint64_t* slots) const;
    void unpack_output(const uint64_t* slots, uint64_t* y) const;
    size_t tokens_per_ciphertext() const;
    // two tokens per ciphertext: x0 -> slot row 0, x1 -> slot row 1 (each repeated with the input period); and back from output_ciphertexts() * N slots
    void pack_input_rows(const uint64_t* x0, const uint64_t* x1, uint64_t* slots) const;
    void unpack_output_rows(const uint64_t* slots, uint64_t* y0, uint64_t* y1) const;
    // x: T items (T tokens, each packed with pack_input), y: output_ciphertexts() * T items, output ciphertext o of token t at item
    // o * T + t; 2 components, coefficient domain.  All tokens share ONE pass over the rotation keys and the diagonals (hoisted
    // baby steps per token, one multi-right-hand-side matvec, the giant steps' key inner products summed before ONE division by P).
    // Enqueues on `stream` and returns (no host synchronisation).  Scratch: the layer's own buffers (they grow only when a larger T
    // than ever before arrives) AND the key switcher's (digit images, packed keys) - so one apply() at a time per LAYER, and all layers
    // built on one HybridKeySwitcher share ONE stream at a time; synchronise before reading y on the host.
    void apply(const Ciphertext& x, Ciphertext& y, Stream* stream = nullptr) const;

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

// PackedSelect: `length` consecutive slots of slot row 0 of a packed ciphertext, starting at `offset`, re-packed as a vector that
// repeats with period `period` (a power of two >= length dividing N/2) along BOTH slot rows - the input packing of a PackedLinear.
// The on-device hand-over between two packed layers when the next layer consumes a SLICE of the previous one's output (the V third of
// a fused QKV product feeding the attention-output projection): one rotation by `offset`, one plaintext mask product (it zeroes
// everything else; costs one multiplicative level of noise, like a layer), log2(N/2 / period) rotate-and-add steps, one row swap.
class PackedSelect {
public:
    // tokens_per_ciphertext = 2: the slice is taken in BOTH slot rows (each row carries its own token) and re-packed inside its row - no row swap at the end
    PackedSelect(const Context& data_ctx, const BatchEncoder& enc, HybridKeySwitcher& ks, size_t offset, size_t length, size_t period, size_t tokens_per_ciphertext = 1);
    ~PackedSelect();
    PackedSelect(const PackedSelect&) = delete;
    PackedSelect& operator=(const PackedSelect&) = delete;
    size_t key_switches_per_apply() const;
    // x, y: T items, 2 components, coefficient domain; enqueues on `stream` (scratch belongs to the object: one apply() at a time)
    void apply(const Ciphertext& x, Ciphertext& y, Stream* stream = nullptr) const;

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

// PackedTransformerBlock: the LINEAR SKELETON of one GPT-2 block on encrypted, slot-packed hidden states - every dense site of
// /root/reference/src/core/execution/models/gpt_model.cpp:786-859 chained on the device for T tokens:
//     qkv = W_qkv x                      (:793, d -> 3 d, one fused product)
//     a   = v                            attention over ONE position: softmax over a single key is 1, so the attention output IS v
//     h1  = x + W_o a                    attention-output projection + residual
//     h2  = h1 + W_down (W_up h1)        (:848, d -> h -> d) + residual
// over Z_t.  LayerNorm, GELU and the softmax over longer contexts are the non-linear parts an FHE forward cannot take as is
// (SURVEY.md section 7): they are the identity here.  Five multiplicative levels (four matrices + the mask of the v hand-over); the
// caller can watch the budget with Decryptor::noise_budget_bits.  Tokens are independent: a multi-rank driver gives each rank its own
// tokens and needs no collective before the final gather (examples/encrypted_gpt2_block.cpp).
class PackedTransformerBlock {
public:
    // row-major weights with entries < t: W_qkv (3 d x d, rows [q | k | v]), W_o (d x d), W_up (h x d), W_down (d x h)
    PackedTransformerBlock(const Context& data_ctx, const BatchEncoder& enc, HybridKeySwitcher& ks, const uint64_t* W_qkv, const uint64_t* W_o,
                           const uint64_t* W_up, const uint64_t* W_down, size_t d, size_t h);
    ~PackedTransformerBlock();
    PackedTransformerBlock(const PackedTransformerBlock&) = delete;
    PackedTransformerBlock& operator=(const PackedTransformerBlock&) = delete;
    size_t hidden() const;
    size_t inner() const;
    size_t key_switches_per_token() const;
    void pack_input(const uint64_t* x /* d values */, uint64_t* slots /* N */) const;
    void unpack_output(const uint64_t* slots /* N */, uint64_t* y /* d values */) const;
    // x, y: T items (tokens), 2 components, coefficient domain, packed as pack_input packs; y is packed the same way (it can enter the
    // next block as it is).  Enqueues on `stream`; one apply() at a time per object.
    void apply(const Ciphertext& x, Ciphertext& y, Stream* stream = nullptr) const;
    // intermediate results of the LAST apply() (T items each; valid until the next one): 0 = qkv (slot r of row 0 = output r), 1 = a = v in
    // the input packing, 2 = h1, 3 = W_up h1 in the input packing of W_down, 4 = h2 (= y)
    const Ciphertext& stage(int index) const;

private:
    class Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace fhe
}  // namespace deeppowers
