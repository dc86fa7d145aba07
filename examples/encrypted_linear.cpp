// encrypted_linear.cpp - an encrypted linear layer at the shape of BASELINE configs[2] (hidden = 768 outputs, 64 encrypted
// inputs), the operation that replaces the reference's plaintext matvec kernels
// (/root/reference/src/core/execution/models/gpt_model.cpp:793 qkv_transform, :848 ffn, :883 compute_logits).
// Packing: one ciphertext per input feature, one SAMPLE per polynomial coefficient (N = 4096 samples ride along), so
// y_i = sum_j w_ij x_j with integer plaintext weights is exactly Evaluator::matvec_scalar.
#include <cstdio>
#include <vector>

#include "deeppowers/fhe.hpp"

using namespace deeppowers::fhe;

int main() {
    try {
        const FheParams params = FheParams::n4096_l4();
        const size_t n = params.n(), rows = 768, cols = 64;
        Context ctx(params, 0);
        Evaluator evaluator(ctx);
        KeyGenerator keygen(ctx);
        Encryptor encryptor(ctx, keygen.secret_key());
        Decryptor decryptor(ctx, keygen.secret_key());

        // activations: cols features x n samples, small integers; weights: rows x cols small integers
        std::vector<int64_t> x(cols * n), w(rows * cols), y(rows * n);
        uint64_t s = 7;
        auto rnd = [&](int span) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (int64_t)((s >> 33) % (2 * span + 1)) - span; };
        for (auto& v : x) v = rnd(127);
        for (auto& v : w) v = rnd(127);

        const unsigned log2_scale = 30;
        Ciphertext ct_x(ctx, 2, cols), ct_y(ctx, 2, rows);
        encryptor.encrypt(x.data(), log2_scale, ct_x);
        ScalarMatrix W(ctx, rows, cols);
        W.set(w.data());
        evaluator.matvec_scalar(W, ct_x, ct_y);          // 768 x 64 encrypted matvec, one HIP kernel
        decryptor.decrypt(ct_y, log2_scale, y.data());

        size_t bad = 0;
        for (size_t i = 0; i < rows; ++i)
            for (size_t k = 0; k < n; k += 97) {           // spot-check samples
                int64_t ref = 0;
                for (size_t j = 0; j < cols; ++j) ref += w[i * cols + j] * x[j * n + k];
                bad += (y[i * n + k] != ref);
            }
        std::printf("encrypted 768x64 linear layer over %zu samples: %s\n", n, bad ? "MISMATCH" : "OK");
        return bad ? 1 : 0;
    } catch (const Exception& e) {
        std::fprintf(stderr, "deeppowers::fhe error %d: %s\n", (int)e.code(), e.what());
        return 2;
    }
}
