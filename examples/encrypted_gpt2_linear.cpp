// encrypted_gpt2_linear.cpp - one token through a GPT-2-small dense layer under encryption, slot-packed (N3 of SURVEY.md 8f):
// N = 8192, six pinned 60-bit primes (five carry the data, the sixth is the special prime of hybrid key switching), hidden size
// 768 padded to the power of two 1024, y = W x over Z_65537 by the diagonal method with 32 baby and 32 giant steps.
// The layer shapes are the reference's matmul sites (/root/reference/src/core/execution/models/gpt_model.cpp:793 QKV,
// :848 FFN, :883 logits; hidden_size 768 at execution/model.hpp:47-50): a 768 x 768 block here, the attention output
// projection; QKV / FFN / LM head are row blocks of the same operator sharing the baby-step rotations.
//   usage: encrypted_gpt2_linear [d=1024] [reps=2]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <deeppowers/fhe.hpp>

using namespace deeppowers::fhe;

int main(int argc, char** argv) {
    const size_t d = argc > 1 ? (size_t)std::atol(argv[1]) : 1024, hidden = d >= 768 ? 768 : d;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 2;
    try {
        FheParams p = FheParams::n8192_l6();
        const uint64_t special = p.moduli.back(), special_psi = p.psi.back();
        p.moduli.pop_back(); p.psi.pop_back();
        const size_t n = p.n(), row = n / 2;
        Context ctx(p, 0);
        KeyGenerator kg(ctx);   // OS CSPRNG (TestSeed{..} would make the run reproducible)
        Encryptor enc(ctx, kg.secret_key());
        Decryptor dec(ctx, kg.secret_key());
        BatchEncoder be(ctx, 65537);
        const uint64_t t = be.plain_modulus();
        HybridKeySwitcher hks(ctx, kg.secret_key(), special, special_psi);

        // 8-bit quantised weights and activations (the reference's INT8 path: src/core/quantization), zero-padded to d
        uint64_t s = 5;
        auto rnd = [&](uint64_t m) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (s >> 33) % m; };
        std::vector<uint64_t> W(d * d, 0), x(d, 0), want(d, 0);
        for (size_t r = 0; r < hidden; ++r)
            for (size_t c = 0; c < hidden; ++c) W[r * d + c] = (t + rnd(255) - 127) % t;   // [-127, 127] mod t
        for (size_t c = 0; c < hidden; ++c) x[c] = (t + rnd(255) - 127) % t;
        for (size_t r = 0; r < d; ++r) {
            unsigned __int128 acc = 0;
            for (size_t c = 0; c < d; ++c) acc += (unsigned __int128)W[r * d + c] * x[c];
            want[r] = (uint64_t)(acc % t);
        }

        auto t0 = std::chrono::steady_clock::now();
        PackedLinear layer(ctx, be, hks, W.data(), d);     // encodes + transforms d diagonals, generates the rotation keys
        const double setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

        std::vector<uint64_t> slots(n), dm(n), got(n);
        std::vector<int64_t> coeffs(n);
        for (size_t r = 0; r < row; ++r) slots[r] = slots[row + r] = x[r % d];
        be.encode(slots.data(), coeffs.data());
        Ciphertext cx(ctx, 2, 1), cy(ctx, 2, 1);
        enc.encrypt_exact(coeffs.data(), t, cx);

        layer.apply(cx, cy);                                // warm-up (code objects, allocator)
        ctx.synchronize();
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) layer.apply(cx, cy);
        ctx.synchronize();
        const double apply_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3 / reps;

        dec.decrypt_exact(cy, t, dm.data());
        be.decode(dm.data(), got.data());
        size_t bad = 0;
        for (size_t r = 0; r < row; ++r) bad += (got[r] != want[r % d]) + (got[row + r] != want[r % d]);
        std::printf("encrypted %zux%zu layer (hidden %zu), N=%zu, %zu data limbs + special prime, t=%llu: %zu baby + %zu giant steps\n", d, d, hidden, n,
                    p.n_limbs(), (unsigned long long)t, layer.baby_steps(), layer.giant_steps());
        std::printf("setup (encode + NTT of %zu diagonals, %zu rotation keys): %.2f s;  apply: %.2f ms per token  (%zu key switches, 1 matvec_plain %zux%zu)\n", d,
                    layer.baby_steps() + layer.giant_steps() - 2, setup_s, apply_ms, layer.baby_steps() + layer.giant_steps() - 2, layer.giant_steps(), layer.baby_steps());
        std::printf(bad ? "MISMATCH in %zu slots\n" : "decrypted result equals W x mod t in all %zu slots: OK\n", bad ? bad : n);
        return bad ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
}
