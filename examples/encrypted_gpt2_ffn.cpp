// encrypted_gpt2_ffn.cpp - the linear path of one GPT-2-small feed-forward block under encryption, layers CHAINED on the device:
//     y = x + W_down (W_up x)   over Z_65537,   W_up: 768 -> 3072, W_down: 3072 -> 768
// (/root/reference/src/core/execution/models/gpt_model.cpp:848 FFN 768 -> 3072 -> 768 and the residual connection around it; the GELU
// between the two matrices is the non-linear part an FHE forward cannot take as is - SURVEY.md section 7 - so this is the block's
// linear skeleton.)  N = 8192, five 60-bit data primes + the special prime of hybrid key switching; several tokens per application.
// What the example shows beyond encrypted_gpt2_linear: the OUTPUT packing of one PackedLinear is turned into the INPUT packing of the
// next one on the device (one row-swap rotation + add), the result of the second layer is already a valid input again (the residual
// is a plain ciphertext add), and the noise of two chained plaintext products + 2 x 62 key switches stays far below the budget.
//   usage: encrypted_gpt2_ffn [tokens = 4] [reps = 3] [json | text]
// WHAT THIS IS NOT: only the FFN's two dense layers with the residual (x + W_down (W_up x)) - no activation at all, no LayerNorm.
// SECURITY: N = 8192 with 5 data primes + special prime = 360 bits under key switching; 128-bit security at N = 8192 allows 218 (Homomorphic Encryption
// Standard, ternary secret, sigma 3.2).  BASELINE configs[4]'s performance shape, not a deployable parameter set (encrypted_gpt2_linear / _block_act ... 14 run N = 16384).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <deeppowers/fhe.hpp>

using namespace deeppowers::fhe;

int main(int argc, char** argv) {
    const size_t T = argc > 1 ? (size_t)std::atol(argv[1]) : 4;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 3;
    const bool json = argc > 3 && !std::strcmp(argv[3], "json");
    const size_t d = 768, h = 3072;
    try {
        FheParams p = FheParams::n8192_l6();
        const uint64_t special = p.moduli.back(), special_psi = p.psi.back();
        p.moduli.pop_back(); p.psi.pop_back();
        const size_t n = p.n();
        Context ctx(p, 0);
        Evaluator ev(ctx);
        KeyGenerator kg(ctx);
        Encryptor enc(ctx, kg.secret_key());
        Decryptor dec(ctx, kg.secret_key());
        BatchEncoder be(ctx, 65537);
        const uint64_t t = be.plain_modulus();
        HybridKeySwitcher hks(ctx, kg.secret_key(), special, special_psi);

        uint64_t s = 2024;
        auto rnd = [&](uint64_t m) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (s >> 33) % m; };
        std::vector<uint64_t> Wu(h * d), Wd(d * h), x(T * d), mid(T * h), want(T * d);
        for (auto& v : Wu) v = (t + rnd(255) - 127) % t;     // 8-bit quantised weights and activations, as in encrypted_gpt2_linear
        for (auto& v : Wd) v = (t + rnd(255) - 127) % t;
        for (auto& v : x) v = (t + rnd(255) - 127) % t;
        for (size_t tk = 0; tk < T; ++tk) {
            for (size_t r = 0; r < h; ++r) {
                unsigned __int128 acc = 0;
                for (size_t c = 0; c < d; ++c) acc += (unsigned __int128)Wu[r * d + c] * x[tk * d + c];
                mid[tk * h + r] = (uint64_t)(acc % t);
            }
            for (size_t r = 0; r < d; ++r) {
                unsigned __int128 acc = x[tk * d + r];
                for (size_t c = 0; c < h; ++c) acc += (unsigned __int128)Wd[r * h + c] * mid[tk * h + c];
                want[tk * d + r] = (uint64_t)(acc % t);
            }
        }

        auto t0 = std::chrono::steady_clock::now();
        PackedLinear up(ctx, be, hks, Wu.data(), h, d), down(ctx, be, hks, Wd.data(), d, h);
        const uint32_t row_swap = (uint32_t)(2 * n - 1);
        hks.add_galois_element(row_swap);
        const double setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (up.output_ciphertexts() != 1 || down.output_ciphertexts() != 1) throw Exception(ErrorCode::INVALID_STATE, "one output ciphertext per layer expected");

        std::vector<uint64_t> slots(n);
        std::vector<int64_t> coeffs(T * n);
        for (size_t tk = 0; tk < T; ++tk) {
            up.pack_input(&x[tk * d], slots.data());
            be.encode(slots.data(), &coeffs[tk * n]);
        }
        Ciphertext cx(ctx, 2, T), c1(ctx, 2, T), c1s(ctx, 2, T), c1r(ctx, 2, T), c2(ctx, 2, T), cy(ctx, 2, T);
        enc.encrypt_exact(coeffs.data(), t, cx);
        const std::vector<uint32_t> swaps(T, row_swap);
        auto block = [&] {
            up.apply(cx, c1);                              // outputs r < 3072 at slot r of row 0; rows >= 3072 (and all of slot row 1) are 0
            hks.apply_galois_many(c1, swaps, c1s);         // X -> X^(2N-1) swaps the two slot rows
            ev.add(c1, c1s, c1r);                          // both rows now carry W_up x with period 4096: the input packing of `down`
            down.apply(c1r, c2);                           // repeats with period 1024 on both rows: the packing x itself came in
            ev.add(cx, c2, cy);                            // residual
        };
        block();
        ctx.synchronize();
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) block();
        ctx.synchronize();
        const double ms_per_token = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3 / reps / (double)T;

        // checks: the hand-over between the layers is exactly what `down` would have packed itself, and the block's result
        std::vector<uint64_t> dm(T * n), got(n), yv(d), expect(n);
        size_t bad_mid = 0, bad = 0;
        dec.decrypt_exact(c1r, t, dm.data());
        for (size_t tk = 0; tk < T; ++tk) {
            be.decode(dm.data() + tk * n, got.data());
            down.pack_input(&mid[tk * h], expect.data());
            for (size_t i = 0; i < n; ++i) bad_mid += got[i] != expect[i];
        }
        dec.decrypt_exact(cy, t, dm.data());
        for (size_t tk = 0; tk < T; ++tk) {
            be.decode(dm.data() + tk * n, got.data());
            down.unpack_output(got.data(), yv.data());
            for (size_t r = 0; r < d; ++r) bad += yv[r] != want[tk * d + r];
            // and the result is packed like the block's input: it could enter the next block as it is
            std::vector<uint64_t> again(n);
            up.pack_input(&want[tk * d], again.data());
            for (size_t i = 0; i < n; ++i) bad += got[i] != again[i];
        }
        const size_t ks = up.key_switches_per_apply() + down.key_switches_per_apply() + 1;
        if (json)
            std::printf("{\"block\": \"ffn_linear_residual\", \"hidden\": %zu, \"inner\": %zu, \"log2_n\": 13, \"data_limbs\": %zu, \"plain_modulus\": %llu, \"tokens_per_apply\": %zu, "
                        "\"key_switches_per_token\": %zu, \"setup_s\": %.2f, \"ms_per_token\": %.3f, \"handover_correct\": %s, \"correct\": %s}\n",
                        d, h, p.n_limbs(), (unsigned long long)t, T, ks, setup_s, ms_per_token, bad_mid ? "false" : "true", bad ? "false" : "true");
        else
            std::printf("FFN block (linear path + residual) %zu -> %zu -> %zu, %zu token(s) per application, %zu key switches per token; setup %.2f s, %.3f ms per token: "
                        "hand-over %s, result %s\n", d, h, d, T, ks, setup_s, ms_per_token, bad_mid ? "MISMATCH" : "matches the next layer's packing",
                        bad ? "MISMATCH" : "decrypts to x + W_down (W_up x) mod t");
        std::printf((bad || bad_mid) ? "FAILED\n" : "OK\n");
        return (bad || bad_mid) ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
}
