// encrypted_gpt2_ffn_act.cpp - one GPT-2-small feed-forward block under encryption WITH a non-linearity between its matrices:
//     y = x + W_down (W_up x)^2   over Z_65537,   W_up: 768 -> 3072, W_down: 3072 -> 768
// (/root/reference/src/core/execution/models/gpt_model.cpp:842-859 forward_mlp: W_up, activation, W_down; gpt_kernels.cu:191-233.  The square
// is the degree-2 polynomial standing in for GELU - the activation an FHE forward can take.)  What this shows beyond encrypted_gpt2_ffn:
//   * the block calls THE METRIC OP: the activation is a ciphertext x ciphertext multiply (ExactMultiplier: exact base extension ->
//     the fused tensor-product kernel on five limbs -> scale by t / q and round -> back), followed by a relinearisation;
//   * the modulus is SWITCHED between the layers: W_up runs on five 60-bit data limbs, its output is rescaled to two limbs (the noise
//     shrinks with the modulus), the square, the relinearisation and W_down run there - three limbs of work per layer less;
//   * the noise budget after every stage (Decryptor::noise_budget_bits).
//   usage: encrypted_gpt2_ffn_act [tokens = 4] [reps = 2] [json | text]
// STAND-INS: x^2 for GELU, no LayerNorm; one FFN block, no attention.
// SECURITY: N = 8192, 360 bits under key switching against the 218 bits of 128-bit security at that ring (Homomorphic Encryption Standard): a performance shape,
// not a deployable parameter set - encrypted_gpt2_block_act ... 14 runs the whole activated block at N = 16384 (360 of 438 bits).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <deeppowers/fhe.hpp>

using namespace deeppowers::fhe;

int main(int argc, char** argv) {
    const size_t T = argc > 1 ? (size_t)std::atol(argv[1]) : 4;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 2;
    const bool json = argc > 3 && !std::strcmp(argv[3], "json");
    const size_t d = 768, h = 3072, level = 2;
    try {
        FheParams p5 = FheParams::n8192_l6();
        const uint64_t special = p5.moduli.back(), special_psi = p5.psi.back();
        p5.moduli.pop_back(); p5.psi.pop_back();
        const FheParams p4 = p5.drop_last_limb(), p3 = p4.drop_last_limb(), p2 = p3.drop_last_limb();
        const size_t n = p5.n();
        Context ctx5(p5, 0), ctx4(p4, 0), ctx3(p3, 0), ctx2(p2, 0);
        Evaluator ev5(ctx5), ev4(ctx4), ev3(ctx3), ev2(ctx2);
        KeyGenerator kg(ctx5);
        SecretKey sk2(ctx2, kg.secret_key().coefficients());           // the same secret, seen at the two-limb level
        Encryptor enc(ctx5, kg.secret_key());
        Decryptor dec5(ctx5, kg.secret_key()), dec2(ctx2, sk2);
        BatchEncoder be5(ctx5, 65537), be2(ctx2, 65537);
        const uint64_t t = be5.plain_modulus();
        HybridKeySwitcher hks5(ctx5, kg.secret_key(), special, special_psi), hks2(ctx2, sk2, special, special_psi);
        ExactMultiplier mul(ctx5, ctx2, t);

        uint64_t s = 4242;
        auto rnd = [&](uint64_t m) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (s >> 33) % m; };
        std::vector<uint64_t> Wu(h * d), Wd(d * h), x(T * d), mid(T * h), act(T * h), want(T * d);
        for (auto& v : Wu) v = (t + rnd(255) - 127) % t;     // 8-bit quantised weights and activations
        for (auto& v : Wd) v = (t + rnd(255) - 127) % t;
        for (auto& v : x) v = (t + rnd(255) - 127) % t;
        for (size_t tk = 0; tk < T; ++tk) {
            for (size_t r = 0; r < h; ++r) {
                unsigned __int128 acc = 0;
                for (size_t c = 0; c < d; ++c) acc += (unsigned __int128)Wu[r * d + c] * x[tk * d + c];
                mid[tk * h + r] = (uint64_t)(acc % t);
                act[tk * h + r] = (uint64_t)((unsigned __int128)mid[tk * h + r] * mid[tk * h + r] % t);
            }
            for (size_t r = 0; r < d; ++r) {
                unsigned __int128 acc = x[tk * d + r];
                for (size_t c = 0; c < h; ++c) acc += (unsigned __int128)Wd[r * h + c] * act[tk * h + c];
                want[tk * d + r] = (uint64_t)(acc % t);
            }
        }

        auto t0 = std::chrono::steady_clock::now();
        PackedLinear up(ctx5, be5, hks5, Wu.data(), h, d), down(ctx2, be2, hks2, Wd.data(), d, h);
        const uint32_t row_swap = (uint32_t)(2 * n - 1);
        hks5.add_galois_element(row_swap);
        const double setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

        std::vector<uint64_t> slots(n);
        std::vector<int64_t> coeffs(T * n);
        for (size_t tk = 0; tk < T; ++tk) {
            up.pack_input(&x[tk * d], slots.data());
            be5.encode(slots.data(), &coeffs[tk * n]);
        }
        Ciphertext cx(ctx5, 2, T), c1(ctx5, 2, T), c1s(ctx5, 2, T), c1r(ctx5, 2, T);
        Ciphertext u4(ctx4, 2, T), u3(ctx3, 2, T), u2(ctx2, 2, T), x4(ctx4, 2, T), x3(ctx3, 2, T), x2(ctx2, 2, T);
        Ciphertext sq3(ctx2, 3, T), sq(ctx2, 2, T), c2(ctx2, 2, T), cy(ctx2, 2, T);
        enc.encrypt_exact(coeffs.data(), t, cx);
        const std::vector<uint32_t> swaps(T, row_swap);
        auto block = [&] {
            up.apply(cx, c1);                              // W_up x on five limbs: outputs r < 3072 at slot r of row 0
            hks5.apply_galois_many(c1, swaps, c1s);
            ev5.add(c1, c1s, c1r);                         // the input packing of `down` (period 4096 on both rows)
            ev5.rescale(c1r, u4); ev4.rescale(u4, u3); ev3.rescale(u3, u2);     // modulus switch 5 -> 2 limbs
            mul.multiply(u2, u2, sq3);                     // the activation: slot-wise square (exact multiply around the fused ct x ct kernel)
            hks2.relinearize(sq3, sq);
            down.apply(sq, c2);                            // W_down on two limbs
            ev5.rescale(cx, x4); ev4.rescale(x4, x3); ev3.rescale(x3, x2);      // the residual's operand at the same level
            ev2.add(x2, c2, cy);
        };
        block();
        ctx5.synchronize();
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) block();
        ctx5.synchronize();
        const double ms_per_token = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3 / reps / (double)T;

        // checks: the squared hand-over is exactly what `down` would have packed from (W_up x)^2, and the block's result
        std::vector<uint64_t> dm(T * n), got(n), yv(d), expect(n);
        size_t bad_act = 0, bad = 0;
        dec2.decrypt_exact(sq, t, dm.data());
        for (size_t tk = 0; tk < T; ++tk) {
            be2.decode(dm.data() + tk * n, got.data());
            down.pack_input(&act[tk * h], expect.data());
            for (size_t i = 0; i < n; ++i) bad_act += got[i] != expect[i];
        }
        dec2.decrypt_exact(cy, t, dm.data());
        for (size_t tk = 0; tk < T; ++tk) {
            be2.decode(dm.data() + tk * n, got.data());
            down.unpack_output(got.data(), yv.data());
            for (size_t r = 0; r < d; ++r) bad += yv[r] != want[tk * d + r];
        }
        const double b_in = dec5.noise_budget_bits(cx, t), b_up = dec5.noise_budget_bits(c1r, t), b_sw = dec2.noise_budget_bits(u2, t),
                     b_sq = dec2.noise_budget_bits(sq, t), b_out = dec2.noise_budget_bits(cy, t);
        const size_t ks = up.key_switches_per_apply() + down.key_switches_per_apply() + 2;
        if (json)
            std::printf("{\"block\": \"ffn_square_activation_residual\", \"hidden\": %zu, \"inner\": %zu, \"log2_n\": 13, \"levels\": \"5 limbs -> 2 limbs\", \"plain_modulus\": %llu, "
                        "\"tokens\": %zu, \"key_switches_per_token\": %zu, \"ct_ct_multiplies_per_token\": 1, \"setup_s\": %.2f, \"ms_per_token\": %.3f, "
                        "\"budget_bits\": [%.0f, %.0f, %.0f, %.0f, %.0f], \"activation_correct\": %s, \"correct\": %s}\n",
                        d, h, (unsigned long long)t, T, ks, setup_s, ms_per_token, b_in, b_up, b_sw, b_sq, b_out, bad_act ? "false" : "true", bad ? "false" : "true");
        else
            std::printf("FFN block with a square activation %zu -> %zu -> (.)^2 -> %zu + residual, %zu token(s) per application, %zu key switches and one ct x ct multiply per token; "
                        "setup %.2f s, %.3f ms per token\n  noise budget (bits): fresh %.0f -> W_up (5 limbs) %.0f -> switched to 2 limbs %.0f -> squared + relinearised %.0f -> W_down + residual %.0f\n"
                        "  activation %s, result %s\n", d, h, d, T, ks, setup_s, ms_per_token, b_in, b_up, b_sw, b_sq, b_out,
                        bad_act ? "MISMATCH" : "decrypts to (W_up x)^2 in the next layer's packing", bad ? "MISMATCH" : "decrypts to x + W_down (W_up x)^2 mod t");
        std::printf((bad || bad_act) ? "FAILED\n" : "OK\n");
        return (bad || bad_act) ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
}
