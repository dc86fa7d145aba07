// encrypted_gpt2_block_act.cpp - one GPT-2-small transformer block on encrypted, slot-packed hidden states WITH the FFN's non-linearity,
// the closest this library comes to BASELINE configs[4] as a forward pass:
//     qkv = W_qkv x (768 -> 2304);  a = v (attention over one position);  h1 = x + W_o a;  h2 = h1 + W_down (W_up h1)^2
// over Z_65537 (/root/reference/src/core/execution/models/gpt_model.cpp:722-784 forward_transformer_layer, :786-840 attention, :842-859
// forward_mlp with x^2 standing in for GELU).  N = 8192.  The attention half and W_up run on five 60-bit limbs; then the modulus is switched
// to two limbs, the activation is an EXACT ciphertext x ciphertext multiply (ExactMultiplier: the fused tensor-product kernel, i.e. the
// metric op, inside the forward) + relinearisation, and W_down and the residual run on two limbs.  Every stage is decrypted and compared
// with the plaintext computation; the noise budget is reported after every stage (six levels: qkv, the v mask, W_o, W_up, the square, W_down).
//   usage: encrypted_gpt2_block_act [tokens = 4] [reps = 2] [json | text]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <deeppowers/fhe.hpp>

using namespace deeppowers::fhe;

namespace {
const size_t D = 768, H = 3072;
const uint64_t TM = 65537;
uint64_t g_seed = 77;
uint64_t rnd(uint64_t m) { g_seed = g_seed * 6364136223846793005ull + 1442695040888963407ull; return (g_seed >> 33) % m; }
void fill8(std::vector<uint64_t>& v) { for (auto& x : v) x = (TM + rnd(255) - 127) % TM; }
void matvec(const std::vector<uint64_t>& W, size_t rows, size_t cols, const uint64_t* x, uint64_t* y) {
    for (size_t r = 0; r < rows; ++r) {
        unsigned __int128 acc = 0;
        for (size_t c = 0; c < cols; ++c) acc += (unsigned __int128)W[r * cols + c] * x[c];
        y[r] = (uint64_t)(acc % TM);
    }
}
}  // namespace

int main(int argc, char** argv) {
    const size_t T = argc > 1 ? (size_t)std::atol(argv[1]) : 4;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 2;
    const bool json = argc > 3 && !std::strcmp(argv[3], "json");
    try {
        FheParams p5 = FheParams::n8192_l6();
        const uint64_t special = p5.moduli.back(), special_psi = p5.psi.back();
        p5.moduli.pop_back(); p5.psi.pop_back();
        const FheParams p4 = p5.drop_last_limb(), p3 = p4.drop_last_limb(), p2 = p3.drop_last_limb();
        const size_t n = p5.n();
        Context ctx5(p5, 0), ctx4(p4, 0), ctx3(p3, 0), ctx2(p2, 0);
        Evaluator ev5(ctx5), ev4(ctx4), ev3(ctx3), ev2(ctx2);
        KeyGenerator kg(ctx5);
        SecretKey sk2(ctx2, kg.secret_key().coefficients());
        Encryptor enc(ctx5, kg.secret_key());
        Decryptor dec5(ctx5, kg.secret_key()), dec2(ctx2, sk2);
        BatchEncoder be5(ctx5, TM), be2(ctx2, TM);
        HybridKeySwitcher hks5(ctx5, kg.secret_key(), special, special_psi), hks2(ctx2, sk2, special, special_psi);
        ExactMultiplier mul(ctx5, ctx2, TM);

        std::vector<uint64_t> Wqkv(3 * D * D), Wo(D * D), Wu(H * D), Wd(D * H), x(T * D);
        fill8(Wqkv); fill8(Wo); fill8(Wu); fill8(Wd); fill8(x);
        // plaintext forward of every token
        std::vector<uint64_t> qkv(T * 3 * D), h1(T * D), u(T * H), act(T * H), h2(T * D), tmp(H);
        for (size_t tk = 0; tk < T; ++tk) {
            matvec(Wqkv, 3 * D, D, &x[tk * D], &qkv[tk * 3 * D]);
            matvec(Wo, D, D, &qkv[tk * 3 * D + 2 * D], tmp.data());                      // attention over one position: its output is v
            for (size_t r = 0; r < D; ++r) h1[tk * D + r] = (x[tk * D + r] + tmp[r]) % TM;
            matvec(Wu, H, D, &h1[tk * D], &u[tk * H]);
            for (size_t r = 0; r < H; ++r) act[tk * H + r] = (uint64_t)((unsigned __int128)u[tk * H + r] * u[tk * H + r] % TM);
            matvec(Wd, D, H, &act[tk * H], tmp.data());
            for (size_t r = 0; r < D; ++r) h2[tk * D + r] = (h1[tk * D + r] + tmp[r]) % TM;
        }

        auto t0 = std::chrono::steady_clock::now();
        PackedLinear lqkv(ctx5, be5, hks5, Wqkv.data(), 3 * D, D), lo(ctx5, be5, hks5, Wo.data(), D, D), lup(ctx5, be5, hks5, Wu.data(), H, D);
        PackedLinear ldown(ctx2, be2, hks2, Wd.data(), D, H);
        PackedSelect take_v(ctx5, be5, hks5, 2 * D, D, lo.input_period());
        const uint32_t row_swap = (uint32_t)(2 * n - 1);
        hks5.add_galois_element(row_swap);
        const double setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

        std::vector<uint64_t> slots(n);
        std::vector<int64_t> coeffs(T * n);
        for (size_t tk = 0; tk < T; ++tk) {
            lqkv.pack_input(&x[tk * D], slots.data());
            be5.encode(slots.data(), &coeffs[tk * n]);
        }
        Ciphertext cx(ctx5, 2, T), cqkv(ctx5, 2, T), ca(ctx5, 2, T), co(ctx5, 2, T), ch1(ctx5, 2, T), cu(ctx5, 2, T), cus(ctx5, 2, T), cur(ctx5, 2, T);
        Ciphertext u4(ctx4, 2, T), u3(ctx3, 2, T), u2(ctx2, 2, T), g4(ctx4, 2, T), g3(ctx3, 2, T), g2(ctx2, 2, T);
        Ciphertext sq3(ctx2, 3, T), sq(ctx2, 2, T), cdn(ctx2, 2, T), ch2(ctx2, 2, T);
        enc.encrypt_exact(coeffs.data(), TM, cx);
        const std::vector<uint32_t> swaps(T, row_swap);
        auto block = [&] {
            lqkv.apply(cx, cqkv);                               // q | k | v at slots 0 .. 3d-1 of row 0                       (gpt_model.cpp:793)
            take_v.apply(cqkv, ca);                             // attention over one position: v, re-packed as a layer input    (one mask level)
            lo.apply(ca, co);
            ev5.add(cx, co, ch1);                               // h1 = x + W_o a
            lup.apply(ch1, cu);                                 // W_up h1                                                        (gpt_model.cpp:848)
            hks5.apply_galois_many(cu, swaps, cus);
            ev5.add(cu, cus, cur);                              // W_down's input packing
            ev5.rescale(cur, u4); ev4.rescale(u4, u3); ev3.rescale(u3, u2);        // modulus switch 5 -> 2 limbs
            mul.multiply(u2, u2, sq3);                          // the activation (exact multiply around the fused ct x ct kernel)
            hks2.relinearize(sq3, sq);
            ldown.apply(sq, cdn);                               // W_down on two limbs
            ev5.rescale(ch1, g4); ev4.rescale(g4, g3); ev3.rescale(g3, g2);        // the residual's operand at that level
            ev2.add(g2, cdn, ch2);                              // h2 = h1 + W_down (W_up h1)^2
        };
        block();
        ctx5.synchronize();
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) block();
        ctx5.synchronize();
        const double ms_per_token = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3 / reps / (double)T;

        std::vector<uint64_t> dm(T * n), got(n), expect(n), yv(D);
        size_t bad_h1 = 0, bad_act = 0, bad = 0;
        dec5.decrypt_exact(ch1, TM, dm.data());
        for (size_t tk = 0; tk < T; ++tk) {
            be5.decode(dm.data() + tk * n, got.data());
            lup.pack_input(&h1[tk * D], expect.data());
            for (size_t i = 0; i < n; ++i) bad_h1 += got[i] != expect[i];
        }
        dec2.decrypt_exact(sq, TM, dm.data());
        for (size_t tk = 0; tk < T; ++tk) {
            be2.decode(dm.data() + tk * n, got.data());
            ldown.pack_input(&act[tk * H], expect.data());
            for (size_t i = 0; i < n; ++i) bad_act += got[i] != expect[i];
        }
        dec2.decrypt_exact(ch2, TM, dm.data());
        for (size_t tk = 0; tk < T; ++tk) {
            be2.decode(dm.data() + tk * n, got.data());
            ldown.unpack_output(got.data(), yv.data());
            for (size_t r = 0; r < D; ++r) bad += yv[r] != h2[tk * D + r];
        }
        const double b[8] = {dec5.noise_budget_bits(cx, TM), dec5.noise_budget_bits(cqkv, TM), dec5.noise_budget_bits(ca, TM), dec5.noise_budget_bits(ch1, TM),
                             dec5.noise_budget_bits(cur, TM), dec2.noise_budget_bits(u2, TM), dec2.noise_budget_bits(sq, TM), dec2.noise_budget_bits(ch2, TM)};
        const size_t ks = lqkv.key_switches_per_apply() + take_v.key_switches_per_apply() + lo.key_switches_per_apply() + lup.key_switches_per_apply() + 1 +
                          ldown.key_switches_per_apply() + 1;
        const bool ok = !(bad || bad_act || bad_h1);
        if (json)
            std::printf("{\"block\": \"transformer_block_square_activation\", \"hidden\": %zu, \"inner\": %zu, \"log2_n\": 13, \"levels\": \"5 limbs (attention half, W_up) -> 2 limbs (square, W_down)\", "
                        "\"plain_modulus\": %llu, \"tokens\": %zu, \"key_switches_per_token\": %zu, \"ct_ct_multiplies_per_token\": 1, \"setup_s\": %.2f, \"ms_per_token\": %.3f, "
                        "\"budget_bits\": [%.0f, %.0f, %.0f, %.0f, %.0f, %.0f, %.0f, %.0f], \"correct\": %s}\n",
                        D, H, (unsigned long long)TM, T, ks, setup_s, ms_per_token, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], ok ? "true" : "false");
        else
            std::printf("transformer block with a square activation, %zu token(s) per application, %zu key switches + one ct x ct multiply per token; setup %.2f s, %.3f ms per token\n"
                        "  noise budget (bits): fresh %.0f -> qkv %.0f -> v hand-over %.0f -> h1 %.0f -> W_up hand-over %.0f -> 2 limbs %.0f -> squared %.0f -> h2 %.0f\n"
                        "  h1 %s, activation %s, h2 %s\n", T, ks, setup_s, ms_per_token, b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7],
                        bad_h1 ? "MISMATCH" : "ok", bad_act ? "MISMATCH" : "ok", bad ? "MISMATCH" : "decrypts to h1 + W_down (W_up h1)^2 mod t");
        std::printf(ok ? "OK\n" : "FAILED\n");
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
}
