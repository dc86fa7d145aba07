// encrypted_gpt2_linear.cpp - one token through GPT-2-small's dense layers under encryption, slot-packed (N3 of SURVEY.md 8f):
// N = 8192, six pinned 60-bit primes (five carry the data, the sixth is the special prime of hybrid key switching), y = W x over
// Z_65537 by the diagonal method with baby-step / giant-step rotations (deeppowers::fhe::PackedLinear).
// The layer shapes are the reference's matmul sites (/root/reference/src/core/execution/models/gpt_model.cpp:793 QKV 768 -> 2304,
// :848 FFN 768 -> 3072 -> 768, :883 logits 768 -> 50257; hidden_size 768, vocab 50257 at execution/model.hpp:47-50).
//   usage: encrypted_gpt2_linear [layer = all | square | qkv | ffn_up | ffn_down | lm_head | <out>x<in>] [reps = 2] [json | text] [tokens = 1] [log2_n = 13 | 14] [tokens_per_ciphertext = 1 | 2]
// tokens_per_ciphertext = 2 (round 6): the two slot rows of a ciphertext carry two tokens (PackedLinear's two-token packing) - the same kernels on half the ciphertexts.
// WHAT THIS IS: single dense layers (matrix x encrypted vector over Z_65537), nothing else of the model.  SECURITY: at the default N = 8192 the 360-bit modulus
// under key switching is beyond the 218 bits of 128-bit security at that ring (Homomorphic Encryption Standard): a performance shape (BASELINE configs[4]).
// log2_n = 14: the same layer at N = 16384 on six primes that are 1 mod 2^15 (round 5: the packed pipeline's rotations above N = 8192 are composed
// from the batched transforms; a 360-bit modulus under key switching at N = 16384 is inside the 128-bit-security budget of 438 bits).
// Prints one line per layer; with a third argument "json" the lines are JSON objects (bench.py other_configs.packed_linear).
// tokens > 1: that many encrypted hidden states go through the layer in ONE application (keys and diagonals read once).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <deeppowers/fhe.hpp>

using namespace deeppowers::fhe;

struct Shape { const char* name; size_t out, in; };

int main(int argc, char** argv) {
    const std::string which = argc > 1 ? argv[1] : "all";
    const int reps = argc > 2 ? std::atoi(argv[2]) : 2;
    const bool json = argc > 3 && !std::strcmp(argv[3], "json");
    const size_t T = argc > 4 ? (size_t)std::atol(argv[4]) : 1;
    const int log2n = argc > 5 ? std::atoi(argv[5]) : 13;
    if (log2n != 13 && log2n != 14) { std::fprintf(stderr, "log2_n must be 13 or 14\n"); return 1; }
    const size_t tpc = argc > 6 ? (size_t)std::atol(argv[6]) : 1;
    if ((tpc != 1 && tpc != 2) || T % tpc) { std::fprintf(stderr, "tokens_per_ciphertext must be 1 or 2 and divide the token count\n"); return 1; }
    const size_t C = T / tpc;   // ciphertexts per application
    std::vector<Shape> shapes;
    const Shape known[] = {{"square", 768, 768}, {"qkv", 2304, 768}, {"ffn_up", 3072, 768}, {"ffn_down", 768, 3072}, {"lm_head", 50257, 768}};
    for (const Shape& k : known)
        if (which == k.name || (which == "all" && std::strcmp(k.name, "lm_head") != 0)) shapes.push_back(k);
    if (shapes.empty()) {
        size_t o = 0, i = 0;
        if (std::sscanf(which.c_str(), "%zux%zu", &o, &i) == 2 && o && i) shapes.push_back(Shape{"custom", o, i});
        else { std::fprintf(stderr, "unknown layer '%s'\n", which.c_str()); return 1; }
    }
    try {
        FheParams p = log2n == 14 ? FheParams::n16384(6) : FheParams::n8192_l6();
        const uint64_t special = p.moduli.back(), special_psi = p.psi.back();
        p.moduli.pop_back(); p.psi.pop_back();
        const size_t n = p.n();
        Context ctx(p, 0);
        KeyGenerator kg(ctx);   // OS CSPRNG (TestSeed{..} would make the run reproducible)
        Encryptor enc(ctx, kg.secret_key());
        Decryptor dec(ctx, kg.secret_key());
        BatchEncoder be(ctx, 65537);
        const uint64_t t = be.plain_modulus();
        HybridKeySwitcher hks(ctx, kg.secret_key(), special, special_psi);
        int rc = 0;
        for (const Shape& sh : shapes) {
            // 8-bit quantised weights and activations (the reference's INT8 path: src/core/quantization), values in [-127, 127] mod t
            uint64_t s = 5 + sh.out;
            auto rnd = [&](uint64_t m) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (s >> 33) % m; };
            std::vector<uint64_t> W(sh.out * sh.in), x(T * sh.in), want(T * sh.out);
            for (auto& v : W) v = (t + rnd(255) - 127) % t;
            for (auto& v : x) v = (t + rnd(255) - 127) % t;
            for (size_t tk = 0; tk < T; ++tk)
                for (size_t r = 0; r < sh.out; ++r) {
                    unsigned __int128 acc = 0;
                    for (size_t c = 0; c < sh.in; ++c) acc += (unsigned __int128)W[r * sh.in + c] * x[tk * sh.in + c];
                    want[tk * sh.out + r] = (uint64_t)(acc % t);
                }
            auto t0 = std::chrono::steady_clock::now();
            PackedLinear layer(ctx, be, hks, W.data(), sh.out, sh.in, tpc);     // encodes + transforms the diagonals, generates the rotation keys
            const double setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const size_t outs = layer.output_ciphertexts();
            std::vector<uint64_t> slots(n), dm(outs * C * n), got(outs * n), y(sh.out), y1(sh.out);
            std::vector<int64_t> coeffs(C * n);
            for (size_t c = 0; c < C; ++c) {   // ciphertext c: token c, or tokens 2 c | 2 c + 1 in its two slot rows
                if (tpc == 1) layer.pack_input(&x[c * sh.in], slots.data());
                else layer.pack_input_rows(&x[(2 * c) * sh.in], &x[(2 * c + 1) * sh.in], slots.data());
                be.encode(slots.data(), &coeffs[c * n]);
            }
            Ciphertext cx(ctx, 2, C), cy(ctx, 2, outs * C);
            enc.encrypt_exact(coeffs.data(), t, cx);
            layer.apply(cx, cy);                                // warm-up (code objects, allocator)
            ctx.synchronize();
            t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) layer.apply(cx, cy);
            ctx.synchronize();
            const double apply_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3 / reps / (double)T;
            dec.decrypt_exact(cy, t, dm.data());
            size_t bad = 0;
            for (size_t c = 0; c < C; ++c) {
                for (size_t o = 0; o < outs; ++o) be.decode(dm.data() + (o * C + c) * n, got.data() + o * n);
                if (tpc == 1) {
                    layer.unpack_output(got.data(), y.data());
                    for (size_t r = 0; r < sh.out; ++r) bad += y[r] != want[c * sh.out + r];
                } else {
                    layer.unpack_output_rows(got.data(), y.data(), y1.data());
                    for (size_t r = 0; r < sh.out; ++r) bad += (y[r] != want[(2 * c) * sh.out + r]) + (y1[r] != want[(2 * c + 1) * sh.out + r]);
                }
            }
            if (json)
                std::printf("{\"layer\": \"%s\", \"out_dim\": %zu, \"in_dim\": %zu, \"log2_n\": %d, \"data_limbs\": %zu, \"plain_modulus\": %llu, \"baby_steps\": %zu, "
                            "\"giant_steps\": %zu, \"output_ciphertexts\": %zu, \"key_switches\": %zu, \"tokens_per_apply\": %zu, \"tokens_per_ciphertext\": %zu, \"setup_s\": %.2f, \"ms_per_token\": %.3f, \"correct\": %s}\n",
                            sh.name, sh.out, sh.in, log2n, p.n_limbs(), (unsigned long long)t, layer.baby_steps(), layer.giant_steps(), outs, layer.key_switches_per_apply(), T, tpc,
                            setup_s, apply_ms, bad ? "false" : "true");
            else
                std::printf("%-8s %5zu <- %4zu: period %zu, %zu baby x %zu giant steps, %zu output ciphertext(s), %zu key switches, %zu token(s) per apply; setup %.2f s, apply %.3f ms per token: %s\n",
                            sh.name, sh.out, sh.in, layer.input_period(), layer.baby_steps(), layer.giant_steps(), outs, layer.key_switches_per_apply(), T, setup_s, apply_ms,
                            bad ? "MISMATCH" : "decrypts to W x mod t");
            if (bad) rc = 1;
        }
        std::printf(rc ? "FAILED\n" : "OK\n");
        return rc;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
}
