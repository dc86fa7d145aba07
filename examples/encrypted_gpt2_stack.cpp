// encrypted_gpt2_stack.cpp - SEVERAL GPT-2-small transformer blocks in a row on encrypted, slot-packed hidden states, each with the FFN's
// non-linearity (x^2 standing in for GELU), on a modulus CHAIN that falls with the noise budget:
//     per block:  qkv = W_qkv x;  a = v (attention over one position);  h1 = x + W_o a;  h2 = h1 + W_down (W_up h1)^2;   x <- h2
// over Z_65537 (/root/reference/src/core/execution/models/gpt_model.cpp:626-672 the layer loop, :722-784 forward_transformer_layer,
// :842-859 forward_mlp).  N = 8192.  A block has six multiplicative levels (qkv, the v mask, W_o, W_up, the square, W_down: ~155 bits of
// noise budget), so a stack needs more modulus than BASELINE configs[4]'s five data limbs: `data_limbs` of FheParams::n8192 (default 7 =
// 420 bits: two blocks; 10 = 600 bits: three).  Before every level the ciphertexts are switched down to the FEWEST limbs that still hold the budget they have left
// (a layer's cost goes with digits x limbs), so the first block runs on 7..5 limbs and the last on 4..2.  The activation is an EXACT
// ciphertext x ciphertext multiply (ExactMultiplier around the fused tensor-product kernel, the metric op) at whatever level the chain is on
// at that point (five limbs in the first of two blocks: an eleven-limb workspace).  Every block's output is decrypted and compared with the
// plaintext forward; the budget is read after every level.
//   usage: encrypted_gpt2_stack [tokens = 4] [reps = 1] [json | text] [data_limbs = 7] [blocks = 0: as many as the budget model allows]
// STAND-INS: x^2 for GELU, no LayerNorm, attention = v (exact at one position only), at most 3 of the reference's 12 blocks, no LM head.
// SECURITY: NONE TO SPEAK OF - N = 8192 with 7 ... 20 data primes + special prime is 480 ... 1260 bits under key switching against the 218 bits of 128-bit security
// at that ring (Homomorphic Encryption Standard).  This program demonstrates the budget arithmetic of a deep chain (limb count per level planned from a noise model,
// exact multiplies at several levels), not a deployable forward pass; a chain this long needs N >= 32768.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <deeppowers/fhe.hpp>

using namespace deeppowers::fhe;

namespace {
const size_t D = 768, H = 3072;
const uint64_t TM = 65537;
uint64_t g_seed = 99;
uint64_t rnd(uint64_t m) { g_seed = g_seed * 6364136223846793005ull + 1442695040888963407ull; return (g_seed >> 33) % m; }
void fill8(std::vector<uint64_t>& v) { for (auto& x : v) x = (TM + rnd(255) - 127) % TM; }
void matvec(const std::vector<uint64_t>& W, size_t rows, size_t cols, const uint64_t* x, uint64_t* y) {
    for (size_t r = 0; r < rows; ++r) {
        unsigned __int128 acc = 0;
        for (size_t c = 0; c < cols; ++c) acc += (unsigned __int128)W[r * cols + c] * x[c];
        y[r] = (uint64_t)(acc % TM);
    }
}
// the budget model the level schedule is planned with (a server has no secret key to measure with): bits a level consumes, and what is
// left right after a switch to l limbs at the most (60 l minus log2 t, the rounding noise of the switch and its key-switch floor)
const double kCost[6] = {27, 25, 26, 27, 30, 28};   // qkv, v mask, W_o, W_up, square, W_down  (measured 26 / 23 / 25 / 25 / 29 / 27)
double cap_bits(int limbs) { return 60.0 * limbs - 27.0; }
const double kBlockCost = 27 + 25 + 26 + 27 + 30 + 28, kMargin = 12, kWaste = 8;
enum { QKV = 0, MASK, WO, WUP, SQUARE, WDOWN };
}  // namespace

int main(int argc, char** argv) {
    const size_t T = argc > 1 ? (size_t)std::atol(argv[1]) : 4;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 1;
    const bool json = argc > 3 && !std::strcmp(argv[3], "json");
    const int LD = argc > 4 ? std::atoi(argv[4]) : 7;
    int want_blocks = argc > 5 ? std::atoi(argv[5]) : 0;
    if (LD < 3 || LD > 10 || T == 0) { std::printf("data_limbs must be in [3, 10], tokens > 0\n"); return 2; }
    // round 6: two tokens per ciphertext (argv[6] = 2): the two slot rows of a ciphertext carry two tokens (PackedLinear's two-token packing)
    const size_t tpc = argc > 6 ? (size_t)std::atol(argv[6]) : 1;
    if ((tpc != 1 && tpc != 2) || T % tpc) { std::printf("tokens_per_ciphertext must be 1 or 2 and divide the token count\n"); return 2; }
    const size_t C = T / tpc;   // ciphertexts per application
    try {
        const FheParams chain = FheParams::n8192(20);
        // the special prime of the hybrid key switches: the first prime of the chain that is not a data limb
        const uint64_t special = chain.moduli[LD], special_psi = chain.psi[LD];
        const size_t n = chain.n();
        struct Level {
            std::unique_ptr<Context> ctx;
            std::unique_ptr<Evaluator> ev;
            std::unique_ptr<SecretKey> sk;
            std::unique_ptr<Decryptor> dec;
            std::unique_ptr<BatchEncoder> be;
            std::unique_ptr<HybridKeySwitcher> hks;
            std::unique_ptr<Context> work;              // 2 l + 1 limbs: the workspace of an exact multiply at this level
            std::unique_ptr<ExactMultiplier> mul;
            std::unique_ptr<PackedLinear> lin[6];
            std::unique_ptr<PackedSelect> take_v;
        };
        std::vector<Level> lv(LD + 1);
        std::unique_ptr<KeyGenerator> kg;
        for (int l = LD; l >= 2; --l) {
            lv[l].ctx.reset(new Context(FheParams::n8192((size_t)l), 0));
            lv[l].ev.reset(new Evaluator(*lv[l].ctx));
            if (l == LD) kg.reset(new KeyGenerator(*lv[l].ctx));
            else lv[l].sk.reset(new SecretKey(*lv[l].ctx, kg->secret_key().coefficients()));   // the same secret, seen at that level
            const SecretKey& s = l == LD ? kg->secret_key() : *lv[l].sk;
            lv[l].dec.reset(new Decryptor(*lv[l].ctx, s));
            lv[l].be.reset(new BatchEncoder(*lv[l].ctx, TM));
        }
        auto secret = [&](int l) -> const SecretKey& { return l == LD ? kg->secret_key() : *lv[l].sk; };
        auto hks = [&](int l) -> HybridKeySwitcher& {
            if (!lv[l].hks) lv[l].hks.reset(new HybridKeySwitcher(*lv[l].ctx, secret(l), special, special_psi));
            return *lv[l].hks;
        };
        Encryptor enc(*lv[LD].ctx, kg->secret_key());

        std::vector<uint64_t> Wm[6];
        Wm[QKV].resize(3 * D * D); Wm[WO].resize(D * D); Wm[WUP].resize(H * D); Wm[WDOWN].resize(D * H);
        fill8(Wm[QKV]); fill8(Wm[WO]); fill8(Wm[WUP]); fill8(Wm[WDOWN]);
        const size_t dims[6][2] = {{3 * D, D}, {0, 0}, {D, D}, {H, D}, {0, 0}, {D, H}};
        double setup_s = 0;
        auto layer = [&](int kind, int l) -> PackedLinear& {
            if (!lv[l].lin[kind]) {
                const auto t0 = std::chrono::steady_clock::now();
                lv[l].lin[kind].reset(new PackedLinear(*lv[l].ctx, *lv[l].be, hks(l), Wm[kind].data(), dims[kind][0], dims[kind][1], tpc));
                setup_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            return *lv[l].lin[kind];
        };
        const uint32_t row_swap = (uint32_t)(2 * n - 1);

        // every buffer of the forward is created on first use (the first, untimed pass) and reused afterwards
        std::map<std::string, std::unique_ptr<Ciphertext>> pool;
        auto buf = [&](const std::string& tag, int l, size_t comps = 2) -> Ciphertext& {
            auto& p = pool[tag + "@" + std::to_string(l) + "x" + std::to_string(comps)];
            if (!p) p.reset(new Ciphertext(*lv[l].ctx, comps, C));
            return *p;
        };
        // walks `in` (level `from`) down to level `to` through this tag's buffers
        auto down = [&](const Ciphertext& in, int from, int to, const std::string& tag) -> const Ciphertext& {
            const Ciphertext* cur = &in;
            for (int l = from; l > to; --l) { Ciphertext& nx = buf(tag, l - 1); lv[l].ev->rescale(*cur, nx); cur = &nx; }
            return *cur;
        };
        // the fewest limbs that keep (almost) all of the estimated budget
        auto choose = [&](double est, int cur) { int l = cur; while (l > 2 && cap_bits(l - 1) >= est - kWaste) --l; return l; };

        // plaintext forward of every token through every block, block by block
        std::vector<uint64_t> x(T * D);
        fill8(x);
        struct Reading { int block; const char* what; int limbs; double est, bits; };
        std::vector<Reading> trace;
        std::vector<int> levels_used;
        std::vector<std::vector<uint64_t>> h2_plain;       // per block
        std::vector<const Ciphertext*> h2_ct;              // per block: its output ciphertext and level
        std::vector<int> h2_level;
        int blocks_done = 0;
        const std::vector<uint32_t> swaps(C, row_swap);

        // ---- one pass over the stack: `measure` (first pass only) reads the budgets and records the schedule --------------------------------
        Ciphertext cx(*lv[LD].ctx, 2, C);
        double fresh_bits = 0;
        auto forward = [&](bool measure) {
            const Ciphertext* cur = &cx;
            int l = LD;
            double est = cap_bits(LD) - 3;                    // a fresh ciphertext, by the model (measured: `fresh_bits`)
            int b = 0;
            auto note = [&](const char* what, const Ciphertext& c, int level) {
                if (measure) trace.push_back(Reading{b, what, level, est, lv[level].dec->noise_budget_bits(c, TM)});
            };
            for (;; ++b) {
                if (want_blocks > 0 ? b >= want_blocks : est < kBlockCost + kMargin) break;
                const std::string B = "b" + std::to_string(b);
                // qkv
                int l1 = choose(est, l);
                const Ciphertext& xin = down(*cur, l, l1, B + "x");
                Ciphertext& cqkv = buf(B + "qkv", l1);
                layer(QKV, l1).apply(xin, cqkv);                                               // gpt_model.cpp:793
                est = std::min(est, cap_bits(l1)) - kCost[QKV];
                note("qkv", cqkv, l1);
                // the v hand-over (one mask level) and W_o + residual
                int l2 = choose(est, l1);
                const Ciphertext& qkv_l = down(cqkv, l1, l2, B + "qkvd");
                // (every 768-wide layer input shares one packing, so the qkv layer's period is W_o's)
                if (!lv[l2].take_v) lv[l2].take_v.reset(new PackedSelect(*lv[l2].ctx, *lv[l2].be, hks(l2), 2 * D, D, layer(QKV, l1).input_period(), tpc));
                Ciphertext& ca = buf(B + "a", l2);
                lv[l2].take_v->apply(qkv_l, ca);                                              // attention over one position: its output is v
                est = std::min(est, cap_bits(l2)) - kCost[MASK];
                note("v hand-over", ca, l2);
                int l3 = choose(est, l2);
                const Ciphertext& a_l = down(ca, l2, l3, B + "ad");
                Ciphertext& co = buf(B + "o", l3);
                layer(WO, l3).apply(a_l, co);
                const Ciphertext& x_l3 = down(xin, l1, l3, B + "xd");
                Ciphertext& ch1 = buf(B + "h1", l3);
                lv[l3].ev->add(x_l3, co, ch1);                                                // h1 = x + W_o a
                est = std::min(est, cap_bits(l3)) - kCost[WO];
                note("h1", ch1, l3);
                // W_up and W_down's input packing
                int l4 = choose(est, l3);
                const Ciphertext& h1_l4 = down(ch1, l3, l4, B + "h1d");
                Ciphertext &cu = buf(B + "u", l4), &cus = buf(B + "us", l4), &cur2 = buf(B + "ur", l4);
                layer(WUP, l4).apply(h1_l4, cu);                                              // gpt_model.cpp:848
                const Ciphertext* w_in = &cur2;
                if (tpc == 1) {
                    hks(l4).add_galois_element(row_swap);
                    hks(l4).apply_galois_many(cu, swaps, cus);
                    lv[l4].ev->add(cu, cus, cur2);
                } else {
                    w_in = &cu;   // two tokens per ciphertext: every row already holds ITS token's W_up outputs (a row of 4096 slots is W_down's whole input window)
                }
                est = std::min(est, cap_bits(l4)) - kCost[WUP];
                note("W_up", *w_in, l4);
                // the activation: exact multiply at the level the chain is on + relinearisation
                int l5 = std::min(choose(est, l4), 9);                                        // (a multiply's workspace is 2 l + 1 <= 19 limbs)
                const Ciphertext& u_l5 = down(*w_in, l4, l5, B + "urd");
                if (!lv[l5].mul) {
                    lv[l5].work.reset(new Context(FheParams::n8192((size_t)(2 * l5 + 1)), 0));
                    lv[l5].mul.reset(new ExactMultiplier(*lv[l5].work, *lv[l5].ctx, TM));
                }
                Ciphertext &sq3 = buf(B + "sq3", l5, 3), &sq = buf(B + "sq", l5);
                lv[l5].mul->multiply(u_l5, u_l5, sq3);
                hks(l5).relinearize(sq3, sq);
                est = std::min(est, cap_bits(l5)) - kCost[SQUARE];
                note("square", sq, l5);
                // W_down + residual
                int l6 = choose(est, l5);
                const Ciphertext& sq_l6 = down(sq, l5, l6, B + "sqd");
                Ciphertext &cdn = buf(B + "dn", l6), &ch2 = buf(B + "h2", l6);
                layer(WDOWN, l6).apply(sq_l6, cdn);
                const Ciphertext& h1_l6 = down(ch1, l3, l6, B + "h1r");
                lv[l6].ev->add(h1_l6, cdn, ch2);                                              // h2 = h1 + W_down (W_up h1)^2
                est = std::min(est, cap_bits(l6)) - kCost[WDOWN];
                note("h2", ch2, l6);
                if (measure) {
                    for (int v : {l1, l2, l3, l4, l5, l6}) levels_used.push_back(v);
                    h2_ct.push_back(&ch2); h2_level.push_back(l6);
                }
                cur = &ch2; l = l6;
            }
            return b;
        };

        // encrypt
        std::vector<uint64_t> slots(n);
        std::vector<int64_t> coeffs(C * n);
        {
            PackedLinear& lq = layer(QKV, LD);
            for (size_t c = 0; c < C; ++c) {
                if (tpc == 1) lq.pack_input(&x[c * D], slots.data());
                else lq.pack_input_rows(&x[(2 * c) * D], &x[(2 * c + 1) * D], slots.data());
                lv[LD].be->encode(slots.data(), &coeffs[c * n]);
            }
        }
        enc.encrypt_exact(coeffs.data(), TM, cx);
        fresh_bits = lv[LD].dec->noise_budget_bits(cx, TM);

        const auto s0 = std::chrono::steady_clock::now();
        blocks_done = forward(true);                              // builds layers, keys, buffers; reads the budgets
        lv[LD].ctx->synchronize();
        const double first_pass_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count();
        if (want_blocks == 0) want_blocks = blocks_done;          // the timed passes repeat exactly this schedule
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) forward(false);
        lv[LD].ctx->synchronize();
        const double ms_per_token = reps > 0 ? std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3 / reps / (double)T : 0;

        // plaintext forward, block by block, and the comparison of every block's decrypted output
        std::vector<uint64_t> xp = x, qkv(3 * D), h1(D), u(H), tmp(H);
        for (int b = 0; b < blocks_done; ++b) {
            std::vector<uint64_t> out(T * D);
            for (size_t tk = 0; tk < T; ++tk) {
                matvec(Wm[QKV], 3 * D, D, &xp[tk * D], qkv.data());
                matvec(Wm[WO], D, D, &qkv[2 * D], tmp.data());
                for (size_t r = 0; r < D; ++r) h1[r] = (xp[tk * D + r] + tmp[r]) % TM;
                matvec(Wm[WUP], H, D, h1.data(), u.data());
                for (size_t r = 0; r < H; ++r) u[r] = (uint64_t)((unsigned __int128)u[r] * u[r] % TM);
                matvec(Wm[WDOWN], D, H, u.data(), tmp.data());
                for (size_t r = 0; r < D; ++r) out[tk * D + r] = (h1[r] + tmp[r]) % TM;
            }
            h2_plain.push_back(out);
            xp = out;
        }
        std::vector<uint64_t> dm(C * n), got(n), yv(D), yw(D);
        std::vector<size_t> bad(blocks_done, 0);
        int correct_blocks = 0;
        for (int b = 0; b < blocks_done; ++b) {
            const int l = h2_level[b];
            lv[l].dec->decrypt_exact(*h2_ct[b], TM, dm.data());
            PackedLinear& ld = layer(WDOWN, l);
            for (size_t c = 0; c < C; ++c) {
                lv[l].be->decode(dm.data() + c * n, got.data());
                if (tpc == 1) {
                    ld.unpack_output(got.data(), yv.data());
                    for (size_t r = 0; r < D; ++r) bad[b] += yv[r] != h2_plain[b][c * D + r];
                } else {
                    ld.unpack_output_rows(got.data(), yv.data(), yw.data());
                    for (size_t r = 0; r < D; ++r) bad[b] += (yv[r] != h2_plain[b][(2 * c) * D + r]) + (yw[r] != h2_plain[b][(2 * c + 1) * D + r]);
                }
            }
            if (!bad[b] && correct_blocks == b) ++correct_blocks;
        }
        const bool ok = blocks_done > 0 && correct_blocks == blocks_done;
        size_t ks = 0;
        for (int b = 0; b < blocks_done; ++b) {
            const int* L6 = &levels_used[b * 6];
            ks += layer(QKV, L6[0]).key_switches_per_apply() + lv[L6[1]].take_v->key_switches_per_apply() + layer(WO, L6[2]).key_switches_per_apply() +
                  layer(WUP, L6[3]).key_switches_per_apply() + 1 + 1 + layer(WDOWN, L6[5]).key_switches_per_apply();
        }
        std::string lev, bits, ests;
        for (size_t i = 0; i < levels_used.size(); ++i) lev += (i ? (i % 6 ? " " : " | ") : "") + std::to_string(levels_used[i]);
        for (size_t i = 0; i < trace.size(); ++i) {
            char t[32];
            std::snprintf(t, sizeof t, "%s%.0f", i ? ", " : "", trace[i].bits); bits += t;
            std::snprintf(t, sizeof t, "%s%.0f", i ? ", " : "", trace[i].est); ests += t;
        }
        if (json)
            std::printf("{\"stack\": \"transformer_blocks_square_activation\", \"hidden\": %zu, \"inner\": %zu, \"log2_n\": 13, \"data_limbs\": %d, \"blocks\": %d, "
                        "\"correct_blocks\": %d, \"limbs_per_level\": \"%s\", \"plain_modulus\": %llu, \"tokens\": %zu, \"tokens_per_ciphertext\": %zu, \"key_switches_per_token\": %zu, "
                        "\"ct_ct_multiplies_per_token\": %d, \"setup_s\": %.2f, \"first_pass_s\": %.2f, \"ms_per_token\": %.3f, \"ms_per_token_per_block\": %.3f, "
                        "\"fresh_budget_bits\": %.0f, \"budget_bits\": [%s], \"planned_bits\": [%s], \"correct\": %s}\n",
                        D, H, LD, blocks_done, correct_blocks, lev.c_str(), (unsigned long long)TM, T, tpc, ks, blocks_done, setup_s, first_pass_s, ms_per_token,
                        blocks_done ? ms_per_token / blocks_done : 0.0, fresh_bits, bits.c_str(), ests.c_str(), ok ? "true" : "false");
        else {
            std::printf("%d transformer block(s) with a square activation on %d data limbs, %zu token(s) per application: %zu key switches + %d ct x ct multiplies per token; "
                        "layer setup %.2f s, first pass %.2f s, %.3f ms per token (%.3f per block)\n  limbs per level (qkv, v, W_o, W_up, square, W_down | next block): %s\n"
                        "  noise budget (bits), fresh %.0f, then after every level measured (planned):",
                        blocks_done, LD, T, ks, blocks_done, setup_s, first_pass_s, ms_per_token, blocks_done ? ms_per_token / blocks_done : 0.0, lev.c_str(), fresh_bits);
            for (const Reading& r : trace) std::printf("%s %s %.0f (%.0f)", std::strcmp(r.what, "qkv") ? "," : "\n    block", r.what, r.bits, r.est);
            std::printf("\n");
            for (int b = 0; b < blocks_done; ++b) std::printf("  block %d output %s\n", b, bad[b] ? "MISMATCH" : "decrypts to the plaintext forward mod t");
        }
        std::printf(ok ? "OK\n" : "FAILED\n");
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
}
