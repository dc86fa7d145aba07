// encrypted_gpt2_block.cpp - BASELINE configs[4] as ONE object: the linear skeleton of a whole GPT-2-small transformer block on
// encrypted, slot-packed hidden states (deeppowers::fhe::PackedTransformerBlock), every dense site of
// /root/reference/src/core/execution/models/gpt_model.cpp:786-859 chained on the device for T tokens:
//     qkv = W_qkv x (:793, 768 -> 2304);  a = v (attention over one position);  h1 = x + W_o a;  h2 = h1 + W_down (W_up h1) (:848)
// over Z_65537, N = 8192, five 60-bit data primes + the special prime of hybrid key switching (examples/quantization_example.cpp's
// 8-bit weights and activations).  Every stage is decrypted and compared with the plaintext computation, and the remaining noise
// budget is reported after each layer.  LayerNorm / GELU / softmax over longer contexts are the non-linear parts (SURVEY.md
// section 7): identity here.
//
// TOKEN SHARDING (what scales): tokens are independent, so with `ranks` > 0 the T tokens are cut into contiguous
// slices, one per rank, each rank runs the block on its slice with NO communication, and one all-gather of the output ciphertexts
// ends the step (deeppowers::fhe::Communicator = RCCL; /root/reference/src/core/distributed/distributed_context.cpp:97-122).  One
// PROCESS per GPU when the node has `ranks` devices (forked before any HIP call; evaluation keys regenerated from the same test seed
// on every rank - a deployment ships them); otherwise one process plays the ranks one after the other on device 0 (same slices, same
// hand-overs, the gather a device copy), which is what runs on the single-GPU test box.
//
//   usage: encrypted_gpt2_block [tokens = 8] [reps = 2] [json | text] [ranks = 0] [layers = 1]
// STAND-INS: the block's LINEAR skeleton only - no activation, no LayerNorm, attention = v (exact at one position only); 1-4 of the reference's 12 blocks.
// SECURITY: N = 8192, 360 bits under key switching against the 218 bits of 128-bit security at that ring (Homomorphic Encryption Standard): BASELINE configs[4]'s
// performance shape, not a deployable parameter set.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <deeppowers/fhe.hpp>

using namespace deeppowers::fhe;

namespace {
const size_t D = 768, H = 3072;
const uint64_t T_MOD = 65537;

struct Weights {
    std::vector<uint64_t> qkv, o, up, down;
};
uint64_t g_seed = 0;
uint64_t rnd(uint64_t m) { g_seed = g_seed * 6364136223846793005ull + 1442695040888963407ull; return (g_seed >> 33) % m; }
void fill8(std::vector<uint64_t>& v) { for (auto& x : v) x = (T_MOD + rnd(255) - 127) % T_MOD; }   // 8-bit quantised, centred

void matvec(const std::vector<uint64_t>& W, size_t rows, size_t cols, const uint64_t* x, uint64_t* y) {
    for (size_t r = 0; r < rows; ++r) {
        unsigned __int128 acc = 0;
        for (size_t c = 0; c < cols; ++c) acc += (unsigned __int128)W[r * cols + c] * x[c];
        y[r] = (uint64_t)(acc % T_MOD);
    }
}
struct Plain {   // plaintext stages of one token
    std::vector<uint64_t> qkv, h1, u, h2;
};
Plain forward_plain(const Weights& w, const uint64_t* x) {
    Plain p;
    p.qkv.resize(3 * D); p.h1.resize(D); p.u.resize(H); p.h2.resize(D);
    matvec(w.qkv, 3 * D, D, x, p.qkv.data());
    std::vector<uint64_t> o(D), dn(D);
    matvec(w.o, D, D, &p.qkv[2 * D], o.data());                  // attention output over one position = v
    for (size_t i = 0; i < D; ++i) p.h1[i] = (x[i] + o[i]) % T_MOD;
    matvec(w.up, H, D, p.h1.data(), p.u.data());
    matvec(w.down, D, H, p.u.data(), dn.data());
    for (size_t i = 0; i < D; ++i) p.h2[i] = (p.h1[i] + dn[i]) % T_MOD;
    return p;
}

struct Result {
    bool ok = false;
    size_t bad_stage[5] = {0, 0, 0, 0, 0};
    double budget[6] = {0, 0, 0, 0, 0, 0};   // fresh, qkv, v hand-over, h1, W_up hand-over, h2
    double ms_per_token = 0, setup_s = 0;
    size_t key_switches = 0;
};

// one rank's work: tokens [lo, hi) of the global batch through `layers` blocks; `comm` (optional) gathers the outputs
int run(int rank, int world, size_t T, int reps, int layers, int device, Communicator* comm_in, const std::string& dir, Result* res, bool emulate_all) {
    FheParams p = FheParams::n8192_l6();
    const uint64_t special = p.moduli.back(), special_psi = p.psi.back();
    p.moduli.pop_back(); p.psi.pop_back();
    const size_t n = p.n();
    Context ctx(p, device);
    Evaluator ev(ctx);
    KeyGenerator kg(ctx, TestSeed{20240917});                    // the same keys on every rank (tests only; a deployment ships the evaluation keys)
    Encryptor enc(ctx, kg.secret_key(), TestSeed{77 + (uint64_t)rank});
    Decryptor dec(ctx, kg.secret_key());
    BatchEncoder be(ctx, T_MOD);
    HybridKeySwitcher hks(ctx, kg.secret_key(), special, special_psi, TestSeed{4242});

    g_seed = 2024;
    std::vector<Weights> w((size_t)layers);
    for (auto& l : w) { l.qkv.resize(3 * D * D); l.o.resize(D * D); l.up.resize(H * D); l.down.resize(D * H); fill8(l.qkv); fill8(l.o); fill8(l.up); fill8(l.down); }
    std::vector<uint64_t> x(T * D);
    fill8(x);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::unique_ptr<PackedTransformerBlock>> blocks;
    for (auto& l : w) blocks.emplace_back(new PackedTransformerBlock(ctx, be, hks, l.qkv.data(), l.o.data(), l.up.data(), l.down.data(), D, H));
    res->setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    res->key_switches = blocks[0]->key_switches_per_token() * (size_t)layers;

    // slices: rank r owns tokens [lo_r, hi_r); in emulation this process walks all of them one after the other
    auto bounds = [&](int r, size_t& lo, size_t& hi) { const size_t base = T / (size_t)world, extra = T % (size_t)world; lo = (size_t)r * base + std::min((size_t)r, extra); hi = lo + base + ((size_t)r < extra ? 1 : 0); };
    size_t max_slice = 0;
    for (int r = 0; r < world; ++r) { size_t lo, hi; bounds(r, lo, hi); max_slice = std::max(max_slice, hi - lo); }
    if (comm_in && T % (size_t)world) { std::fprintf(stderr, "the RCCL all-gather needs equal slices: tokens must be a multiple of ranks\n"); return 6; }
    Ciphertext all_y(ctx, 2, T);                                 // the gathered outputs, token-major
    std::vector<uint64_t> slots(n);
    std::vector<int64_t> coeffs(max_slice * n);
    double elapsed = 0;
    std::vector<std::unique_ptr<Ciphertext>> stage_copy(5);
    for (int r = emulate_all ? 0 : rank; r < (emulate_all ? world : rank + 1); ++r) {
        size_t lo, hi;
        bounds(r, lo, hi);
        const size_t Tr = hi - lo;
        if (Tr == 0) continue;
        for (size_t tk = 0; tk < Tr; ++tk) {
            blocks[0]->pack_input(&x[(lo + tk) * D], slots.data());
            be.encode(slots.data(), &coeffs[tk * n]);
        }
        Ciphertext cx(ctx, 2, Tr), cy(ctx, 2, Tr), cz(ctx, 2, Tr);
        enc.encrypt_exact(coeffs.data(), T_MOD, cx);
        if (r == 0 || !emulate_all) res->budget[0] = dec.noise_budget_bits(cx, T_MOD);
        auto step = [&] {
            const Ciphertext* in = &cx;
            for (int l = 0; l < layers; ++l) {
                Ciphertext& out = (l & 1) ? cz : cy;
                blocks[(size_t)l]->apply(*in, out);
                in = &out;
            }
            return in;
        };
        const Ciphertext* out = step();
        ctx.synchronize();
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) out = step();
        ctx.synchronize();
        elapsed += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // the step's only exchange: all ranks' output ciphertexts gathered token-major
        if (comm_in) comm_in->all_gather(*out, all_y);
        else if (hipMemcpy(all_y.data() + lo * 2 * p.n_limbs() * n, out->data(), Tr * 2 * p.n_limbs() * n * sizeof(uint64_t), hipMemcpyDeviceToDevice) != hipSuccess) return 4;
        ctx.synchronize();
        // stage-by-stage check of the LAST block on this slice (rank 0's slice in a real multi-process run is checked by rank 0)
        if (layers == 1) {
            std::vector<uint64_t> dm(Tr * n), got(n), expect(n), vec(3 * D);
            for (int sidx = 0; sidx < 5; ++sidx) {
                const Ciphertext& sc = blocks[0]->stage(sidx);
                dec.decrypt_exact(sc, T_MOD, dm.data());
                const double b = dec.noise_budget_bits(sc, T_MOD);
                if (r == 0 || !emulate_all || b < res->budget[1 + sidx]) res->budget[1 + sidx] = b;
                for (size_t tk = 0; tk < Tr; ++tk) {
                    const Plain pl = forward_plain(w[0], &x[(lo + tk) * D]);
                    be.decode(dm.data() + tk * n, got.data());
                    if (sidx == 0) {            // q | k | v at slots 0 .. 2303 of row 0, zero elsewhere
                        for (size_t i = 0; i < n; ++i) res->bad_stage[0] += got[i] != (i < 3 * D ? pl.qkv[i] : 0);
                    } else if (sidx == 3) {     // W_up h1 with period 4096 on both rows
                        for (size_t i = 0; i < n; ++i) res->bad_stage[3] += got[i] != ((i % (n / 2)) < H ? pl.u[i % (n / 2)] : 0);
                    } else {                    // v, h1, h2: period 1024 on both rows
                        const uint64_t* want = sidx == 1 ? &pl.qkv[2 * D] : sidx == 2 ? pl.h1.data() : pl.h2.data();
                        for (size_t i = 0; i < n; ++i) { const size_t c = (i % (n / 2)) % 1024; res->bad_stage[sidx] += got[i] != (c < D ? want[c] : 0); }
                    }
                }
            }
        }
    }
    // the gathered result against the plaintext forward of every token (all `layers` blocks)
    size_t bad = 0;
    if (rank == 0) {
        std::vector<uint64_t> dm(T * n), got(n), yv(D);
        dec.decrypt_exact(all_y, T_MOD, dm.data());
        res->budget[5] = dec.noise_budget_bits(all_y, T_MOD);
        for (size_t tk = 0; tk < T; ++tk) {
            std::vector<uint64_t> cur(&x[tk * D], &x[tk * D] + D);
            for (int l = 0; l < layers; ++l) cur = forward_plain(w[(size_t)l], cur.data()).h2;
            be.decode(dm.data() + tk * n, got.data());
            blocks[0]->unpack_output(got.data(), yv.data());
            for (size_t i = 0; i < D; ++i) bad += yv[i] != cur[i];
        }
    }
    for (size_t b : res->bad_stage) bad += b;
    res->ok = bad == 0;
    // per-rank time: in emulation the ranks ran one after the other, so a rank's time is the total over its share
    const double my_tokens = emulate_all ? (double)T : (double)(T / (size_t)world + ((size_t)rank < T % (size_t)world ? 1 : 0));
    res->ms_per_token = elapsed * 1e3 / reps / (my_tokens > 0 ? my_tokens : 1);
    if (!dir.empty()) std::ofstream(dir + "/r" + std::to_string(rank)) << (res->ok ? 1 : 0) << " " << res->ms_per_token << " " << res->budget[5] << "\n";
    return res->ok ? 0 : 1;
}
}  // namespace

int main(int argc, char** argv) {
    const size_t T = argc > 1 ? (size_t)std::atol(argv[1]) : 8;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 2;
    const bool json = argc > 3 && !std::strcmp(argv[3], "json");
    const int ranks = argc > 4 ? std::atoi(argv[4]) : 0;
    const int layers = argc > 5 ? std::atoi(argv[5]) : 1;
    if (T == 0 || reps < 1 || ranks < 0 || ranks > 64 || layers < 1 || layers > 4) { std::fprintf(stderr, "usage: encrypted_gpt2_block [tokens] [reps] [json|text] [ranks] [layers <= 4]\n"); return 1; }
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    try {
        Result res;
        int rc = 0;
        const char* mode = "one process, no sharding";
        double slowest_rank_ms = 0;
        int n_dev = 1;
        if (ranks >= 1) {   // probe the device count in a child: the parent must not touch HIP before forking the ranks
            int fds[2];
            if (pipe(fds)) return 1;
            const pid_t pid = fork();
            if (pid == 0) { int c = 0; if (hipGetDeviceCount(&c) != hipSuccess) c = 0; (void)!write(fds[1], &c, sizeof c); _exit(0); }
            int st = 0; waitpid(pid, &st, 0);
            (void)!read(fds[0], &n_dev, sizeof n_dev);
            close(fds[0]); close(fds[1]);
        }
        if (ranks >= 1 && n_dev >= ranks) {
            mode = "one process per GPU, RCCL all-gather of the output ciphertexts";
            char tmpl[] = "/tmp/dpfhe_block_XXXXXX";
            if (!mkdtemp(tmpl)) { std::perror("mkdtemp"); return 1; }
            const std::string dir = tmpl;
            std::vector<pid_t> kids;
            for (int r = 0; r < ranks; ++r) {
                const pid_t pid = fork();
                if (pid < 0) { std::perror("fork"); return 1; }
                if (pid == 0) {
                    int crc = 2;
                    try {
                        std::vector<uint8_t> id(128);
                        const std::string id_path = dir + "/rccl_id";
                        if (r == 0) {
                            id = Communicator::unique_id();
                            std::ofstream(id_path + ".tmp", std::ios::binary).write(reinterpret_cast<const char*>(id.data()), 128);
                            std::rename((id_path + ".tmp").c_str(), id_path.c_str());
                        } else {
                            for (int tries = 0;; ++tries) {
                                std::ifstream f(id_path, std::ios::binary);
                                if (f && f.read(reinterpret_cast<char*>(id.data()), 128)) break;
                                if (tries > 6000) _exit(3);
                                std::this_thread::sleep_for(std::chrono::milliseconds(10));
                            }
                        }
                        Communicator comm(id, r, ranks, r);
                        Result rr;
                        crc = run(r, ranks, T, reps, layers, r, &comm, dir, &rr, false);
                        if (r == 0) {
                            std::ofstream f(dir + "/rank0");
                            f << rr.setup_s << " " << rr.key_switches;
                            for (double b : rr.budget) f << " " << b;
                            for (size_t b : rr.bad_stage) f << " " << b;
                            f << "\n";
                        }
                    } catch (const std::exception& e) { std::fprintf(stderr, "rank %d: %s\n", r, e.what()); }
                    _exit(crc);
                }
                kids.push_back(pid);
            }
            for (pid_t k : kids) { int st = 0; waitpid(k, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1; }
            res.ok = rc == 0;
            for (int r = 0; r < ranks; ++r) {
                std::ifstream f(dir + "/r" + std::to_string(r));
                int ok = 0; double ms = 0, bud = 0;
                if (!(f >> ok >> ms >> bud) || !ok) res.ok = false;
                slowest_rank_ms = std::max(slowest_rank_ms, ms);
                if (r == 0) res.budget[5] = bud;
                std::remove((dir + "/r" + std::to_string(r)).c_str());
            }
            {
                std::ifstream f(dir + "/rank0");
                f >> res.setup_s >> res.key_switches;
                for (double& b : res.budget) f >> b;
                for (size_t& b : res.bad_stage) f >> b;
            }
            std::remove((dir + "/rank0").c_str()); std::remove((dir + "/rccl_id").c_str()); rmdir(dir.c_str());
            // ranks run concurrently: the step takes what the slowest rank takes for its slice; per token of the WHOLE batch:
            res.ms_per_token = slowest_rank_ms * (double)((T + (size_t)ranks - 1) / (size_t)ranks) / (double)T;
        } else {
            if (ranks >= 1) mode = "one process playing every rank in turn on device 0 (fewer devices than ranks): slices and gather as in the multi-process run";
            rc = run(0, ranks >= 1 ? ranks : 1, T, reps, layers, 0, nullptr, "", &res, true);
        }
        if (json)
            std::printf("{\"block\": \"transformer_linear_skeleton\", \"sites\": \"qkv 768->2304, attention output 768->768 (attention over one position: a = v), ffn 768->3072->768, residuals\", "
                        "\"hidden\": %zu, \"inner\": %zu, \"layers\": %d, \"log2_n\": 13, \"data_limbs\": 5, \"plain_modulus\": %llu, \"tokens\": %zu, \"ranks\": %d, \"mode\": \"%s\", "
                        "\"key_switches_per_token\": %zu, \"setup_s\": %.2f, \"ms_per_token\": %.3f, \"ms_per_token_per_block\": %.3f, "
                        "\"noise_budget_bits\": {\"fresh\": %.0f, \"qkv\": %.0f, \"v_handover\": %.0f, \"h1\": %.0f, \"ffn_up_handover\": %.0f, \"h2\": %.0f, \"gathered_output\": %.0f}, "
                        "\"stage_mismatches\": [%zu, %zu, %zu, %zu, %zu], \"correct\": %s}\n",
                        D, H, layers, (unsigned long long)T_MOD, T, ranks, mode, res.key_switches, res.setup_s, res.ms_per_token, res.ms_per_token / layers, res.budget[0], res.budget[1],
                        res.budget[2], res.budget[3], res.budget[4], res.budget[5], res.budget[5], res.bad_stage[0], res.bad_stage[1], res.bad_stage[2], res.bad_stage[3], res.bad_stage[4],
                        res.ok ? "true" : "false");
        else
            std::printf("transformer block skeleton x %d, %zu token(s), ranks %d (%s): %zu key switches per token, setup %.2f s, %.3f ms per token; noise budget fresh %.0f -> qkv %.0f -> v %.0f -> h1 %.0f "
                        "-> up %.0f -> h2 %.0f bits; %s\n", layers, T, ranks, mode, res.key_switches, res.setup_s, res.ms_per_token, res.budget[0], res.budget[1], res.budget[2], res.budget[3],
                        res.budget[4], res.budget[5], res.ok ? "every stage decrypts to the plaintext result" : "MISMATCH");
        std::printf(res.ok ? "OK\n" : "FAILED\n");
        return res.ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
}
