// sharded_ffn.cpp - the encrypted FFN's linear path, TENSOR-PARALLEL over the GPUs of one node (the shape of BASELINE configs[4]: "N=8192,
// 6 RNS limbs, encrypted GPT-2-small single-token forward, 8 GPUs"): one PROCESS per GPU, host code in C++ over the C ABI.
//
//     y = W_down (W_up x) mod t  =  sum_r  W_down[:, S_r] ( W_up[S_r, :] x ),      S_r = rank r's slice of the 3072 inner features
//
// Rank r holds the rows S_r of W_up (768 -> 3072/world) and the matching columns of W_down (3072/world -> 768) as two PackedLinear
// layers, applies them to the SAME input ciphertext, and the ranks exchange ONE partial ciphertext each with the path's only
// collective - the RCCL all-gather of encrypted partials (deeppowers::fhe::Communicator) - and sum them locally.  No secret key is
// needed on the evaluating side; this demo creates the key pair, the evaluation keys and the input ciphertext from one TestSeed in
// every process (what a client would ship to all ranks), so that all ranks hold identical operands.
// The decrypted result is compared with the plaintext product; it does not depend on <world> by construction.
//
//   sharded_ffn <world> [reps = 2] [first_device = 0] [emulate]        (world in {1, 2, 4, 8}; `emulate`: one process plays all ranks on one GPU)
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "deeppowers/fhe.hpp"

using namespace deeppowers::fhe;

// emulate = true: ONE process on one GPU plays every rank in turn (slices, hand-over and the sum are the real ones; the all-gather is a
// host copy) - what a single-GPU box can check of the world > 1 layouts.
static int run_rank(int rank, int world, int reps, int device, const std::string& dir, bool emulate) {
    try {
        const size_t d = 768, h = 3072, hs = h / (size_t)world;
        FheParams p = FheParams::n8192_l6();
        const uint64_t special = p.moduli.back(), special_psi = p.psi.back();
        p.moduli.pop_back(); p.psi.pop_back();
        const size_t n = p.n();
        Context ctx(p, device);
        Evaluator ev(ctx);
        // identical key material and input in every process (see the header): TestSeed is for reproducible demos only
        KeyGenerator kg(ctx, TestSeed{41});
        Encryptor enc(ctx, kg.secret_key(), TestSeed{42});
        Decryptor dec(ctx, kg.secret_key());
        BatchEncoder be(ctx, 65537);
        const uint64_t t = be.plain_modulus();
        HybridKeySwitcher hks(ctx, kg.secret_key(), special, special_psi, TestSeed{43});

        std::vector<uint8_t> id(128);
        const std::string id_path = dir + "/rccl_id";
        if (emulate) {
        } else if (rank == 0) {
            id = Communicator::unique_id();
            std::ofstream(id_path + ".tmp", std::ios::binary).write(reinterpret_cast<const char*>(id.data()), 128);
            std::rename((id_path + ".tmp").c_str(), id_path.c_str());
        } else {
            for (int tries = 0;; ++tries) {
                std::ifstream f(id_path, std::ios::binary);
                if (f && f.read(reinterpret_cast<char*>(id.data()), 128)) break;
                if (tries > 6000) { std::fprintf(stderr, "rank %d: no RCCL id after 60 s\n", rank); return 3; }
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        std::unique_ptr<Communicator> comm;
        if (!emulate) comm.reset(new Communicator(id, rank, world, device));

        // the whole model and input from one stream of pseudo-random 8-bit values (every rank generates all of it, keeps its slice)
        uint64_t s = 777;
        auto rnd = [&](uint64_t m) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (s >> 33) % m; };
        std::vector<uint64_t> Wu(h * d), Wd(d * h), x(d), mid(h), want(d);
        for (auto& v : Wu) v = (t + rnd(255) - 127) % t;
        for (auto& v : Wd) v = (t + rnd(255) - 127) % t;
        for (auto& v : x) v = (t + rnd(255) - 127) % t;
        for (size_t r = 0; r < h; ++r) {
            unsigned __int128 acc = 0;
            for (size_t c = 0; c < d; ++c) acc += (unsigned __int128)Wu[r * d + c] * x[c];
            mid[r] = (uint64_t)(acc % t);
        }
        for (size_t r = 0; r < d; ++r) {
            unsigned __int128 acc = 0;
            for (size_t c = 0; c < h; ++c) acc += (unsigned __int128)Wd[r * h + c] * mid[c];
            want[r] = (uint64_t)(acc % t);
        }
        std::vector<uint64_t> slots(n);
        std::vector<int64_t> coeffs(n);
        Ciphertext cx(ctx, 2, 1), c1(ctx, 2, 1), tmp(ctx, 2, 1), c1r(ctx, 2, 1), part(ctx, 2, 1), gathered(ctx, 2, (size_t)world), total(ctx, 2, 1);
        std::vector<uint64_t> host_gather(gathered.words());
        double ms = 0;
        std::unique_ptr<PackedLinear> down_keep;
        for (int rr = emulate ? 0 : rank; rr < (emulate ? world : rank + 1); ++rr) {
            std::vector<uint64_t> Wu_r(hs * d), Wd_r(d * hs);
            for (size_t r = 0; r < hs; ++r) std::memcpy(&Wu_r[r * d], &Wu[(rr * hs + r) * d], d * sizeof(uint64_t));
            for (size_t r = 0; r < d; ++r) std::memcpy(&Wd_r[r * hs], &Wd[r * h + rr * hs], hs * sizeof(uint64_t));
            PackedLinear up(ctx, be, hks, Wu_r.data(), hs, d);
            std::unique_ptr<PackedLinear> down(new PackedLinear(ctx, be, hks, Wd_r.data(), d, hs));
            if (up.output_ciphertexts() != 1 || down->output_ciphertexts() != 1) throw Exception(ErrorCode::INVALID_STATE, "one output ciphertext per layer expected");
            // hand-over up -> down.  up's output either already repeats with a period that is down's input period (slice < 1024), or sits
            // once in slots [0, P) of slot row 0 and is replicated: doubling rotations to the right inside the row, then the row swap.
            const size_t row = n / 2, period = down->input_period();
            std::vector<uint32_t> fill;
            if (hs >= up.input_period()) {
                for (size_t w = period; w < row; w *= 2) fill.push_back(be.galois_element(-(int)w));
                fill.push_back((uint32_t)(2 * n - 1));
                for (uint32_t g : fill) hks.add_galois_element(g);
            }
            if (rr == (emulate ? 0 : rank)) {
                up.pack_input(x.data(), slots.data());
                be.encode(slots.data(), coeffs.data());
                enc.encrypt_exact(coeffs.data(), t, cx);
            }
            auto forward = [&] {
                up.apply(cx, c1);
                Ciphertext* cur = &c1;
                Ciphertext* nxt = &c1r;
                for (uint32_t g : fill) {
                    hks.apply_galois(*cur, g, tmp);
                    ev.add(*cur, tmp, *nxt);
                    std::swap(cur, nxt);
                }
                down->apply(*cur, part);                 // this rank's partial: W_down[:, S_r] (W_up[S_r, :] x)
                if (!emulate) {
                    comm->all_gather(part, gathered);    // the only exchange: one ciphertext per rank over RCCL / xGMI
                    ev.reduce_sum(gathered, total);      // every rank ends with the same encrypted y
                }
            };
            forward();
            ctx.synchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) forward();
            ctx.synchronize();
            ms = std::max(ms, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3 / reps);
            if (emulate) part.copy_to_host(&host_gather[(size_t)rr * part.words()]);
            down_keep = std::move(down);
        }
        if (emulate) {
            gathered.copy_from_host(host_gather.data());
            ev.reduce_sum(gathered, total);
            ctx.synchronize();
        }
        PackedLinear& down = *down_keep;

        std::vector<uint64_t> dm(n), got(n), y(d);
        dec.decrypt_exact(total, t, dm.data());
        be.decode(dm.data(), got.data());
        down.unpack_output(got.data(), y.data());
        size_t bad = 0;
        uint64_t sum = 0;
        for (size_t r = 0; r < d; ++r) { bad += y[r] != want[r]; sum = ((sum << 7) | (sum >> 57)) ^ y[r]; }
        if (bad) { std::fprintf(stderr, "rank %d: %zu of %zu outputs differ from W_down (W_up x) mod t\n", rank, bad, d); return 5; }
        std::ofstream(dir + "/r" + std::to_string(rank)) << sum << " " << ms << "\n";
        return 0;
    } catch (const Exception& e) {
        std::fprintf(stderr, "rank %d: deeppowers::fhe error %d: %s\n", rank, (int)e.code(), e.what());
        return 2;
    }
}

int main(int argc, char** argv) {
    const int world = argc > 1 ? std::atoi(argv[1]) : 1;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 2;
    const int first_device = argc > 3 ? std::atoi(argv[3]) : 0;
    const bool emulate = argc > 4 && !std::strcmp(argv[4], "emulate");
    if (!(world == 1 || world == 2 || world == 4 || world == 8) || reps < 1) { std::fprintf(stderr, "usage: sharded_ffn <world in 1|2|4|8> [reps] [first_device]\n"); return 1; }
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);   // dmabuf IPC: what this driver supports across processes
    char tmpl[] = "/tmp/dpfhe_ffn_XXXXXX";
    if (!mkdtemp(tmpl)) { std::perror("mkdtemp"); return 1; }
    const std::string dir = tmpl;
    std::vector<pid_t> kids;
    const int procs = emulate ? 1 : world;
    for (int r = 0; r < procs; ++r) {
        const pid_t pid = fork();   // before any HIP call in this process
        if (pid < 0) { std::perror("fork"); return 1; }
        if (pid == 0) _exit(run_rank(r, world, reps, first_device + r, dir, emulate));
        kids.push_back(pid);
    }
    int bad = 0;
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) ++bad;
    }
    uint64_t ref = 0;
    double worst_ms = 0;
    for (int r = 0; r < procs && !bad; ++r) {
        std::ifstream f(dir + "/r" + std::to_string(r));
        uint64_t sum = 0; double ms = 0;
        if (!(f >> sum >> ms)) { ++bad; break; }
        if (r == 0) ref = sum; else if (sum != ref) { std::fprintf(stderr, "rank %d decrypts a different result\n", r); ++bad; }
        if (ms > worst_ms) worst_ms = ms;
    }
    for (int r = 0; r < world; ++r) std::remove((dir + "/r" + std::to_string(r)).c_str());
    std::remove((dir + "/rccl_id").c_str());
    rmdir(dir.c_str());
    if (bad) { std::printf("FAILED\n"); return 1; }
    std::printf("{\"host\": \"c++\", \"block\": \"ffn_linear_tensor_parallel\", \"world\": %d, \"emulated_on_one_gpu\": %s, \"inner_per_rank\": %d, \"ms_per_token\": %.3f, \"result_checksum\": \"%016llx\", "
                "\"correct\": true}\nOK\n", world, emulate ? "true" : "false", 3072 / world, worst_ms, (unsigned long long)ref);
    return 0;
}
