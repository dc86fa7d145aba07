// sharded_ct_mul.cpp - BASELINE configs[3] in pure C++: batches of independent ciphertext pairs sharded over the GPUs of one node,
// one PROCESS per GPU, host code in C++ calling HIP through the C ABI, the only communication being the RCCL all-gather of one
// partial ciphertext per rank (deeppowers::fhe::Communicator = dpfhe_comm_*).  No Python, no torch.distributed.
//
//   sharded_ct_mul <world> [pairs_per_rank = 256] [steps = 3] [first_device = 0]
//
// The parent forks <world> children BEFORE any HIP call; rank r runs on device first_device + r.  The 128-byte RCCL id travels
// through a file in a private temporary directory (rank 0 writes it, the others wait for it) - the "host program's own
// rendezvous" the C ABI asks for.  Pair i of the GLOBAL batch is generated from i alone, so the global result does not depend on
// <world>: every rank prints the SHA-free checksum (xor-rotate of all words) of the final sum and the parent checks that all ranks
// agree AND that the sum equals a world-size-1 recomputation of the same global batch by one more child on the first device
// (run_reference: no communicator at all).  Stands in for the reference's all-gather call site,
// /root/reference/src/core/distributed/distributed_context.cpp:97-122.
//
//   g++ -O2 -std=c++17 -Iinclude -I/opt/rocm/include examples/sharded_ct_mul.cpp -o examples/sharded_ct_mul \
//       -Ldeeppowers_amd -ldpfhe_api -ldpfhe_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/deeppowers_amd -Wl,-rpath,/opt/rocm/lib
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "deeppowers/fhe.hpp"

using namespace deeppowers::fhe;

static uint64_t splitmix(uint64_t& s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// words of ciphertext pair `i` of the global batch (operand `which` = 0 for a, 1 for b): a function of i only
static void fill_pair(const FheParams& p, uint64_t i, int which, uint64_t* dst) {
    const size_t n = p.n(), L = p.n_limbs();
    uint64_t s = 0xC0FFEEull + 2 * i + (uint64_t)which;
    for (size_t w = 0; w < 2 * L * n; ++w) dst[w] = splitmix(s) % p.moduli[(w / n) % L];
}

static uint64_t checksum(const std::vector<uint64_t>& v) {
    uint64_t h = 0;
    for (uint64_t x : v) h = ((h << 7) | (h >> 57)) ^ x;
    return h;
}

static int run_rank(int rank, int world, size_t pairs, int steps, int device, const std::string& dir) {
    try {
        const FheParams p = FheParams::n4096_l4();
        const size_t n = p.n(), L = p.n_limbs();
        Context ctx(p, device);
        Evaluator ev(ctx);
        // rendezvous: rank 0 creates the id and publishes it with an atomic rename
        std::vector<uint8_t> id(128);
        const std::string id_path = dir + "/rccl_id";
        if (rank == 0) {
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
        Communicator comm(id, rank, world, device);

        // this rank's contiguous shard [rank * pairs, (rank + 1) * pairs) of the global batch
        std::vector<uint64_t> ha(pairs * 2 * L * n), hb(pairs * 2 * L * n);
        for (size_t i = 0; i < pairs; ++i) {
            fill_pair(p, (uint64_t)rank * pairs + i, 0, &ha[i * 2 * L * n]);
            fill_pair(p, (uint64_t)rank * pairs + i, 1, &hb[i * 2 * L * n]);
        }
        Ciphertext a(ctx, 2, pairs), b(ctx, 2, pairs), c(ctx, 3, pairs);
        Ciphertext partial(ctx, 3, 1), gathered(ctx, 3, (size_t)world), total(ctx, 3, 1);
        a.copy_from_host(ha.data());
        b.copy_from_host(hb.data());
        hipStream_t s;
        if (hipStreamCreate(&s) != hipSuccess) return 4;
        auto step = [&]() {
            ev.multiply(a, b, c, s);               // no communication during compute
            ev.reduce_sum(c, partial, s);          // shard-local: one partial ciphertext (3 L N words)
            comm.all_gather(partial, gathered, s); // the only exchange: RCCL all-gather over xGMI
            ev.reduce_sum(gathered, total, s);     // every rank ends with the same global sum
        };
        step();
        (void)hipStreamSynchronize(s);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < steps; ++i) step();
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> ht(total.words());
        total.copy_to_host(ht.data());
        uint64_t sum = checksum(ht);
        if (world == 1) {   // serial recomputation without the communicator
            Ciphertext t2(ctx, 3, 1);
            ev.multiply(a, b, c);
            ev.reduce_sum(c, t2);
            std::vector<uint64_t> h2(t2.words());
            t2.copy_to_host(h2.data());
            if (h2 != ht) { std::fprintf(stderr, "rank 0: all-gather path differs from the serial recomputation\n"); return 5; }
        }
        std::ofstream(dir + "/r" + std::to_string(rank)) << sum << " " << (double)pairs * steps / (ms * 1e-3) << "\n";
        (void)hipStreamDestroy(s);
        return 0;
    } catch (const Exception& e) {
        std::fprintf(stderr, "rank %d: deeppowers::fhe error %d: %s\n", rank, (int)e.code(), e.what());
        return 2;
    }
}

// World-size-1 recomputation of the SAME global batch (world * pairs pairs) on one device, without any communicator: shard after
// shard -> its partial, partials summed locally.  Agreement between ranks is not correctness; equality with this is.
static int run_reference(int world, size_t pairs, int device, const std::string& dir) {
    try {
        const FheParams p = FheParams::n4096_l4();
        const size_t n = p.n(), L = p.n_limbs();
        Context ctx(p, device);
        Evaluator ev(ctx);
        std::vector<uint64_t> ha(pairs * 2 * L * n), hb(pairs * 2 * L * n);
        Ciphertext a(ctx, 2, pairs), b(ctx, 2, pairs), c(ctx, 3, pairs), partials(ctx, 3, (size_t)world), total(ctx, 3, 1), one(ctx, 3, 1);
        const size_t ct3 = 3 * L * n;
        for (int r = 0; r < world; ++r) {
            for (size_t i = 0; i < pairs; ++i) {
                fill_pair(p, (uint64_t)r * pairs + i, 0, &ha[i * 2 * L * n]);
                fill_pair(p, (uint64_t)r * pairs + i, 1, &hb[i * 2 * L * n]);
            }
            a.copy_from_host(ha.data());
            b.copy_from_host(hb.data());
            ev.multiply(a, b, c);
            ev.reduce_sum(c, one);
            ctx.synchronize();
            if (hipMemcpy(partials.data() + (size_t)r * ct3, one.data(), ct3 * sizeof(uint64_t), hipMemcpyDeviceToDevice) != hipSuccess) return 4;
        }
        ev.reduce_sum(partials, total);
        std::vector<uint64_t> ht(total.words());
        total.copy_to_host(ht.data());
        std::ofstream(dir + "/ref") << checksum(ht) << "\n";
        return 0;
    } catch (const Exception& e) {
        std::fprintf(stderr, "reference: deeppowers::fhe error %d: %s\n", (int)e.code(), e.what());
        return 2;
    }
}

int main(int argc, char** argv) {
    const int world = argc > 1 ? std::atoi(argv[1]) : 1;
    const size_t pairs = argc > 2 ? (size_t)std::atol(argv[2]) : 256;
    const int steps = argc > 3 ? std::atoi(argv[3]) : 3;
    const int first_device = argc > 4 ? std::atoi(argv[4]) : 0;
    if (world < 1 || world > 64 || pairs == 0 || steps < 1) { std::fprintf(stderr, "usage: sharded_ct_mul <world> [pairs_per_rank] [steps] [first_device]\n"); return 1; }
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);   // dmabuf IPC: what this driver supports across processes
    char tmpl[] = "/tmp/dpfhe_shard_XXXXXX";
    if (!mkdtemp(tmpl)) { std::perror("mkdtemp"); return 1; }
    const std::string dir = tmpl;
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();   // before any HIP call in this process
        if (pid < 0) { std::perror("fork"); return 1; }
        if (pid == 0) _exit(run_rank(r, world, pairs, steps, first_device + r, dir));
        kids.push_back(pid);
    }
    int bad = 0;
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) ++bad;
    }
    uint64_t ref = 0;
    double rate = 0;
    for (int r = 0; r < world && !bad; ++r) {
        std::ifstream f(dir + "/r" + std::to_string(r));
        uint64_t sum = 0; double per_s = 0;
        if (!(f >> sum >> per_s)) { ++bad; break; }
        if (r == 0) ref = sum; else if (sum != ref) { std::fprintf(stderr, "rank %d holds a different global sum\n", r); ++bad; }
        rate += per_s;
    }
    bool matches_world1 = false;
    if (!bad) {   // one more child: the whole global batch at world size 1 on the first device
        const pid_t pid = fork();
        if (pid < 0) { std::perror("fork"); return 1; }
        if (pid == 0) _exit(run_reference(world, pairs, first_device, dir));
        int st = 0;
        waitpid(pid, &st, 0);
        uint64_t want = 0;
        std::ifstream f(dir + "/ref");
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0 || !(f >> want)) { std::fprintf(stderr, "the world-size-1 recomputation failed\n"); ++bad; }
        else if (want != ref) { std::fprintf(stderr, "the sharded global sum differs from the world-size-1 recomputation of the same batch\n"); ++bad; }
        else matches_world1 = true;
        std::remove((dir + "/ref").c_str());
    }
    for (int r = 0; r < world; ++r) std::remove((dir + "/r" + std::to_string(r)).c_str());
    std::remove((dir + "/rccl_id").c_str());
    rmdir(dir.c_str());
    if (bad) { std::printf("FAILED\n"); return 1; }
    std::printf("{\"host\": \"c++\", \"world\": %d, \"pairs_per_rank\": %zu, \"steps\": %d, \"global_sum_checksum\": \"%016llx\", \"matches_world1_recomputation\": %s, \"ct_mul_per_s\": %.1f}\nOK\n",
                world, pairs, steps, (unsigned long long)ref, matches_world1 ? "true" : "false", rate);
    return 0;
}
