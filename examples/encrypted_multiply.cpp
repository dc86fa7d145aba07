// encrypted_multiply.cpp - the FHE hot path end to end through the C++ operator API, in the style of the reference's
// examples (/root/reference/examples/basic_generation.cpp:11-26: construct, run, print).
//
//   g++ -O2 -std=c++17 -Iinclude examples/encrypted_multiply.cpp -o encrypted_multiply \
//       -Ldeeppowers_amd -ldpfhe_api -ldpfhe_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/deeppowers_amd -Wl,-rpath,/opt/rocm/lib
#include <cstdio>
#include <vector>

#include "deeppowers/fhe.hpp"

using namespace deeppowers::fhe;

int main() {
    try {
        const FheParams params = FheParams::n4096_l4();   // N = 4096, 4 x 60-bit RNS limbs: the benchmark configuration
        Context ctx(params, /*device_id=*/0);
        Evaluator evaluator(ctx);
        KeyGenerator keygen(ctx);
        Encryptor encryptor(ctx, keygen.secret_key());
        Decryptor decryptor(ctx, keygen.secret_key());
        RelinKeys relin_keys(ctx);
        keygen.create_relin_keys(relin_keys);

        const size_t n = params.n();
        std::vector<int64_t> a(n, 0), b(n, 0), out(n);
        a[0] = 3; a[1] = 1;            // 3 + X
        b[0] = 5; b[n - 1] = 2;        // 5 + 2 X^(N-1)
        const unsigned log2_scale = 45;

        Ciphertext ct_a(ctx), ct_b(ctx), prod(ctx, 3), relin(ctx, 2);
        encryptor.encrypt(a.data(), log2_scale, ct_a);
        encryptor.encrypt(b.data(), log2_scale, ct_b);
        evaluator.multiply(ct_a, ct_b, prod);            // tensor product: one fused HIP kernel
        evaluator.relinearize(prod, relin_keys, relin);  // back to 2 components
        decryptor.decrypt(relin, 2 * log2_scale, out.data());

        // (3 + X)(5 + 2 X^(N-1)) = 15 + 5X + 6 X^(N-1) + 2 X^N = 13 + 5X + 6 X^(N-1)      (X^N = -1)
        std::printf("decrypted product: %lld + %lld X + ... + %lld X^(N-1)\n", (long long)out[0], (long long)out[1], (long long)out[n - 1]);
        const bool ok = out[0] == 13 && out[1] == 5 && out[n - 1] == 6;
        std::printf(ok ? "OK\n" : "MISMATCH\n");
        return ok ? 0 : 1;
    } catch (const Exception& e) {
        std::fprintf(stderr, "deeppowers::fhe error %d: %s\n", (int)e.code(), e.what());
        return 2;
    }
}
