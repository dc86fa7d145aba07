// GPU test of the C++ facade (deeppowers::fhe) against the C oracle.  Built and run by
// tests/test_gpu_cpp_api.py (-m gpu).  Exit code 0 = all checks passed.
#include <algorithm>
#include <cmath>
#include <sstream>
#include <cstdio>
#include <cstring>
#include <vector>

#include "deeppowers/fhe.hpp"

extern "C" {
struct orc_ctx;
int orc_ctx_create(orc_ctx** out, uint32_t log2n, uint32_t n_limbs, const uint64_t* moduli, const uint64_t* psi);
void orc_ctx_destroy(orc_ctx* c);
void orc_fill_splitmix(const orc_ctx* c, uint64_t* out, size_t n_rns_polys, uint64_t seed);
void orc_ct_mul(const orc_ctx* c, uint64_t* out3, const uint64_t* a2, const uint64_t* b2, size_t batch, int threads);
void orc_ntt_fwd(const orc_ctx* c, uint64_t* io, size_t n_rns_polys, int threads);
void orc_relinearize(const orc_ctx* c, uint64_t* out2, const uint64_t* in3, const uint64_t* evk, size_t batch, int threads);
void orc_reduce_sum(const orc_ctx* c, uint64_t* out, const uint64_t* in, size_t count, size_t comps);
void orc_matvec_plain(const orc_ctx* c, uint64_t* y, const uint64_t* W, const uint64_t* x, size_t rows, size_t cols, size_t comps, int threads);
}

using namespace deeppowers::fhe;
static int failures = 0;
#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static void run(const FheParams& p, size_t batch) {
    const size_t L = p.n_limbs(), n = p.n();
    orc_ctx* orc = nullptr;
    CHECK(orc_ctx_create(&orc, p.log2_n, (uint32_t)L, p.moduli.data(), p.psi.data()) == 0);
    std::vector<uint64_t> a(batch * 2 * L * n), b(a.size()), want(batch * 3 * L * n), got(want.size());
    orc_fill_splitmix(orc, a.data(), batch * 2, 1001);
    orc_fill_splitmix(orc, b.data(), batch * 2, 1002);
    orc_ct_mul(orc, want.data(), a.data(), b.data(), batch, 0);

    Context ctx(p, 0);
    Evaluator ev(ctx);
    Ciphertext A(ctx, 2, batch), B(ctx, 2, batch), C(ctx, 3, batch);
    A.copy_from_host(a.data());
    B.copy_from_host(b.data());
    ev.multiply(A, B, C);
    ctx.synchronize();
    C.copy_to_host(got.data());
    CHECK(std::memcmp(got.data(), want.data(), want.size() * 8) == 0);

    // the SUM of the products taken in the NTT domain (coefficient-domain operands, NTT-domain products, ONE inverse transform of the sum): the inverse
    // transform is linear, so this is the sum of the coefficient-domain products word for word
    {
        std::vector<uint64_t> s_want(3 * L * n), s_got(s_want.size());
        orc_reduce_sum(orc, s_want.data(), want.data(), batch, 3);
        Ciphertext Cs(ctx, 3, batch, /*is_ntt=*/true), S(ctx, 3, 1, /*is_ntt=*/true);
        ev.multiply(A, B, Cs);
        ev.reduce_sum(Cs, S);
        ev.transform_from_ntt_inplace(S);
        ctx.synchronize();
        S.copy_to_host(s_got.data());
        CHECK(!S.is_ntt() && s_got == s_want);
    }

    // NTT-domain route: transform, multiply with an NTT-domain output, transform back
    ev.transform_to_ntt_inplace(A);
    ev.transform_to_ntt_inplace(B);
    std::vector<uint64_t> an(a);
    orc_ntt_fwd(orc, an.data(), batch * 2, 0);
    std::vector<uint64_t> an_got(a.size());
    A.copy_to_host(an_got.data());
    CHECK(an_got == an);
    Ciphertext Cn(ctx, 3, batch, /*is_ntt=*/true);
    ev.multiply(A, B, Cn);
    ev.transform_from_ntt_inplace(Cn);
    Cn.copy_to_host(got.data());
    CHECK(std::memcmp(got.data(), want.data(), want.size() * 8) == 0);

    // ct x pt matvec (rows=3, cols=batch) in the NTT domain
    const size_t rows = 3, cols = batch;
    std::vector<uint64_t> W(rows * cols * L * n), y_want(rows * 2 * L * n), y_got(y_want.size());
    orc_fill_splitmix(orc, W.data(), rows * cols, 1003);
    orc_matvec_plain(orc, y_want.data(), W.data(), an.data(), rows, cols, 2, 0);
    Plaintext Wd(ctx, rows * cols, true);
    Wd.copy_from_host(W.data());
    Ciphertext Y(ctx, 2, rows, true);
    ev.matvec_plain(Wd, A, Y);
    Y.copy_to_host(y_got.data());
    CHECK(y_got == y_want);

    // relinearisation of the product with random (canonical) key words: bit-exact vs the oracle
    {
        std::vector<uint64_t> evk(L * 2 * L * n), r_want(batch * 2 * L * n), r_got(r_want.size());
        orc_fill_splitmix(orc, evk.data(), L * 2, 1004);
        orc_relinearize(orc, r_want.data(), want.data(), evk.data(), batch, 0);
        RelinKeys K(ctx);
        K.copy_from_host(evk.data());
        Ciphertext R(ctx, 2, batch);
        ev.relinearize(C, K, R);
        R.copy_to_host(r_got.data());
        CHECK(r_got == r_want);
    }

    // N4: wire format round trip (header + payload) and rejection of a stream from another shape
    {
        std::stringstream ss;
        C.save(ss);
        const std::string blob = ss.str();
        CHECK(blob.size() == 40 + 8 * L + want.size() * 8 && blob.compare(0, 7, "DPFHEv1") == 0);
        Ciphertext D(ctx, 3, batch, /*is_ntt=*/true);
        D.load(ss);
        std::vector<uint64_t> back(want.size());
        D.copy_to_host(back.data());
        CHECK(back == want && !D.is_ntt());
        std::stringstream ss2(blob);
        try { Ciphertext E(ctx, 2, batch); E.load(ss2); CHECK(!"expected INVALID_ARGUMENT"); } catch (const Exception& e) { CHECK(e.code() == ErrorCode::INVALID_ARGUMENT); }
    }

    // error behaviour: exceptions with reference error codes
    try {
        Ciphertext bad(ctx, 2, batch, true);
        ev.multiply(bad, Ciphertext(ctx, 2, batch, false), C);
        CHECK(!"expected INVALID_STATE");
    } catch (const Exception& e) { CHECK(e.code() == ErrorCode::INVALID_STATE); }
    try {
        Ciphertext four(ctx, 4, 1);
        CHECK(!"expected INVALID_ARGUMENT");
    } catch (const Exception& e) { CHECK(e.code() == ErrorCode::INVALID_ARGUMENT); }
    orc_ctx_destroy(orc);
}

// N = 16384: no fused kernels - dpfhe_ct_mul / dpfhe_relinearize compose the batched transforms with streaming kernels (kernels_large.h)
static void large_ring() {
    FheParams p;
    p.log2_n = 14;
    for (uint64_t q : {1152921504606748673ull, 1152921504606683137ull, 1152921504606584833ull}) {   // pinned primes = 1 mod 32768
        auto pw = [q](uint64_t b, uint64_t e) { uint64_t r = 1; for (b %= q; e; e >>= 1) { if (e & 1) r = (uint64_t)((unsigned __int128)r * b % q); b = (uint64_t)((unsigned __int128)b * b % q); } return r; };
        uint64_t psi = 0;
        for (uint64_t g = 2; !psi; ++g) { const uint64_t z = pw(g, (q - 1) / 32768); if (pw(z, 16384) == q - 1) psi = z; }
        p.moduli.push_back(q); p.psi.push_back(psi);
    }
    const size_t L = p.n_limbs(), n = p.n(), batch = 2;
    orc_ctx* orc = nullptr;
    CHECK(orc_ctx_create(&orc, p.log2_n, (uint32_t)L, p.moduli.data(), p.psi.data()) == 0);
    std::vector<uint64_t> a(batch * 2 * L * n), b(a.size()), want(batch * 3 * L * n), got(want.size());
    orc_fill_splitmix(orc, a.data(), batch * 2, 2001);
    orc_fill_splitmix(orc, b.data(), batch * 2, 2002);
    orc_ct_mul(orc, want.data(), a.data(), b.data(), batch, 0);
    Context ctx(p, 0);
    Evaluator ev(ctx);
    Ciphertext A(ctx, 2, batch), B(ctx, 2, batch), C(ctx, 3, batch);
    A.copy_from_host(a.data());
    B.copy_from_host(b.data());
    ev.multiply(A, B, C);
    ctx.synchronize();
    C.copy_to_host(got.data());
    CHECK(got == want);
    ev.transform_to_ntt_inplace(A);
    ev.transform_to_ntt_inplace(B);
    Ciphertext Cn(ctx, 3, batch, /*is_ntt=*/true);
    ev.multiply(A, B, Cn);
    ev.transform_from_ntt_inplace(Cn);
    ctx.synchronize();
    Cn.copy_to_host(got.data());
    CHECK(got == want);
    {   // relinearisation above N = 8192 is composed behind the C ABI too (random canonical key words: bit-exact vs the oracle)
        std::vector<uint64_t> evk(L * 2 * L * n), r_want(batch * 2 * L * n), r_got(r_want.size());
        orc_fill_splitmix(orc, evk.data(), L * 2, 2003);
        orc_relinearize(orc, r_want.data(), want.data(), evk.data(), batch, 0);
        RelinKeys K(ctx);
        K.copy_from_host(evk.data());
        Ciphertext R(ctx, 2, batch);
        ev.relinearize(C, K, R);
        R.copy_to_host(r_got.data());
        CHECK(r_got == r_want);
    }
    orc_ctx_destroy(orc);
}

// N3: rectangular packed layers (row blocks in the windows of one ciphertext, several output ciphertexts, wrapped diagonals +
// folding for wide inputs) against plain modular arithmetic, at a small ring so that every case takes well under a second
static void packed_rect(unsigned log2n) {
    FheParams p = FheParams::n8192_l6();
    const size_t n = (size_t)1 << log2n;
    const uint64_t special = p.moduli.back();
    auto pw = [](uint64_t b, uint64_t e, uint64_t q) { uint64_t r = 1; for (b %= q; e; e >>= 1) { if (e & 1) r = (uint64_t)((unsigned __int128)r * b % q); b = (uint64_t)((unsigned __int128)b * b % q); } return r; };
    for (size_t l = 0; l < p.moduli.size(); ++l) p.psi[l] = pw(p.psi[l], 8192 / n, p.moduli[l]);
    const uint64_t special_psi = p.psi.back();
    p.log2_n = log2n; p.moduli.pop_back(); p.psi.pop_back();
    Context ctx(p, 0);
    KeyGenerator kg(ctx, TestSeed{31});
    Encryptor enc(ctx, kg.secret_key(), TestSeed{32});
    Decryptor dec(ctx, kg.secret_key());
    BatchEncoder be(ctx, 65537);
    const uint64_t t = be.plain_modulus();
    HybridKeySwitcher hks(ctx, kg.secret_key(), special, special_psi, TestSeed{33});
    uint64_t s = 17;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (s >> 33) % t; };
    // (out, in): blocks sharing one ciphertext | ragged sizes | more blocks than windows (2 outputs) | one block | wide input
    // with folding | wide input, ragged | single-row output
    const size_t shapes[][2] = {{96, 32}, {77, 24}, {n + 40, 16}, {20, 32}, {16, 128}, {12, 100}, {1, 64}, {n / 2, n / 2}};
    for (auto& sh : shapes) {
        const size_t out = sh[0], in = sh[1];
        std::vector<uint64_t> W(out * in), x(in), want(out), slots(n);
        for (auto& v : W) v = rnd();
        for (auto& v : x) v = rnd();
        for (size_t r = 0; r < out; ++r) {
            unsigned __int128 acc = 0;
            for (size_t c = 0; c < in; ++c) acc += (unsigned __int128)W[r * in + c] * x[c];
            want[r] = (uint64_t)(acc % t);
        }
        PackedLinear lin(ctx, be, hks, W.data(), out, in);
        const size_t outs = lin.output_ciphertexts();
        CHECK(lin.baby_steps() * lin.giant_steps() == lin.dim());
        std::vector<int64_t> cx(n);
        lin.pack_input(x.data(), slots.data());
        be.encode(slots.data(), cx.data());
        Ciphertext ct(ctx, 2, 1), cy(ctx, 2, outs);
        enc.encrypt_exact(cx.data(), t, ct);
        lin.apply(ct, cy);
        lin.apply(ct, cy);           // scratch reuse: a second application gives the same answer
        ctx.synchronize();
        std::vector<uint64_t> dm(outs * n), got(outs * n), y(out);
        dec.decrypt_exact(cy, t, dm.data());
        for (size_t o = 0; o < outs; ++o) be.decode(dm.data() + o * n, got.data() + o * n);
        lin.unpack_output(got.data(), y.data());
        CHECK(y == want);
        // several tokens per launch (T = 5: the 4 + 1 grouping of the multi-right-hand-side matvec), then a single token again
        // (scratch sized for 5 tokens, used for 1)
        const size_t T = 5;
        std::vector<uint64_t> xs(T * in), wants(T * out);
        std::vector<int64_t> cxs(T * n);
        for (size_t tk = 0; tk < T; ++tk) {
            for (size_t c = 0; c < in; ++c) xs[tk * in + c] = rnd();
            for (size_t r = 0; r < out; ++r) {
                unsigned __int128 acc = 0;
                for (size_t c = 0; c < in; ++c) acc += (unsigned __int128)W[r * in + c] * xs[tk * in + c];
                wants[tk * out + r] = (uint64_t)(acc % t);
            }
            lin.pack_input(&xs[tk * in], slots.data());
            be.encode(slots.data(), &cxs[tk * n]);
        }
        Ciphertext ctT(ctx, 2, T), cyT(ctx, 2, outs * T);
        enc.encrypt_exact(cxs.data(), t, ctT);
        lin.apply(ctT, cyT);
        ctx.synchronize();
        std::vector<uint64_t> dmT(outs * T * n), gotT(outs * n);
        dec.decrypt_exact(cyT, t, dmT.data());
        for (size_t tk = 0; tk < T; ++tk) {
            for (size_t o = 0; o < outs; ++o) be.decode(dmT.data() + (o * T + tk) * n, gotT.data() + o * n);   // output o of token tk: item o * T + tk
            lin.unpack_output(gotT.data(), y.data());
            CHECK(std::equal(y.begin(), y.end(), wants.begin() + tk * out));
        }
        lin.apply(ct, cy);
        ctx.synchronize();
        dec.decrypt_exact(cy, t, dm.data());
        for (size_t o = 0; o < outs; ++o) be.decode(dm.data() + o * n, got.data() + o * n);
        lin.unpack_output(got.data(), y.data());
        CHECK(y == want);
    }
}

// default-constructed key material must come from the OS CSPRNG - two generators never agree, two encryptions
// of one message never agree, errors are small and centred, and everything still decrypts.
static void os_randomness(const FheParams& p) {
    const size_t n = p.n();
    Context ctx(p, 0);
    KeyGenerator kg1(ctx), kg2(ctx);
    CHECK(kg1.secret_key().coefficients() != kg2.secret_key().coefficients());
    long nz = 0;
    for (int8_t v : kg1.secret_key().coefficients()) { CHECK(v >= -1 && v <= 1); nz += v != 0; }
    CHECK(nz > (long)n / 2 && nz < (long)n * 5 / 6);          // ~2/3 non-zero
    KeyGenerator det1(ctx, TestSeed{5}), det2(ctx, TestSeed{5});
    CHECK(det1.secret_key().coefficients() == det2.secret_key().coefficients());   // the testing path stays reproducible
    Encryptor e1(ctx, kg1.secret_key()), e2(ctx, kg1.secret_key());
    Decryptor dec(ctx, kg1.secret_key());
    std::vector<int64_t> m(n), out(n);
    for (size_t i = 0; i < n; ++i) m[i] = (int64_t)(i % 199) - 99;
    Ciphertext c1(ctx, 2, 1), c2(ctx, 2, 1);
    e1.encrypt(m.data(), 40, c1);
    e2.encrypt(m.data(), 40, c2);
    std::vector<uint64_t> h1(c1.words()), h2(c2.words());
    c1.copy_to_host(h1.data()); c2.copy_to_host(h2.data());
    size_t same = 0;
    for (size_t i = 0; i < h1.size(); ++i) same += h1[i] == h2[i];
    CHECK(same < 4);                                            // independent (a, e): no word-level coincidences beyond chance
    dec.decrypt(c1, 40, out.data()); CHECK(out == m);
    dec.decrypt(c2, 40, out.data()); CHECK(out == m);
    // the noise itself: decrypt at scale 0 -> m * 2^40 + e, |e| <= 21, mean ~ 0
    dec.decrypt(c1, 0, out.data());
    double mean = 0;
    for (size_t i = 0; i < n; ++i) {
        const int64_t e = out[i] - m[i] * (int64_t(1) << 40);
        CHECK(e >= -21 && e <= 21);
        mean += (double)e;
    }
    CHECK(std::abs(mean / (double)n) < 0.5);
}

// N2 + N1 end to end: encrypt -> multiply -> (relinearize) -> decrypt == negacyclic product of the messages
static void end_to_end(const FheParams& p, size_t batch) {
    const size_t n = p.n();
    Context ctx(p, 0);
    Evaluator ev(ctx);
    KeyGenerator kg(ctx, TestSeed{11});
    Encryptor enc(ctx, kg.secret_key(), TestSeed{12});
    Decryptor dec(ctx, kg.secret_key());
    RelinKeys rk(ctx);
    kg.create_relin_keys(rk);
    std::vector<int64_t> m1(batch * n), m2(batch * n), out(batch * n), want(batch * n, 0);
    uint64_t s = 99;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (int64_t)((s >> 33) % 201) - 100; };
    for (auto& v : m1) v = rnd();
    for (auto& v : m2) v = rnd();
    for (size_t b = 0; b < batch; ++b)
        for (size_t i = 0; i < n; ++i)
            for (size_t j = 0; j < n; ++j) {
                const int64_t pr = m1[b * n + i] * m2[b * n + j];
                if (i + j < n) want[b * n + i + j] += pr; else want[b * n + i + j - n] -= pr;
            }
    const unsigned scale = 45;
    Ciphertext c1(ctx, 2, batch), c2(ctx, 2, batch), c3(ctx, 3, batch), cr(ctx, 2, batch);
    enc.encrypt(m1.data(), scale, c1);
    enc.encrypt(m2.data(), scale, c2);
    dec.decrypt(c1, scale, out.data());
    CHECK(out == m1);                                   // fresh ciphertext decrypts to its message
    ev.multiply(c1, c2, c3);
    dec.decrypt(c3, 2 * scale, out.data());
    CHECK(out == want);                                 // 3-component product decrypts to m1 * m2 in Z[X]/(X^N+1)
    ev.relinearize(c3, rk, cr);
    dec.decrypt(cr, 2 * scale, out.data());
    CHECK(out == want);                                 // and so does its relinearisation
    // public-key encryption: same messages, encrypted with pk only, decrypt under the secret key and multiply homomorphically
    {
        PublicKey pk(ctx);
        kg.create_public_key(pk);
        Encryptor penc(ctx, pk, TestSeed{13});
        Ciphertext p1(ctx, 2, batch), p2(ctx, 2, batch), p3(ctx, 3, batch);
        penc.encrypt(m1.data(), scale, p1);
        penc.encrypt(m2.data(), scale, p2);
        dec.decrypt(p1, scale, out.data());
        CHECK(out == m1);
        ev.multiply(p1, p2, p3);
        dec.decrypt(p3, 2 * scale, out.data());
        CHECK(out == want);
    }
    // N3: m(X) -> m(X^g) under encryption.  RNS-digit key switching adds ~ L N q sigma ~ 2^77 of noise, so the
    // rotated message needs a scale well above that (the product above sits at 2^90 already).
    const unsigned gscale = 100;
    Ciphertext cg(ctx, 2, batch);
    enc.encrypt(m1.data(), gscale, cg);
    for (uint32_t g : {3u, (uint32_t)(2 * n - 1), 25u}) {
        GaloisKeys gk(ctx, g);
        kg.create_galois_keys(gk);
        Ciphertext rot(ctx, 2, batch);
        ev.apply_galois(cg, gk, rot);
        dec.decrypt(rot, gscale, out.data());
        std::vector<int64_t> wantg(batch * n, 0);
        for (size_t b = 0; b < batch; ++b)
            for (size_t i = 0; i < n; ++i) {
                const size_t idx = (i * (size_t)g) & (2 * n - 1);
                if (idx < n) wantg[b * n + idx] = m1[b * n + i]; else wantg[b * n + idx - n] = -m1[b * n + i];
            }
        CHECK(out == wantg);
    }
    // hybrid key switching (special prime = the 6th pinned prime): the product decrypts at scale 2^60 already,
    // a rotation of a FRESH ciphertext at scale 2^30 - impossible with the plain RNS-digit keys above (noise ~2^77)
    {
        HybridKeySwitcher hks(ctx, kg.secret_key(), 1152921504606109697ull, /*psi for N=4096:*/ 279138086580908ull, TestSeed{3});
        Ciphertext lo1(ctx, 2, batch), lo2(ctx, 2, batch), lo3(ctx, 3, batch), lor(ctx, 2, batch), rot(ctx, 2, batch);
        enc.encrypt(m1.data(), 30, lo1);
        enc.encrypt(m2.data(), 30, lo2);
        ev.multiply(lo1, lo2, lo3);
        hks.relinearize(lo3, lor);
        dec.decrypt(lor, 60, out.data());
        CHECK(out == want);
        hks.add_galois_element(5);
        hks.apply_galois(lo1, 5, rot);
        dec.decrypt(rot, 30, out.data());
        std::vector<int64_t> wantg(batch * n, 0);
        for (size_t b = 0; b < batch; ++b)
            for (size_t i = 0; i < n; ++i) {
                const size_t idx = (i * 5) & (2 * n - 1);
                if (idx < n) wantg[b * n + idx] = m1[b * n + i]; else wantg[b * n + idx - n] = -m1[b * n + i];
            }
        CHECK(out == wantg);
    }

    // N1 second half: rescale the relinearised product to the next level; the same secret (same seed) decrypts it there
    {
        const FheParams p2 = p.drop_last_limb();
        Context ctx2(p2, 0);
        KeyGenerator kg2(ctx2, TestSeed{11});             // same seed -> same ternary secret
        Decryptor dec2(ctx2, kg2.secret_key());
        Ciphertext low(ctx2, 2, batch);
        ev.rescale(cr, low);
        ctx.synchronize();
        // scale 2^90 / q_last: decrypt at the largest power of two below it and compare after the same division
        const double ql = (double)p.moduli.back();
        std::vector<int64_t> got(batch * n);
        dec2.decrypt(low, 30, got.data());               // 2^90 / q_last ~ 2^30 (q_last ~ 2^60)
        bool ok = true;
        for (size_t i = 0; i < got.size(); ++i) {
            const double expect = (double)want[i] * (1152921504606846976.0 / ql);   // * 2^60 / q_last
            if (std::fabs((double)got[i] - expect) > 2.0) ok = false;
        }
        CHECK(ok);
    }
    try { dec.decrypt(c3, 0, out.data()); CHECK(!"expected RUNTIME_ERROR"); } catch (const Exception& e) { CHECK(e.code() == ErrorCode::RUNTIME_ERROR); }
}

// N3: slot packing, exact (mod t) encryption, slot rotations under encryption and the baby-step/giant-step packed
// matrix-vector product, against plain modular arithmetic on the host
static void packed(unsigned log2n, size_t d) {
    FheParams p = FheParams::n8192_l6();          // six pinned primes = 1 mod 16384; psi_N = psi_8192^(8192/N)
    const size_t n = (size_t)1 << log2n, row = n / 2;
    const uint64_t special = p.moduli.back();
    auto pw = [](uint64_t b, uint64_t e, uint64_t q) { uint64_t r = 1; for (b %= q; e; e >>= 1) { if (e & 1) r = (uint64_t)((unsigned __int128)r * b % q); b = (uint64_t)((unsigned __int128)b * b % q); } return r; };
    for (size_t l = 0; l < p.moduli.size(); ++l) p.psi[l] = pw(p.psi[l], 8192 / n, p.moduli[l]);
    const uint64_t special_psi = p.psi.back();
    p.log2_n = log2n; p.moduli.pop_back(); p.psi.pop_back();    // 5 data limbs + the 6th as the special prime
    Context ctx(p, 0);
    KeyGenerator kg(ctx, TestSeed{21});
    Encryptor enc(ctx, kg.secret_key(), TestSeed{22});
    Decryptor dec(ctx, kg.secret_key());
    BatchEncoder be(ctx, 65537);
    const uint64_t t = be.plain_modulus();
    uint64_t s = 7;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (s >> 33) % t; };
    // encoder round trip, and slot-wise product = polynomial product
    std::vector<uint64_t> va(n), vb(n), got(n);
    for (auto& v : va) v = rnd();
    for (auto& v : vb) v = rnd();
    std::vector<int64_t> ca(n), cb(n);
    be.encode(va.data(), ca.data());
    be.encode(vb.data(), cb.data());
    {
        std::vector<uint64_t> um(n);
        for (size_t i = 0; i < n; ++i) um[i] = (uint64_t)(ca[i] < 0 ? ca[i] + (int64_t)t : ca[i]);
        be.decode(um.data(), got.data());
        CHECK(got == va);
    }
    // exact encryption round trip, plaintext multiply in slots, rotations of the rows, row swap
    Evaluator ev(ctx);
    Ciphertext cx(ctx, 2, 1), cy(ctx, 2, 1);
    enc.encrypt_exact(ca.data(), t, cx);
    std::vector<uint64_t> dm(n);
    dec.decrypt_exact(cx, t, dm.data());
    be.decode(dm.data(), got.data());
    CHECK(got == va);
    HybridKeySwitcher hks(ctx, kg.secret_key(), special, special_psi, TestSeed{3});
    for (int rot : {1, 5, -3}) {
        const uint32_t g = be.galois_element(rot);
        hks.add_galois_element(g);
        hks.apply_galois(cx, g, cy);
        dec.decrypt_exact(cy, t, dm.data());
        be.decode(dm.data(), got.data());
        bool ok = true;
        for (size_t r = 0; r < row; ++r) {
            const size_t src = (r + (size_t)((rot % (long long)row + (long long)row) % (long long)row)) % row;
            ok = ok && got[r] == va[src] && got[row + r] == va[row + src];
        }
        CHECK(ok);                                  // both rows rotate left by `rot`
    }
    {
        const uint32_t g = (uint32_t)(2 * n - 1);
        hks.add_galois_element(g);
        hks.apply_galois(cx, g, cy);
        dec.decrypt_exact(cy, t, dm.data());
        be.decode(dm.data(), got.data());
        bool ok = true;
        for (size_t r = 0; r < row; ++r) ok = ok && got[r] == va[row + r] && got[row + r] == va[r];
        CHECK(ok);                                  // X -> X^(2N-1) swaps the rows
    }
    // packed y = W x
    std::vector<uint64_t> W(d * d), x(d), want(d, 0), slots(n);
    for (auto& v : W) v = rnd();
    for (auto& v : x) v = rnd();
    for (size_t r = 0; r < d; ++r) {
        unsigned __int128 acc = 0;
        for (size_t c = 0; c < d; ++c) acc += (unsigned __int128)W[r * d + c] * x[c];
        want[r] = (uint64_t)(acc % t);
    }
    for (size_t r = 0; r < row; ++r) slots[r] = slots[row + r] = x[r % d];
    std::vector<int64_t> cxv(n);
    be.encode(slots.data(), cxv.data());
    enc.encrypt_exact(cxv.data(), t, cx);
    PackedLinear lin(ctx, be, hks, W.data(), d);
    CHECK(lin.baby_steps() * lin.giant_steps() == d);
    lin.apply(cx, cy);
    ctx.synchronize();
    dec.decrypt_exact(cy, t, dm.data());
    be.decode(dm.data(), got.data());
    bool ok = true;
    for (size_t r = 0; r < row; ++r) ok = ok && got[r] == want[r % d] && got[row + r] == want[r % d];
    CHECK(ok);
}

int main() {
    try {
        packed(10, 64);
        packed(12, 16);
        packed_rect(10);
        large_ring();
        end_to_end(FheParams::n4096_l4(), 2);
        os_randomness(FheParams::n4096_l4());
        {   // a composite modulus with a root of order 2N must be rejected (inverses are Fermat powers)
            FheParams p = FheParams::n4096_l4();
            p.moduli[0] = 8193ull * 40961ull;   // both factors are 1 mod 8192; primality is checked before psi is even looked at
            try { Context bad(p, 0); CHECK(!"expected INVALID_ARGUMENT"); } catch (const Exception& e) { CHECK(e.code() == ErrorCode::INVALID_ARGUMENT); }
        }
        run(FheParams::config1(), 2);
        run(FheParams::n4096_l4(), 3);
        run(FheParams::n8192_l6(), 2);
        try {
            FheParams p = FheParams::n4096_l4();
            p.psi[0] = 3;
            Context bad(p, 0);
            CHECK(!"expected INVALID_ARGUMENT");
        } catch (const Exception& e) { CHECK(e.code() == ErrorCode::INVALID_ARGUMENT); }
    } catch (const std::exception& e) {
        std::printf("unexpected exception: %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
