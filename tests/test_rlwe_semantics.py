"""Semantic end-to-end check (test-side only): a toy RLWE encryption built from
Python big ints shows that the tensor product computed on the ring level is the product under decryption:
    phase(ct) = c0 + c1 s (+ c2 s^2)  and  phase(ct_a (x) ct_b) = phase(ct_a) * phase(ct_b)  in R_Q, Q = prod q_i.
CPU leg uses the C oracle; the GPU leg (-m gpu) runs the HIP fused multiply through the C ABI."""
import numpy as np
import pytest

from deeppowers_amd.params import PRIMES_60, FheParams
from oracle import pyoracle as po
from oracle.cbind import Oracle


def small_params(log2n=8, limbs=3):
    n = 1 << log2n
    qs = tuple(PRIMES_60[i][0] for i in range(limbs))
    return FheParams(log2n, qs, tuple(pow(PRIMES_60[i][2], 8192 // n, qs[i]) for i in range(limbs)))


def crt_compose(residues, moduli):
    """residues[l][k] -> list of big ints mod Q (Garner-free: direct CRT with big ints)."""
    Q = 1
    for q in moduli:
        Q *= q
    out = [0] * len(residues[0])
    for l, q in enumerate(moduli):
        Ql = Q // q
        inv = pow(Ql, -1, q)
        for k, r in enumerate(residues[l]):
            out[k] = (out[k] + int(r) * inv % q * Ql) % Q
    return out, Q


def encrypt(rng, p, s, msg, delta):
    """(c0, c1) = (-a s + e + delta m, a) per limb; returns arrays [2][L][N] and the exact integer phase."""
    n, L = p.n, p.n_limbs
    e = rng.integers(-8, 9, n)
    ct = np.zeros((2, L, n), np.uint64)
    for l, q in enumerate(p.moduli):
        a = [int(rng.integers(0, 2**62)) % q for _ in range(n)]
        a_s = po.negacyclic_schoolbook(a, [int(v) % q for v in s], q)
        ct[0, l] = [(-a_s[k] + int(e[k]) + delta * int(msg[k])) % q for k in range(n)]
        ct[1, l] = a
    return ct, [int(e[k]) + delta * int(msg[k]) for k in range(n)]


def phase(p, ct, s):
    """c0 + c1 s + c2 s^2 per limb, CRT-composed to integers mod Q."""
    res = []
    for l, q in enumerate(p.moduli):
        sq = [int(v) % q for v in s]
        acc = [int(v) for v in ct[0, l]]
        spow = sq
        for comp in range(1, ct.shape[0]):
            term = po.negacyclic_schoolbook([int(v) for v in ct[comp, l]], spow, q)
            acc = po.poly_add(acc, term, q)
            spow = po.negacyclic_schoolbook(spow, sq, q)
        res.append(acc)
    return crt_compose(res, p.moduli)


def negacyclic_int(a, b, Q):
    n = len(a)
    out = [0] * n
    for i, ai in enumerate(a):
        for j, bj in enumerate(b):
            k = i + j
            if k < n:
                out[k] += ai * bj
            else:
                out[k - n] -= ai * bj
    return [v % Q for v in out]


def run_semantic(multiply):
    p = small_params()
    rng = np.random.default_rng(2024)
    s = rng.integers(-1, 2, p.n)  # ternary secret
    m1, m2 = rng.integers(0, 1000, p.n), rng.integers(0, 1000, p.n)
    delta = 1 << 40
    ct1, ph1 = encrypt(rng, p, s, m1, delta)
    ct2, ph2 = encrypt(rng, p, s, m2, delta)
    got1, Q = phase(p, ct1, s)
    assert got1 == [v % Q for v in ph1]                      # decryption of a fresh ciphertext is exact
    ct3 = multiply(p, ct1, ct2)                              # [3][L][N]
    got3, _ = phase(p, ct3, s)
    assert got3 == negacyclic_int(ph1, ph2, Q)               # phase(ct1 (x) ct2) == phase(ct1) * phase(ct2)


def test_tensor_product_is_multiplicative_under_decryption_oracle():
    def mul(p, a, b):
        return Oracle.from_params(p).ct_mul(np.ascontiguousarray(a), np.ascontiguousarray(b))[0]
    run_semantic(mul)


@pytest.mark.gpu
def test_tensor_product_is_multiplicative_under_decryption_hip():
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host

    def mul(p, a, b):
        ctx = Context(p, 0)
        ev = Evaluator(ctx)
        c = ev.multiply(Ciphertext(to_device(a[None], ctx.device)), Ciphertext(to_device(b[None], ctx.device)))
        out = to_host(c.data)[0]
        ctx.close()
        return out
    run_semantic(mul)


# ---- N1: relinearisation ------------------------------------------------------------------------------------------
def keygen_relin(rng, p, s):
    """evk[j] = (-(a_j s) + e_j + g_j s^2, a_j) in the NTT domain; g_j s^2 is s^2 in limb j and 0 elsewhere (CRT basis)."""
    n, L = p.n, p.n_limbs
    orc = Oracle.from_params(p)
    evk = np.zeros((L, 2, L, n), np.uint64)
    errs = []
    for j in range(L):
        e = rng.integers(-8, 9, n)
        errs.append([int(v) for v in e])
        for i, q in enumerate(p.moduli):
            sq = [int(v) % q for v in s]
            a = [int(rng.integers(0, 2**62)) % q for _ in range(n)]
            a_s = po.negacyclic_schoolbook(a, sq, q)
            s2 = po.negacyclic_schoolbook(sq, sq, q) if i == j else [0] * n
            evk[j, 0, i] = [(-a_s[k] + int(e[k]) + s2[k]) % q for k in range(n)]
            evk[j, 1, i] = a
    evk_ntt = orc.ntt_fwd(evk.reshape(-1, L, n)).reshape(evk.shape)
    return evk_ntt, errs


def run_relin_semantic(multiply, relin):
    p = small_params()
    rng = np.random.default_rng(77)
    s = rng.integers(-1, 2, p.n)
    m1, m2 = rng.integers(0, 1000, p.n), rng.integers(0, 1000, p.n)
    ct1, _ = encrypt(rng, p, s, m1, 1 << 40)
    ct2, _ = encrypt(rng, p, s, m2, 1 << 40)
    ct3 = multiply(p, ct1, ct2)
    evk, errs = keygen_relin(rng, p, s)
    ct2r = relin(p, ct3, evk)
    assert ct2r.shape == (2, p.n_limbs, p.n)
    ph3, Q = phase(p, ct3, s)
    phr, _ = phase(p, ct2r, s)
    # exact noise introduced by key switching: sum_j [c2]_{q_j} * e_j  (integer negacyclic products)
    noise = [0] * p.n
    for j in range(p.n_limbs):
        dj = [int(v) for v in ct3[2, j]]
        term = negacyclic_int(dj, errs[j], Q)
        noise = [(a + b) % Q for a, b in zip(noise, term)]
    assert phr == [(a + b) % Q for a, b in zip(ph3, noise)]   # phase(relin(ct)) = phase(ct) + sum_j d_j e_j  exactly
    small = max(min(v, Q - v) for v in noise)
    assert small < (1 << 75) < Q >> 64                         # and that noise is tiny next to Q


def _oracle_mul(p, a, b):
    return Oracle.from_params(p).ct_mul(np.ascontiguousarray(a), np.ascontiguousarray(b))[0]


def test_relinearisation_preserves_the_phase_oracle():
    run_relin_semantic(_oracle_mul, lambda p, ct3, evk: Oracle.from_params(p).relinearize(ct3[None], evk)[0])


@pytest.mark.gpu
def test_relinearisation_preserves_the_phase_hip():
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host

    def relin(p, ct3, evk):
        ctx = Context(p, 0)
        ev = Evaluator(ctx)
        out = to_host(ev.relinearize(Ciphertext(to_device(ct3[None], ctx.device)), to_device(evk, ctx.device)).data)[0]
        ctx.close()
        return out
    run_relin_semantic(_oracle_mul, relin)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["config1", "n4096", "n8192", "shoup10"])
def test_relinearize_bit_exact_vs_oracle(name):
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    if name == "config1":
        p = FheParams.config1()
    elif name == "n4096":
        p = FheParams.n4096_l4()
    elif name == "n8192":
        p = FheParams.n8192_l6()
    else:
        n = 1024
        def gp(bits):
            q = (1 << bits) - ((1 << bits) - 1) % (2 * n)
            while not po.is_prime(q):
                q -= 2 * n
            return q
        qs = (gp(59), gp(50), gp(33))
        p = FheParams(10, qs, tuple(po.min_primitive_2n_root(n, q) for q in qs))
    orc = Oracle.from_params(p)
    L, n, batch = p.n_limbs, p.n, 3
    ct3 = orc.fill(batch * 3, 91).reshape(batch, 3, L, n)
    ct3[0, 2] = (np.array(p.moduli, np.uint64) - np.uint64(1))[:, None]   # worst-case digits
    evk = orc.fill(L * 2, 92).reshape(L, 2, L, n)
    evk[0, 0] = (np.array(p.moduli, np.uint64) - np.uint64(1))[:, None]
    want = orc.relinearize(ct3, evk, threads=0)
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    got = to_host(ev.relinearize(Ciphertext(to_device(ct3, ctx.device)), to_device(evk, ctx.device)).data)
    ctx.close()
    assert np.array_equal(got, want)


# ---- N3: Galois automorphism + key switch ---------------------------------------------------------------------------
def galois_int(a, g, mod=None):
    """a(X) -> a(X^g) on an integer coefficient list (signed permutation); optional reduction mod `mod`."""
    n = len(a)
    out = [0] * n
    for i, v in enumerate(a):
        idx = (i * g) % (2 * n)
        if idx < n:
            out[idx] = v
        else:
            out[idx - n] = -v
    return [x % mod for x in out] if mod else out


def keygen_galois(rng, p, s, g):
    """key_j = (-(a_j s) + e_j + g_j sigma_g(s), a_j), NTT domain."""
    n, L = p.n, p.n_limbs
    orc = Oracle.from_params(p)
    key = np.zeros((L, 2, L, n), np.uint64)
    errs = []
    sg = galois_int([int(v) for v in s], g)
    for j in range(L):
        e = rng.integers(-8, 9, n)
        errs.append([int(v) for v in e])
        for i, q in enumerate(p.moduli):
            sq = [int(v) % q for v in s]
            a = [int(rng.integers(0, 2**62)) % q for _ in range(n)]
            a_s = po.negacyclic_schoolbook(a, sq, q)
            extra = [v % q for v in sg] if i == j else [0] * n
            key[j, 0, i] = [(-a_s[k] + int(e[k]) + extra[k]) % q for k in range(n)]
            key[j, 1, i] = a
    return orc.ntt_fwd(key.reshape(-1, L, n)).reshape(key.shape), errs


def run_galois_semantic(apply):
    p = small_params()
    rng = np.random.default_rng(123)
    s = rng.integers(-1, 2, p.n)
    m = rng.integers(0, 1000, p.n)
    delta = 1 << 40
    ct, ph = encrypt(rng, p, s, m, delta)
    for g in (3, 5, 2 * p.n - 1, 25):
        key, errs = keygen_galois(rng, p, s, g)
        out = apply(p, ct, g, key)
        got, Q = phase(p, out, s)
        # rotated ciphertext (sigma(c0), sigma(c1)); key-switch noise = sum_j [sigma(c1)]_{q_j} * e_j
        sc1 = [galois_int([int(v) for v in ct[1, l]], g, q) for l, q in enumerate(p.moduli)]
        noise = [0] * p.n
        for j in range(p.n_limbs):
            term = negacyclic_int(sc1[j], errs[j], Q)
            noise = [(a + b) % Q for a, b in zip(noise, term)]
        want = [(a + b) % Q for a, b in zip(galois_int(ph, g, Q), noise)]
        assert got == want, f"galois element {g}"


def test_galois_automorphism_and_key_switch_oracle():
    def apply(p, ct, g, key):
        orc = Oracle.from_params(p)
        return orc.switch_key(orc.apply_galois(ct, g)[None], key)[0]
    run_galois_semantic(apply)


@pytest.mark.gpu
def test_galois_automorphism_and_key_switch_hip():
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host

    def apply(p, ct, g, key):
        ctx = Context(p, 0)
        ev = Evaluator(ctx)
        out = to_host(ev.apply_galois(Ciphertext(to_device(ct[None], ctx.device)), g, to_device(key, ctx.device)).data)[0]
        ctx.close()
        return out
    run_galois_semantic(apply)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["config1", "n4096", "n8192"])
def test_apply_galois_and_switch_key_bit_exact_vs_oracle(name):
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    p = {"config1": FheParams.config1, "n4096": FheParams.n4096_l4, "n8192": FheParams.n8192_l6}[name]()
    orc = Oracle.from_params(p)
    L, n, batch = p.n_limbs, p.n, 3
    ct = orc.fill(batch * 2, 301).reshape(batch, 2, L, n)
    ct[0, :, :, :5] = 0
    key = orc.fill(L * 2, 302).reshape(L, 2, L, n)
    ctx = Context(p, 0)
    ev = Evaluator(ctx)
    for g in (3, 2 * n - 1, 5 ** 7 % (2 * n)):
        rot = orc.apply_galois(ct, g)
        got_rot = to_host(ev.apply_galois_words(to_device(ct, ctx.device), g))
        assert np.array_equal(got_rot, rot), g
        want = orc.switch_key(rot, key, threads=0)
        got = to_host(ev.apply_galois(Ciphertext(to_device(ct, ctx.device)), g, to_device(key, ctx.device)).data)
        assert np.array_equal(got, want), g
    ctx.close()


# ---- N1 second half: rescale ---------------------------------------------------------------------------------------------
def run_rescale_semantic(rescale):
    p = small_params()
    rng = np.random.default_rng(9)
    s = rng.integers(-1, 2, p.n)
    m = rng.integers(0, 1000, p.n)
    ct, ph = encrypt(rng, p, s, m, 1 << 90)                 # scale well above q_last ~ 2^60
    out = rescale(p, ct)                                    # [2][L-1][N]
    p2 = p.drop_last_limb()
    got, Q2 = phase(p2, out, s)
    ql = p.moduli[-1]
    centre = lambda v, Q: v - Q if v > Q // 2 else v
    worst = 0
    for k in range(p.n):
        want = ph[k] / ql                                   # exact phase is an integer well inside Q
        err = abs(centre(got[k], Q2) - (ph[k] // ql))
        worst = max(worst, err)
    assert worst <= p.n + 2                                 # rounding error (1 + |s|_1) / 2 at most


def test_rescale_divides_the_phase_oracle():
    run_rescale_semantic(lambda p, ct: Oracle.from_params(p).rescale(ct))


@pytest.mark.gpu
def test_rescale_divides_the_phase_hip():
    from deeppowers_amd.evaluator import Context, Evaluator, to_device, to_host

    def rescale(p, ct):
        ctx = Context(p, 0)
        out = to_host(Evaluator(ctx).rescale_words(to_device(ct, ctx.device)))
        ctx.close()
        return out
    run_rescale_semantic(rescale)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["n4096", "n8192", "mixed"])
def test_rescale_bit_exact_vs_oracle(name):
    from deeppowers_amd.evaluator import Context, Evaluator, to_device, to_host
    if name == "mixed":   # generic (Shoup/Barrett) path: 59/50/33-bit primes
        n = 1024
        def gp(bits):
            q = (1 << bits) - ((1 << bits) - 1) % (2 * n)
            while not po.is_prime(q):
                q -= 2 * n
            return q
        qs = (gp(59), gp(50), gp(33))
        p = FheParams(10, qs, tuple(po.min_primitive_2n_root(n, q) for q in qs))
    else:
        p = FheParams.n4096_l4() if name == "n4096" else FheParams.n8192_l6()
    orc = Oracle.from_params(p)
    x = orc.fill(6, 401).reshape(3, 2, p.n_limbs, p.n)
    x[0, 0, :, :8] = 0
    x[0, 1, :, :8] = (np.array(p.moduli, np.uint64) - np.uint64(1))[:, None]
    ctx = Context(p, 0)
    got = to_host(Evaluator(ctx).rescale_words(to_device(x, ctx.device)))
    ctx.close()
    assert got.shape == (3, 2, p.n_limbs - 1, p.n) and np.array_equal(got, orc.rescale(x))


# ---- N1 hybrid key switching (one special prime) -------------------------------------------------------------------------
def ext_params(p, special_idx=5):
    n = p.n
    P = PRIMES_60[special_idx][0]
    return FheParams(p.log2_n, p.moduli + (P,), p.psi + (pow(PRIMES_60[special_idx][2], 8192 // n, P),))


def keygen_hybrid(rng, p, pe, s, target):
    """key_j = (-(a_j s) + e_j + P g_j target, a_j) over the limbs of the extended parameter set pe; NTT domain."""
    n, Ld, L = p.n, p.n_limbs, pe.n_limbs
    P = pe.moduli[-1]
    orc = Oracle.from_params(pe)
    key = np.zeros((Ld, 2, L, n), np.uint64)
    for j in range(Ld):
        e = rng.integers(-8, 9, n)
        for i, q in enumerate(pe.moduli):
            sq = [int(v) % q for v in s]
            a = [int(rng.integers(0, 2**62)) % q for _ in range(n)]
            a_s = po.negacyclic_schoolbook(a, sq, q)
            extra = [(P % q) * (int(v) % q) % q for v in target] if i == j else [0] * n
            key[j, 0, i] = [(-a_s[k] + int(e[k]) + extra[k]) % q for k in range(n)]
            key[j, 1, i] = a
    return orc.ntt_fwd(key.reshape(-1, L, n)).reshape(key.shape)


def run_hybrid_semantic(keyswitch):
    p = small_params()
    pe = ext_params(p)
    rng = np.random.default_rng(31)
    s = rng.integers(-1, 2, p.n)
    m1, m2 = rng.integers(0, 1000, p.n), rng.integers(0, 1000, p.n)
    ct1, _ = encrypt(rng, p, s, m1, 1 << 30)
    ct2, _ = encrypt(rng, p, s, m2, 1 << 30)
    ct3 = _oracle_mul(p, ct1, ct2)
    s2 = negacyclic_int([int(v) for v in s], [int(v) for v in s], 1 << 200)
    s2 = [v - (1 << 200) if v > (1 << 199) else v for v in s2]
    key = keygen_hybrid(rng, p, pe, s, s2)
    out = keyswitch(pe, ct3, key)
    ph3, Q = phase(p, ct3, s)
    phr, _ = phase(p, out, s)
    centre = lambda v: v - Q if v > Q // 2 else v
    worst = max(abs(centre((a - b) % Q)) for a, b in zip(phr, ph3))
    assert worst < (1 << 24), worst          # ~ Ld N q sigma / P plus rounding, instead of ~2^77 without the special prime
    # rotation: key switch from sigma_g(s) to s with the same machinery
    g = 5
    key_g = keygen_hybrid(rng, p, pe, s, galois_int([int(v) for v in s], g))
    rot = Oracle.from_params(p).apply_galois(ct1, g)
    out_g = keyswitch(pe, rot, key_g)
    ph1, _ = phase(p, ct1, s)
    want = galois_int(ph1, g, Q)
    got, _ = phase(p, out_g, s)
    assert max(abs(centre((a - b) % Q)) for a, b in zip(got, want)) < (1 << 24)


def test_hybrid_key_switching_oracle():
    run_hybrid_semantic(lambda pe, ct, key: Oracle.from_params(pe).keyswitch_hybrid(ct[None], key, ct.shape[0])[0])


@pytest.mark.gpu
def test_hybrid_key_switching_hip():
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host

    def ks(pe, ct, key):
        ctx = Context(pe, 0)
        out = to_host(Evaluator(ctx).keyswitch_hybrid(Ciphertext(to_device(ct[None], ctx.device)), to_device(key, ctx.device)).data)[0]
        ctx.close()
        return out
    run_hybrid_semantic(ks)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["n4096", "n8192", "mixed"])
def test_hybrid_key_switching_bit_exact_vs_oracle(name):
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    if name == "mixed":
        n = 1024
        def gp(bits):
            q = (1 << bits) - ((1 << bits) - 1) % (2 * n)
            while not po.is_prime(q):
                q -= 2 * n
            return q
        qs = (gp(59), gp(50), gp(33), gp(58))
        pe = FheParams(10, qs, tuple(po.min_primitive_2n_root(n, q) for q in qs))
    elif name == "n4096":
        pe = FheParams(12, tuple(x[0] for x in PRIMES_60[:5]), tuple(x[1] for x in PRIMES_60[:5]))   # 4 data limbs + P
    else:
        pe = FheParams.n8192_l6()                                                                      # 5 data limbs + P
    orc = Oracle.from_params(pe)
    L, Ld, n, batch = pe.n_limbs, pe.n_limbs - 1, pe.n, 3
    data = Oracle(pe.log2_n, pe.moduli[:-1], pe.psi[:-1])
    key = orc.fill(Ld * 2, 502).reshape(Ld, 2, L, n)
    ctx = Context(pe, 0)
    ev = Evaluator(ctx)
    for comps in (3, 2):
        ct = data.fill(batch * comps, 501 + comps).reshape(batch, comps, Ld, n)
        ct[0, comps - 1] = (np.array(pe.moduli[:-1], np.uint64) - np.uint64(1))[:, None]
        want = orc.keyswitch_hybrid(ct, key, comps, threads=0)
        got = to_host(ev.keyswitch_hybrid(Ciphertext(to_device(ct, ctx.device)), to_device(key, ctx.device)).data)
        assert np.array_equal(got, want), comps
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["n4096", "n8192", "fold14"])
def test_gpu_batched_rotations_bit_exact(name):
    """dpfhe_rotate_hybrid_batch == per-item automorphism + hybrid key switch of the oracle, for one shared input and for
    one input per item (70 items: more than one 64-element launch group).  fold14: N = 16384, where the key switch is composed from the
    batched transforms with one key per item (round 5)."""
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    if name == "n4096":
        pe = FheParams(12, tuple(x[0] for x in PRIMES_60[:3]), tuple(x[1] for x in PRIMES_60[:3]))     # 2 data limbs + P
    elif name == "fold14":
        from oracle import pyoracle as po
        qs = tuple(PRIMES_60[i][0] for i in (1, 2, 4))
        pe = FheParams(14, qs, tuple(po.min_primitive_2n_root(16384, q) for q in qs))
    else:
        pe = FheParams(13, tuple(x[0] for x in PRIMES_60[:3]), tuple(x[2] for x in PRIMES_60[:3]))
    orc = Oracle.from_params(pe)
    L, Ld, n = pe.n_limbs, pe.n_limbs - 1, pe.n
    data = Oracle(pe.log2_n, pe.moduli[:-1], pe.psi[:-1])
    ctx = Context(pe, 0)
    ev = Evaluator(ctx)
    for k, shared in ((5, True), (70, False)):
        elts = [pow(3, i + 1, 2 * n) for i in range(k)]
        elts[-1] = 2 * n - 1
        keys = orc.fill(k * Ld * 2, 601).reshape(k, Ld, 2, L, n)
        ct = data.fill((1 if shared else k) * 2, 602).reshape(-1, 2, Ld, n)
        got = to_host(ev.rotate_hybrid_batch(Ciphertext(to_device(ct, ctx.device)), elts, to_device(keys, ctx.device)).data)
        for i in (range(k) if k <= 8 else (0, 1, 63, 64, 69)):
            src = ct[0 if shared else i][None]
            rot = data.apply_galois(src, elts[i])
            want = orc.keyswitch_hybrid(rot, keys[i], 2, threads=0)[0]
            assert np.array_equal(got[i], want), (k, i)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["n4096", "n8192", "fold14"])
def test_gpu_hoisted_rotations_bit_exact(name):
    """dpfhe_rotate_hybrid_hoisted == the oracle's hoisted restatement (lift the digits, THEN rotate), for 5 and for 70 rotations of
    one ciphertext (more than one 64-element launch group); and it differs from the non-hoisted path only by a valid
    re-decomposition: both are checked against their own oracle."""
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    if name == "n4096":
        pe = FheParams(12, tuple(x[0] for x in PRIMES_60[:3]), tuple(x[1] for x in PRIMES_60[:3]))     # 2 data limbs + P
    elif name == "fold14":   # N = 16384 (round 5): composed from dpfhe_rotate_hoisted_qp + inverse transform + division by P
        from oracle import pyoracle as po
        qs = tuple(PRIMES_60[i][0] for i in (1, 2, 4))
        pe = FheParams(14, qs, tuple(po.min_primitive_2n_root(16384, q) for q in qs))
    else:
        pe = FheParams(13, tuple(x[0] for x in PRIMES_60[:4]), tuple(x[2] for x in PRIMES_60[:4]))     # 3 data limbs + P
    orc = Oracle.from_params(pe)
    L, Ld, n = pe.n_limbs, pe.n_limbs - 1, pe.n
    data = Oracle(pe.log2_n, pe.moduli[:-1], pe.psi[:-1])
    ctx = Context(pe, 0)
    ev = Evaluator(ctx)
    for k in (5, 70):
        elts = [pow(3, i + 1, 2 * n) for i in range(k)]
        elts[-1] = 2 * n - 1
        keys = orc.fill(k * Ld * 2, 701).reshape(k, Ld, 2, L, n)
        ct = data.fill(2, 702).reshape(1, 2, Ld, n)
        got = to_host(ev.rotate_hybrid_hoisted(Ciphertext(to_device(ct, ctx.device)), elts, to_device(keys, ctx.device)).data)
        idx = list(range(k)) if k <= 8 else [0, 1, 63, 64, 69]
        want = orc.rotate_hoisted(ct[0], [elts[i] for i in idx], keys[idx], threads=0)
        for w, i in zip(want, idx):
            assert np.array_equal(got[i], w), (k, i)
    # several inputs sharing the rotations (tokens): rotation-major output, item r * T + t = rotation r of input t; and the grouped
    # (giant-step) form against the per-item oracle of the un-hoisted path
    T, k = 3, 70
    elts = [pow(3, i + 1, 2 * n) for i in range(k)]
    keys = orc.fill(k * Ld * 2, 703).reshape(k, Ld, 2, L, n)
    cts = data.fill(T * 2, 704).reshape(T, 2, Ld, n)
    dk = to_device(keys, ctx.device)
    got = to_host(ev.rotate_hybrid_hoisted(Ciphertext(to_device(cts, ctx.device)), elts, dk).data).reshape(k, T, 2, Ld, n)
    for r in (0, 21, 22, 63, 64, 69):
        for t in range(T):
            assert np.array_equal(got[r, t], orc.rotate_hoisted(cts[t], [elts[r]], keys[r][None], threads=0)[0]), (r, t)
    if name == "fold14":   # the same call in slices of 3 rotations (23 slices + 1 rotation): inputs and digits are prepared by the first slice only
        ctx.set_scratch_limit(12)
        again = to_host(ev.rotate_hybrid_hoisted(Ciphertext(to_device(cts, ctx.device)), elts, dk).data).reshape(k, T, 2, Ld, n)
        assert np.array_equal(again, got), "sliced hoisted rotations differ from the one-slice run"
        ctx.set_scratch_limit(1024)
    items = data.fill(k * T * 2, 705).reshape(k * T, 2, Ld, n)
    got = to_host(ev.rotate_hybrid_grouped(Ciphertext(to_device(items, ctx.device)), elts, T, dk).data)
    for i in (0, 1, 2, 3, 63, 64, 65, 191, 192, k * T - 1):
        want = orc.keyswitch_hybrid(data.apply_galois(items[i][None], elts[i // T]), keys[i // T], 2, threads=0)[0]
        assert np.array_equal(got[i], want), i
    ctx.close()


def test_hoisted_rotation_is_a_valid_key_switch_oracle():
    """Semantics of the hoisted form on the CPU (toy scheme, big integers): phase(hoisted rotation of ct) == sigma_g(phase(ct)) up to
    the switching noise (< 2^24), exactly like the non-hoisted form - although the two differ word by word."""
    p = small_params()
    pe = ext_params(p)
    rng = np.random.default_rng(77)
    s = rng.integers(-1, 2, p.n)
    m1 = rng.integers(0, 1000, p.n)
    ct1, _ = encrypt(rng, p, s, m1, 1 << 30)
    g = pow(3, 5, 2 * p.n)
    key_g = keygen_hybrid(rng, p, pe, s, galois_int([int(v) for v in s], g))
    orc_e = Oracle.from_params(pe)
    hoisted = orc_e.rotate_hoisted(np.ascontiguousarray(ct1).reshape(2, p.n_limbs, p.n), [g], np.ascontiguousarray(key_g)[None], threads=1)[0]
    ph1, Q = phase(p, ct1, s)
    want = galois_int(ph1, g, Q)
    got, _ = phase(p, hoisted.reshape(np.asarray(ct1).shape), s)
    centre = lambda v: v - Q if v > Q // 2 else v
    assert max(abs(centre((a - b) % Q)) for a, b in zip(got, want)) < (1 << 24)
    plain = orc_e.keyswitch_hybrid(Oracle.from_params(p).apply_galois(ct1, g), key_g, 2)
    assert not np.array_equal(plain.reshape(hoisted.shape), hoisted)      # a different, equally valid decomposition


def _fold_primes(n, count):
    """`count` primes 2^60 - d (d < 2^24: the fold-reduction kernels), q = 1 mod 2n, largest first."""
    out, q = [], (1 << 60) - ((1 << 60) - 1) % (2 * n)
    while len(out) < count:
        if po.is_prime(q):
            out.append(q)
        q -= 2 * n
    assert (1 << 60) - out[-1] < (1 << 24)
    return tuple(out)


@pytest.mark.gpu
@pytest.mark.parametrize("log2n,limbs", [(10, 5), (10, 6), (11, 7), (12, 8)])
def test_key_switch_with_shared_digit_transforms_bit_exact(log2n, limbs):
    """relin_shared_kernel (N <= 4096, 4..7 digits): every remainder path after the first four digits - none (4), one (5), two (6),
    two + one (7) - for the RNS-digit relinearisation (digits = limbs, when <= 7) and for the hybrid forms (digits = limbs - 1)."""
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    n = 1 << log2n
    qs = _fold_primes(n, limbs)
    pe = FheParams(log2n, qs, tuple(po.min_primitive_2n_root(n, q) for q in qs))
    orc = Oracle.from_params(pe)
    ctx = Context(pe, 0)
    assert ctx.uses_fold
    ev = Evaluator(ctx)
    L, Ld, batch = limbs, limbs - 1, 3
    if L <= 7:
        ct3 = orc.fill(batch * 3, 191).reshape(batch, 3, L, n)
        ct3[0, 2] = (np.array(pe.moduli, np.uint64) - np.uint64(1))[:, None]
        evk = orc.fill(L * 2, 192).reshape(L, 2, L, n)
        evk[0, 0] = (np.array(pe.moduli, np.uint64) - np.uint64(1))[:, None]
        got = to_host(ev.relinearize(Ciphertext(to_device(ct3, ctx.device)), to_device(evk, ctx.device)).data)
        assert np.array_equal(got, orc.relinearize(ct3, evk, threads=0))
    data = Oracle(pe.log2_n, pe.moduli[:-1], pe.psi[:-1])
    key = orc.fill(Ld * 2, 193).reshape(Ld, 2, L, n)
    for comps in (3, 2):
        ct = data.fill(batch * comps, 194 + comps).reshape(batch, comps, Ld, n)
        ct[0, comps - 1] = (np.array(pe.moduli[:-1], np.uint64) - np.uint64(1))[:, None]
        want = orc.keyswitch_hybrid(ct, key, comps, threads=0)
        got = to_host(ev.keyswitch_hybrid(Ciphertext(to_device(ct, ctx.device)), to_device(key, ctx.device)).data)
        assert np.array_equal(got, want), comps
    ctx.close()


# ---- round 3: the deferred division by P ("double hoisting") - the new oracle restatements pinned on the CPU ----------------------
def test_qp_restatements_agree_with_the_divided_forms_oracle():
    """orc_rotate_hoisted_qp / orc_switch_key_qp are the hoisted rotation and the hybrid key switch BEFORE the division by P, in the NTT
    domain over Q P.  Dividing them must give the restatements validated above, word for word:
      round(INTT(block r) / P) == orc_rotate_hoisted,   round(INTT(block 0) / P) == the ciphertext itself,
      c0 + round(INTT(switch_key_qp) / P) == orc_keyswitch_hybrid   (round((t + P c0) / P) = c0 + round(t / P) exactly)."""
    for pe in (FheParams(12, tuple(x[0] for x in PRIMES_60[:3]), tuple(x[1] for x in PRIMES_60[:3])),
               ext_params(small_params(8, 3))):
        orc = Oracle.from_params(pe)
        L, Ld, n = pe.n_limbs, pe.n_limbs - 1, pe.n
        data = Oracle(pe.log2_n, pe.moduli[:-1], pe.psi[:-1])
        k = 3
        elts = [pow(3, i + 1, 2 * n) for i in range(k)]
        elts[-1] = 2 * n - 1
        keys = orc.fill(k * Ld * 2, 701).reshape(k, Ld, 2, L, n)
        ct = data.fill(2, 702).reshape(2, Ld, n)
        ct[1, 0] = np.uint64(pe.moduli[0] - 1)                       # worst-case digit
        qp = orc.rotate_hoisted_qp(ct, elts, keys, threads=0)
        assert np.array_equal(orc.rescale(orc.ntt_inv(qp[1:])), orc.rotate_hoisted(ct, elts, keys, threads=0))
        assert np.array_equal(orc.rescale(orc.ntt_inv(qp[0:1]))[0], ct)
        ks = orc.rescale(orc.ntt_inv(orc.switch_key_qp(ct[None], keys[0], threads=0)))[0]
        want = orc.keyswitch_hybrid(ct[None], keys[0], 2)[0]
        assert np.array_equal(data.dyadic("add", ks[0:1].copy(), ct[0:1].copy())[0], want[0]) and np.array_equal(ks[1], want[1])


def test_deferred_giant_step_sum_is_a_valid_key_switched_sum_oracle():
    """Semantics on the toy scheme (big integers): the sum of rotated ciphertexts with ONE division by P for all key-switching terms
        rot_0 + sum_i (c0_i, 0) + round(INTT(sum_i switch_key_qp(sigma_i(ct_i))) / P)
    has the phase  sum_i sigma_i(phase(ct_i))  up to the switching noise - what dpfhe_switch_key_qp + dpfhe_rescale_bsgs compute."""
    p = small_params()
    pe = ext_params(p)
    rng = np.random.default_rng(91)
    s = rng.integers(-1, 2, p.n)
    orc_e, orc_d = Oracle.from_params(pe), Oracle.from_params(p)
    L, Ld, n = pe.n_limbs, p.n_limbs, p.n
    gs = [1, pow(3, 3, 2 * n), pow(3, 17, 2 * n), 2 * n - 1]
    cts, phases = [], []
    for i in range(len(gs)):
        ct, _ = encrypt(rng, p, s, rng.integers(0, 1000, n), 1 << 30)
        cts.append(ct)
        phases.append(phase(p, ct, s)[0])
    Q = phase(p, cts[0], s)[1]
    want = [0] * n
    for ph, g in zip(phases, gs):
        want = [(a + b) % Q for a, b in zip(want, galois_int(ph, g, Q))]
    acc = np.zeros((1, 2, L, n), np.uint64)
    total = cts[0].copy()                                              # the un-rotated term keeps both components
    for ct, g in zip(cts[1:], gs[1:]):
        rot = orc_d.apply_galois(ct, g)
        key_g = keygen_hybrid(rng, p, pe, s, galois_int([int(v) for v in s], g))
        acc = orc_e.dyadic("add", acc, orc_e.switch_key_qp(rot[None], np.ascontiguousarray(key_g), threads=1))
        total[0] = orc_d.dyadic("add", total[0][None].copy(), np.ascontiguousarray(rot[0])[None])[0]      # c0 parts
    total = orc_d.dyadic("add", np.ascontiguousarray(total), orc_e.rescale(orc_e.ntt_inv(acc))[0])
    got, _ = phase(p, total.reshape(np.asarray(cts[0]).shape), s)
    centre = lambda v: v - Q if v > Q // 2 else v
    assert max(abs(centre((a - b) % Q)) for a, b in zip(got, want)) < (1 << 26)
