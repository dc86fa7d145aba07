"""Big-int CPU oracle for the RLWE ciphertext-arithmetic hot path (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED BY THE REFERENCE: deeppowers/deeppowers contains no FHE code (no NTT, no
modular arithmetic, no Ciphertext/Evaluator - SURVEY.md section 0; FHE appears only as prose in
/root/reference/README.md:42,93,197), so there is no reference function, golden vector or test to
restate.  This oracle follows the mathematical definition instead - arithmetic in
R_q = Z_q[X]/(X^N + 1) with canonical residues in [0, q) - which has a unique answer, and is
itself pinned against the hand-checkable / independently computed known answers recorded in
SURVEY.md Appendix B (tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Everything here is plain Python ints (arbitrary precision), so no overflow reasoning is needed;
it is slow by design and used only at small sizes and to generate tests/golden/ fixtures.

Conventions pinned by SURVEY.md section 8(a):
  * forward NTT  (A1): natural-order in, bit-reversed-order out:
        ahat[k] = sum_j a[j] * psi^((2*brv(k)+1) * j)  mod q
  * inverse NTT  (A2): bit-reversed in, natural out, includes the N^-1 scaling
  * ct x ct multiply (A6): tensor product WITHOUT relinearisation:
        (a0,a1) (x) (b0,b1) = (a0*b0, a0*b1 + a1*b0, a1*b1)   per RNS limb
  * data layout [batch][component][limb][N], words are canonical residues
"""
from __future__ import annotations

MASK64 = (1 << 64) - 1


# ----------------------------------------------------------------------------------------------
# deterministic synthetic data (splitmix64)
# ----------------------------------------------------------------------------------------------
class SplitMix64:
    """splitmix64 exactly as written down in SURVEY.md Appendix B."""

    def __init__(self, seed: int):
        self.state = seed & MASK64

    def next(self) -> int:
        self.state = (self.state + 0x9E3779B97F4A7C15) & MASK64
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def words_mod(self, n: int, q: int) -> list[int]:
        return [self.next() % q for _ in range(n)]


# ----------------------------------------------------------------------------------------------
# number theory helpers
# ----------------------------------------------------------------------------------------------
def bit_reverse(x: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def is_prime(n: int) -> bool:
    """Deterministic Miller-Rabin for n < 3.3e24 (first 13 primes as bases)."""
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41)
    for p in small:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in small:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def is_primitive_2n_root(psi: int, n: int, q: int) -> bool:
    """psi has order exactly 2n (n a power of two)  <=>  psi^n == -1 (mod q)."""
    return 0 < psi < q and pow(psi, n, q) == q - 1


def min_primitive_2n_root(n: int, q: int) -> int:
    """Smallest primitive 2n-th root of unity mod prime q (the rule of SURVEY.md Appendix A)."""
    assert (q - 1) % (2 * n) == 0
    e = (q - 1) // (2 * n)
    # find one generator of the order-2n subgroup, then enumerate its odd powers
    g = None
    x = 2
    while g is None:
        c = pow(x, e, q)
        if pow(c, n, q) == q - 1:
            g = c
        x += 1
    best = g
    g2 = g * g % q
    cur = g
    for _ in range(n):  # odd powers g^1, g^3, ... are exactly the primitive 2n-th roots
        if cur < best:
            best = cur
        cur = cur * g2 % q
    return best


# ----------------------------------------------------------------------------------------------
# ring arithmetic: ground truth
# ----------------------------------------------------------------------------------------------
def negacyclic_schoolbook(a: list[int], b: list[int], q: int) -> list[int]:
    """c = a*b in Z_q[X]/(X^N+1), O(N^2), canonical residues. THE ground truth."""
    n = len(a)
    assert len(b) == n
    acc = [0] * n
    for i, ai in enumerate(a):
        if ai == 0:
            continue
        for j, bj in enumerate(b):
            k = i + j
            if k < n:
                acc[k] += ai * bj
            else:
                acc[k - n] -= ai * bj
    return [x % q for x in acc]


def ntt_forward_definition(a: list[int], q: int, psi: int) -> list[int]:
    """Direct O(N^2) evaluation of the pinned forward transform (A1)."""
    n = len(a)
    bits = n.bit_length() - 1
    out = []
    for k in range(n):
        w = pow(psi, 2 * bit_reverse(k, bits) + 1, q)
        acc, wj = 0, 1
        for j in range(n):
            acc += a[j] * wj
            wj = wj * w % q
        out.append(acc % q)
    return out


def root_powers_bitrev(n: int, q: int, psi: int) -> list[int]:
    """rp[i] = psi^brv(i) for i in [0, n) - the table a Cooley-Tukey NTT walks as rp[m + i]."""
    bits = n.bit_length() - 1
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * psi % q
    return [pw[bit_reverse(i, bits)] for i in range(n)]


def ntt_forward(a: list[int], q: int, psi: int) -> list[int]:
    """In-place Cooley-Tukey negacyclic NTT, natural in -> bit-reversed out (fast, exact)."""
    n = len(a)
    a = list(a)
    rp = root_powers_bitrev(n, q, psi)
    t, m = n, 1
    while m < n:
        t >>= 1
        for i in range(m):
            w = rp[m + i]
            j1 = 2 * i * t
            for j in range(j1, j1 + t):
                u, v = a[j], a[j + t] * w % q
                a[j], a[j + t] = (u + v) % q, (u - v) % q
        m <<= 1
    return a


def ntt_inverse(a: list[int], q: int, psi: int) -> list[int]:
    """Gentleman-Sande inverse of ntt_forward incl. the N^-1 scaling; bit-reversed in -> natural out."""
    n = len(a)
    a = list(a)
    rp = root_powers_bitrev(n, q, psi)
    irp = [pow(x, q - 2, q) for x in rp]
    t, m = 1, n
    while m > 1:
        h = m >> 1
        j1 = 0
        for i in range(h):
            w = irp[h + i]
            for j in range(j1, j1 + t):
                u, v = a[j], a[j + t]
                a[j], a[j + t] = (u + v) % q, (u - v) * w % q
            j1 += 2 * t
        t <<= 1
        m = h
    ninv = pow(n, q - 2, q)
    return [x * ninv % q for x in a]


def dyadic_mul(a, b, q):
    return [x * y % q for x, y in zip(a, b)]


def dyadic_mul_add(acc, a, b, q):
    return [(c + x * y) % q for c, x, y in zip(acc, a, b)]


def poly_add(a, b, q):
    return [(x + y) % q for x, y in zip(a, b)]


def poly_sub(a, b, q):
    return [(x - y) % q for x, y in zip(a, b)]


def poly_negate(a, q):
    return [(-x) % q for x in a]


# ----------------------------------------------------------------------------------------------
# ciphertext-level operations (A6, A7, A8).  A ciphertext is ct[component][limb] -> list of N ints
# ----------------------------------------------------------------------------------------------
def ct_mul_schoolbook(a, b, moduli):
    """(a0,a1) (x) (b0,b1) -> (a0b0, a0b1+a1b0, a1b1), per limb, coefficient domain in/out."""
    out = [[None] * len(moduli) for _ in range(3)]
    for l, q in enumerate(moduli):
        a0, a1, b0, b1 = a[0][l], a[1][l], b[0][l], b[1][l]
        out[0][l] = negacyclic_schoolbook(a0, b0, q)
        out[1][l] = poly_add(negacyclic_schoolbook(a0, b1, q), negacyclic_schoolbook(a1, b0, q), q)
        out[2][l] = negacyclic_schoolbook(a1, b1, q)
    return out


def ct_mul_ntt(a, b, moduli, psis):
    """Same result through the NTT path (4 fwd NTT + 4 dyadic + 1 add + 3 inv NTT per limb)."""
    out = [[None] * len(moduli) for _ in range(3)]
    for l, (q, psi) in enumerate(zip(moduli, psis)):
        A0, A1, B0, B1 = (ntt_forward(x, q, psi) for x in (a[0][l], a[1][l], b[0][l], b[1][l]))
        out[0][l] = ntt_inverse(dyadic_mul(A0, B0, q), q, psi)
        out[1][l] = ntt_inverse(dyadic_mul_add(dyadic_mul(A0, B1, q), A1, B0, q), q, psi)
        out[2][l] = ntt_inverse(dyadic_mul(A1, B1, q), q, psi)
    return out


def matvec_plain(W, x, moduli):
    """y_i = sum_j W[i][j] (.) x[j]  (all NTT domain, dyadic).  W[i][j][limb], x[j][comp][limb]."""
    rows, cols = len(W), len(x)
    ncomp = len(x[0])
    y = []
    for i in range(rows):
        yi = [[None] * len(moduli) for _ in range(ncomp)]
        for c in range(ncomp):
            for l, q in enumerate(moduli):
                n = len(x[0][c][l])
                acc = [0] * n
                for j in range(cols):
                    w, v = W[i][j][l], x[j][c][l]
                    for k in range(n):
                        acc[k] += w[k] * v[k]
                yi[c][l] = [z % q for z in acc]
        y.append(yi)
    return y


# ----------------------------------------------------------------------------------------------
# flat-layout helpers: [batch][component][limb][N] little-endian u64 words
# ----------------------------------------------------------------------------------------------
def flatten_ct(ct) -> list[int]:
    return [w for comp in ct for limb in comp for w in limb]


def unflatten_ct(words, ncomp, nlimbs, n):
    it = iter(words)
    return [[[next(it) for _ in range(n)] for _ in range(nlimbs)] for _ in range(ncomp)]


def words_to_bytes(words) -> bytes:
    return b"".join(int(w).to_bytes(8, "little") for w in words)


# ---- round 4: exact base extension / scale-and-round, DEFINITION form (big integers) - pins oracle.c orc_base_extend ------------------
def crt_centered(residues, moduli):
    """the integer in (-Q/2, Q/2] with the given residues (Q = prod moduli, pairwise coprime)"""
    Q = 1
    for m in moduli:
        Q *= m
    x = 0
    for r, m in zip(residues, moduli):
        Qi = Q // m
        x += int(r) * Qi * pow(Qi, -1, m)
    x %= Q
    return x - Q if x > Q // 2 else x


def base_extend(words, src_moduli, dst_moduli):
    """words[i][k]: residue k of limb i -> out[j][k] = X_k mod dst_moduli[j], X_k the centred CRT lift"""
    n = len(words[0])
    out = [[0] * n for _ in dst_moduli]
    for k in range(n):
        x = crt_centered([w[k] for w in words], src_moduli)
        for j, p in enumerate(dst_moduli):
            out[j][k] = x % p
    return out


def scale_round(words, moduli, drop, keep, multiplier):
    """words[i][k] on ALL limbs (moduli); drop / keep: limb index lists -> out[j][k] = round(multiplier * X / prod(drop moduli)) mod keep limb j,
    rounding to nearest (ties cannot occur: the divisor is odd)"""
    n = len(words[0])
    Qd = 1
    for i in drop:
        Qd *= moduli[i]
    out = [[0] * n for _ in keep]
    for k in range(n):
        x = multiplier * crt_centered([w[k] for w in words], moduli)
        y = (2 * x + Qd) // (2 * Qd)          # floor(x / Qd + 1/2)
        for j, i in enumerate(keep):
            out[j][k] = y % moduli[i]
    return out
