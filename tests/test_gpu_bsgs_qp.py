"""GPU: the baby-step / giant-step stages with the division by P deferred (include/dpfhe.h "N3, round 3") against the oracle's
definition-form restatements - automorphisms in the coefficient domain, transforms afterwards - bit for bit, on FoldArith and on
generic (Shoup) primes, with more than one 64-element launch group and several tokens."""
import numpy as np
import pytest

from deeppowers_amd.params import FheParams, PRIMES_60, ntt_primes
from oracle import pyoracle as po
from oracle.cbind import Oracle

pytestmark = pytest.mark.gpu


def _params(name):
    if name == "mixed":   # generic primes: the Shoup kernels
        n = 1024

        def gp(bits):
            q = (1 << bits) - ((1 << bits) - 1) % (2 * n)
            while not po.is_prime(q):
                q -= 2 * n
            return q
        qs = (gp(59), gp(50), gp(33), gp(58))
        return FheParams(10, qs, tuple(po.min_primitive_2n_root(n, q) for q in qs))
    if name == "n4096":
        return FheParams(12, tuple(x[0] for x in PRIMES_60[:3]), tuple(x[1] for x in PRIMES_60[:3]))     # 2 data limbs + P
    if name == "n4096_l6":
        return FheParams(12, tuple(x[0] for x in PRIMES_60[:6]), tuple(x[1] for x in PRIMES_60[:6]))     # 5 data limbs + P: relin_shared_kernel
    if name == "fold14":      # N = 16384: 2 data limbs + P on the pinned primes that are 1 mod 32768 - the stages composed from the batched transforms (round 5)
        qs = tuple(PRIMES_60[i][0] for i in (1, 2, 4))
        return FheParams(14, qs, tuple(po.min_primitive_2n_root(16384, q) for q in qs))
    if name == "n8192_l10":   # 9 data limbs + P: the digit counts of a deep modulus chain (examples/encrypted_gpt2_stack.cpp) - loop-form baby steps with
        return ntt_primes(13, 10)   # their periodic lazy folds, per-component hoisted rotations, more lazily added products per key switch
    return FheParams(13, tuple(x[0] for x in PRIMES_60[:4]), tuple(x[2] for x in PRIMES_60[:4]))         # 3 data limbs + P


@pytest.mark.parametrize("name", ["mixed", "n4096", "n8192", "n8192_l10", "fold14"])
def test_rotate_hoisted_qp_bit_exact(name):
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    pe = _params(name)
    orc = Oracle.from_params(pe)
    L, Ld, n = pe.n_limbs, pe.n_limbs - 1, pe.n
    data = Oracle(pe.log2_n, pe.moduli[:-1], pe.psi[:-1])
    ctx = Context(pe, 0)
    ev = Evaluator(ctx)
    for k, T in ((5, 1), (70, 3), (0, 2)):
        elts = [pow(3, i + 1, 2 * n) for i in range(k)]
        if k:
            elts[-1] = 2 * n - 1
        keys = orc.fill(max(k, 1) * Ld * 2, 801).reshape(max(k, 1), Ld, 2, L, n)[:k]
        cts = data.fill(T * 2, 802 + k).reshape(T, 2, Ld, n)
        cts[0, 1] = (np.array(pe.moduli[:-1], np.uint64) - np.uint64(1))[:, None]       # worst-case digits
        got = to_host(ev.rotate_hoisted_qp(Ciphertext(to_device(cts, ctx.device)), elts, to_device(keys, ctx.device) if k else None))
        assert got.shape == (k + 1, T, 2, L, n)
        idx = list(range(k)) if k <= 8 else [0, 1, 63, 64, 69]
        for t in range(T):
            want = orc.rotate_hoisted_qp(cts[t], [elts[i] for i in idx], keys[idx] if k else keys, threads=0)
            assert np.array_equal(got[0, t], want[0]), (k, t, "identity block")
            for w, i in zip(want[1:], idx):
                assert np.array_equal(got[1 + i, t], w), (k, t, i)
    ctx.close()


@pytest.mark.parametrize("name", ["mixed", "n4096", "n8192", "fold14"])
def test_ntt_inverse_galois_bit_exact(name):
    """sigma_g applied as a gather in the NTT domain + inverse transform == inverse transform + coefficient-domain automorphism;
    in place and out of place; 70 elements (two launch groups), several RNS polynomials per element."""
    from deeppowers_amd.evaluator import Context, Evaluator, to_device, to_host
    pe = _params(name)
    orc = Oracle.from_params(pe)
    L, n = pe.n_limbs, pe.n
    ctx = Context(pe, 0)
    ev = Evaluator(ctx)
    for k, per in ((1, 1), (70, 3)):
        elts = [pow(3, 5 * i, 2 * n) for i in range(k)]     # includes g = 1 (the identity)
        if k > 2:
            elts[2] = 2 * n - 1
        x = orc.fill(k * per, 811).reshape(k, per, L, n)
        want = np.stack([orc.apply_galois(orc.ntt_inv(x[e]), elts[e]) for e in range(k)])
        d = to_device(x, ctx.device)
        got = to_host(ev.ntt_inverse_galois(d, elts))
        assert np.array_equal(got, want)
        ev.ntt_inverse_galois(d, elts, out=d)               # in place
        assert np.array_equal(to_host(d), want)
    ctx.close()


@pytest.mark.parametrize("name", ["mixed", "n4096", "n4096_l6", "n8192", "n8192_l10", "fold14"])
def test_switch_key_qp_bit_exact(name):
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    pe = _params(name)
    orc = Oracle.from_params(pe)
    L, Ld, n = pe.n_limbs, pe.n_limbs - 1, pe.n
    data = Oracle(pe.log2_n, pe.moduli[:-1], pe.psi[:-1])
    ctx = Context(pe, 0)
    ev = Evaluator(ctx)
    for k, group in ((1, 3), (5, 1), (9, 4)):
        keys = orc.fill(k * Ld * 2, 821).reshape(k, Ld, 2, L, n)
        keys[0, 0, 0] = (np.array(pe.moduli, np.uint64) - np.uint64(1))[:, None]
        items = data.fill(k * group * 2, 822).reshape(k * group, 2, Ld, n)
        items[0, 1] = (np.array(pe.moduli[:-1], np.uint64) - np.uint64(1))[:, None]
        got = to_host(ev.switch_key_qp(Ciphertext(to_device(items, ctx.device)), to_device(keys, ctx.device), group))
        for i in range(k * group):
            assert np.array_equal(got[i], orc.switch_key_qp(items[i][None], keys[i // group], threads=0)[0]), (k, group, i)
    ctx.close()


@pytest.mark.parametrize("name", ["mixed", "n8192", "n8192_l10", "fold14"])
def test_rescale_bsgs_and_the_whole_deferred_sum(name):
    """dpfhe_rescale_bsgs == round(x / P) + addends (oracle composition); and the deferred giant-step sum
         rescale_bsgs(INTT(sum_i switch_key_qp(rot_i)), rot)  decrypts like  rot_0 + sum_i keyswitch_hybrid(rot_i):
    word by word it differs (ONE rounding instead of one per term), by at most (number of terms) in every coefficient."""
    from deeppowers_amd.evaluator import Ciphertext, Context, Evaluator, to_device, to_host
    pe = _params(name)
    orc = Oracle.from_params(pe)
    L, Ld, n = pe.n_limbs, pe.n_limbs - 1, pe.n
    data = Oracle(pe.log2_n, pe.moduli[:-1], pe.psi[:-1])
    ctx = Context(pe, 0)
    ev = Evaluator(ctx)
    n2, T = 6, 3
    rot = data.fill(n2 * T * 2, 831).reshape(n2, T, 2, Ld, n)
    t_qp = orc.fill(T * 2, 832).reshape(T, 2, L, n)
    got = to_host(ev.rescale_bsgs(to_device(t_qp, ctx.device), to_device(rot, ctx.device)))
    want = orc.rescale(t_qp)
    for t in range(T):
        for a in range(n2):
            want[t, 0] = data.dyadic("add", want[t, 0][None].copy(), rot[a, t, 0][None].copy())[0]
        want[t, 1] = data.dyadic("add", want[t, 1][None].copy(), rot[0, t, 1][None].copy())[0]
    assert np.array_equal(got, want)
    # no addends at all
    got0 = to_host(ev.rescale_bsgs(to_device(t_qp, ctx.device), to_device(rot[:0], ctx.device)))
    assert np.array_equal(got0, orc.rescale(t_qp))
    # the whole deferred sum on the GPU
    keys = orc.fill((n2 - 1) * Ld * 2, 833).reshape(n2 - 1, Ld, 2, L, n)
    d_rot = to_device(rot, ctx.device)
    terms = ev.switch_key_qp(Ciphertext(d_rot[1:].reshape((n2 - 1) * T, 2, Ld, n)), to_device(keys, ctx.device), T)      # [(n2-1) T][2][L][N]
    # reduce over the giant steps: [n2-1] items of T*2 components
    import torch
    from deeppowers_amd import _cabi
    summed = torch.empty((T, 2, L, n), dtype=torch.int64, device=ctx.device)
    _cabi.check(ctx._lib.dpfhe_reduce_sum(ctx.handle, summed.data_ptr(), terms.data_ptr(), n2 - 1, T * 2, None), "dpfhe_reduce_sum")
    ev.ntt_inverse_(summed)
    got = to_host(ev.rescale_bsgs(summed, d_rot))
    # oracle, same order of operations (bit exact) ...
    acc = np.zeros((T, 2, L, n), np.uint64)
    for i in range(1, n2):
        acc = orc.dyadic("add", acc, orc.switch_key_qp(rot[i], keys[i - 1], threads=0))
    want = orc.rescale(orc.ntt_inv(acc))
    for t in range(T):
        for a in range(n2):
            want[t, 0] = data.dyadic("add", want[t, 0][None].copy(), rot[a, t, 0][None].copy())[0]
        want[t, 1] = data.dyadic("add", want[t, 1][None].copy(), rot[0, t, 1][None].copy())[0]
    assert np.array_equal(got, want)
    # ... and against the per-term path: the sums differ by the roundings only
    per_term = rot[0].copy()
    for i in range(1, n2):
        per_term = data.dyadic("add", per_term, orc.keyswitch_hybrid(rot[i], keys[i - 1], 2, threads=0))
    q = np.array(pe.moduli[:-1], np.uint64)[None, None, :, None]
    diff = (got.astype(object) - per_term.astype(object)) % q.astype(object)
    diff = np.minimum(diff, q.astype(object) - diff)
    assert int(diff.max()) <= n2
    ctx.close()
