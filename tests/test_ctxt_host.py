"""Host control flow of helib_amd.ctxt (Context chain, prime-set decisions, multiplyBy
orchestration) on the CPU with the oracle as the polynomial backend: chain shapes of SURVEY.md
Appendix B, and decrypt(multiplyBy(enc a, enc b)) == a*b through the full reference sequence
bringToSet -> tensorProduct -> dropSmallAndSpecialPrimes -> reLinearize."""
import math

import numpy as np
import pytest

from helib_amd import ctxt as hc
from oracle import oracle as O
from tests import bgv_ref as B
from oracle.backend import OKeySwitch, OPoly, OracleOps


def test_chain_shapes_match_survey_appendix_B():
    c = hc.ChainContext(32768, 65537, 1, bits=950, c=3)
    assert len(c.ctxtPrimes) == 16 and all(q.bit_length() == 60 for q in (c.primes[i] for i in c.ctxtPrimes))
    assert [len(d) for d in c.digits] == [6, 5, 5]
    assert len(c.specialPrimes) == 6 and all(c.primes[i].bit_length() == 56 for i in c.specialPrimes)
    assert sorted(c.primes[i].bit_length() for i in c.smallPrimes) == [40, 40, 48, 51, 54, 57]
    # moduli order: small, ctxt, special
    assert c.smallPrimes == list(range(6)) and c.ctxtPrimes == list(range(6, 22))
    assert len(c.modSizes.sizes) == (1 << 6) * 17


def test_fresh_multiply_prime_set_decision_m32768():
    """fresh ciphertexts at m=32768 bits=950: the reference drops to a set that trades one 60-bit
    ctxt prime for a small prime (cost 100 = one added prime)."""
    c = hc.ChainContext(32768, 65537, 1, bits=950, c=3)
    a = hc.Ctxt(c, None)
    a.parts = {"1": None, "s": None}
    a.primeSet = frozenset(c.ctxtPrimes)
    a.lnNoise = math.log(c.freshNoiseBound())
    lo, hi = hc.Ctxt.computeIntervalForMul(a, a)
    s = c.modSizes.getSet4Size(lo, hi, a.primeSet, a.primeSet, False)
    assert lo <= c.logOfProduct(s) <= hi
    added, removed = s - a.primeSet, a.primeSet - s
    assert len(added) == 1 and added <= set(c.smallPrimes)
    assert removed == {c.ctxtPrimes[-1]}


def make_keys(ctx, octx, seed=3, auto_k=None):
    """Secret key s and a key-switching matrix W (src/keys.cpp:1159-1255) from s^2 to s, or --
    auto_k given -- from s(X^k) to s:  b_j = P*B_j*s' + p*e_j - s*a_j."""
    rng = np.random.default_rng(seed)
    N = octx.N
    s = rng.integers(-1, 2, size=N)
    allp = ctx.ctxtPrimes + ctx.specialPrimes

    def rows(coeffs, idx):
        coef = np.array([[int(v) % ctx.primes[i] for v in coeffs] for i in idx], dtype=np.uint64)
        return octx.fft(idx, coef)

    def mul(a, b, idx):
        return np.stack([O.row_op("mul", a[r], b[r], ctx.primes[i]) for r, i in enumerate(idx)])

    def sub(a, b, idx):
        return np.stack([O.row_op("sub", a[r], b[r], ctx.primes[i]) for r, i in enumerate(idx)])

    def add(a, b, idx):
        return np.stack([O.row_op("add", a[r], b[r], ctx.primes[i]) for r, i in enumerate(idx)])

    s_all = rows(s, allp)
    if auto_k is None:
        s2 = mul(s_all, s_all, allp)
    else:   # s(X^k): the automorphism permutes evaluation rows
        zms = O.zmstar(octx.m)
        s2 = np.stack([O.automorph(r, octx.m, zms, auto_k) for r in s_all])
        rng = np.random.default_rng(seed + 1000 + auto_k)   # fresh errors, same secret key
        rng.integers(-1, 2, size=N)
    P = ctx.productOfPrimes(ctx.specialPrimes)
    p = ctx.ptxtSpace
    kb, ka, Bj = [], [], 1
    for j, d in enumerate(ctx.digits):
        a = np.stack([O.fill_uniform(N, ctx.primes[i], 900 + j * 100 + i + 7919 * (auto_k or 0)) for i in allp])
        e = np.rint(rng.normal(0, 3.2, size=N)).astype(np.int64)
        pe = rows([p * int(x) for x in e], allp)
        fac = P * Bj
        t = np.stack([O.row_op("mul_scalar", s2[r], fac % ctx.primes[i], ctx.primes[i])
                      for r, i in enumerate(allp)])
        kb.append(sub(add(t, pe, allp), mul(s_all, a, allp), allp))
        ka.append(a)
        Bj *= ctx.productOfPrimes(d)
    return s, allp, np.stack(kb), np.stack(ka), rows


def encrypt(ctx, octx, s, msg, seed, rows):
    """c0 + c1*s = p*e + (Q mod p)*msg on the ctxt primes (src/keys.cpp:454-458)."""
    idx = ctx.ctxtPrimes
    rng = np.random.default_rng(100 + seed)
    N, p = octx.N, ctx.ptxtSpace
    QmodP = ctx.productOfPrimes(idx) % p
    c1 = np.stack([O.fill_uniform(N, ctx.primes[i], 77 * seed + i) for i in idx])
    e = np.rint(rng.normal(0, 3.2, size=N)).astype(np.int64)
    rhs = rows([p * int(x) + QmodP * int(mm) for x, mm in zip(e, msg)], idx)
    s_rows = rows(s, idx)
    c0 = np.stack([O.row_op("sub", rhs[r], O.row_op("mul", c1[r], s_rows[r], ctx.primes[i]), ctx.primes[i])
                   for r, i in enumerate(idx)])
    return c0, c1


def decrypt(ctx, octx, s, ct, rows):
    """SecKey::Decrypt core (src/keys.cpp:1327-1420) for a 2-part ciphertext."""
    idx = sorted(ct.primeSet)
    p0, p1 = ct.parts["1"], ct.parts["s"]
    order = p0.getIndexSet()
    d0, d1 = p0.download()[:, 0], p1.download()[:, 0]
    s_rows = rows(s, order)
    t = np.stack([O.row_op("add", d0[r], O.row_op("mul", d1[r], s_rows[r], ctx.primes[i]), ctx.primes[i])
                  for r, i in enumerate(order)])
    poly = octx.to_poly(order, t)
    p = ct.ptxtSpace
    factor = ctx.productOfPrimes(idx) % p * ct.intFactor % p
    finv = pow(factor, -1, p)
    Q = ctx.productOfPrimes(idx)
    assert max(abs(v) for v in poly) < Q // 4, "noise too large"
    return [(v % p) * finv % p for v in poly]


@pytest.mark.parametrize("m,p,bits", [(128, 257, 150), (64, 65537, 250)])
def test_multiplyBy_full_sequence_decrypts(m, p, bits):
    ctx = hc.ChainContext(m, p, 1, bits=bits, c=3)
    octx = O.Ctx(m)
    for q in ctx.primes:
        octx.add_prime(q)
    s, allp, kb, ka, rows = make_keys(ctx, octx)
    ops = OracleOps(octx)
    W = OKeySwitch(allp, kb, ka)
    rng = np.random.default_rng(9)
    ma, mb = rng.integers(0, p, size=octx.N), rng.integers(0, p, size=octx.N)
    ca = hc.Ctxt.fresh(ctx, ops, *(OPoly(octx, ctx.ctxtPrimes, x) for x in encrypt(ctx, octx, s, ma, 1, rows)), ksw=W)
    cb = hc.Ctxt.fresh(ctx, ops, *(OPoly(octx, ctx.ctxtPrimes, x) for x in encrypt(ctx, octx, s, mb, 2, rows)), ksw=W)
    assert decrypt(ctx, octx, s, ca, rows) == [int(v) for v in ma]
    ca.multiplyBy(cb)
    assert set(ca.parts) == {"1", "s"}
    assert ca.primeSet >= frozenset(ctx.specialPrimes)
    want = [int(v) for v in B.polymul_mod_phi(ma, mb, m, p)]
    assert decrypt(ctx, octx, s, ca, rows) == want
    # the next operation's bringToSet drops the special primes again
    ca.dropSmallAndSpecialPrimes()
    assert not (ca.primeSet & frozenset(ctx.specialPrimes))
    assert decrypt(ctx, octx, s, ca, rows) == want


@pytest.mark.parametrize("m,p,bits", [(128, 257, 150), (64, 65537, 250)])
def test_multiplyBy_leaves_the_tensor_product_to_its_consumer(m, p, bits, monkeypatch):
    """Ctxt.multiplyBy parks the tensor product (as the C++ host's pendingTensor) and hands the operand parts to
    ops.tensorBringToSet (a mod-switch follows) or ops.mulRelin (none does): the calls are seen, tensorProduct is
    not, and the result -- parts, prime set, intFactor, noise estimate -- equals the eager sequence's word for word,
    at level 1 and at level 2 (operands that carry the special primes)."""
    ctx = hc.ChainContext(m, p, 1, bits=bits, c=3)
    octx = O.Ctx(m)
    for q in ctx.primes:
        octx.add_prime(q)
    s, allp, kb, ka, rows = make_keys(ctx, octx)
    W = OKeySwitch(allp, kb, ka)
    rng = np.random.default_rng(19)
    ma, mb = rng.integers(0, p, size=octx.N), rng.integers(0, p, size=octx.N)
    ea, eb = encrypt(ctx, octx, s, ma, 1, rows), encrypt(ctx, octx, s, mb, 2, rows)

    class Spy(OracleOps):
        def __init__(self, o):
            super().__init__(o)
            self.calls = []

        def tensorProduct(self, *a):
            self.calls.append("tensorProduct")
            return super().tensorProduct(*a)

        def tensorBringToSet(self, *a, **k):
            self.calls.append("tensorBringToSet")
            return OracleOps.tensorBringToSet(OracleOps(self.o), *a, **k)

        def mulRelin(self, *a, **k):
            self.calls.append("mulRelin")
            return OracleOps.mulRelin(OracleOps(self.o), *a, **k)

    def run(lazy):
        monkeypatch.setattr(hc.Ctxt, "lazyTensor", lazy)
        ops = Spy(octx)
        ca = hc.Ctxt.fresh(ctx, ops, *(OPoly(octx, ctx.ctxtPrimes, x) for x in ea), ksw=W)
        cb = hc.Ctxt.fresh(ctx, ops, *(OPoly(octx, ctx.ctxtPrimes, x) for x in eb), ksw=W)
        ca.multiplyBy(cb)
        l1 = ca.clone()
        ca.multiplyBy(l1)
        return ops.calls, l1, ca

    calls_lazy, a1, a2 = run(True)
    calls_eager, b1, b2 = run(False)
    assert "tensorProduct" not in calls_lazy and len(calls_lazy) == 2 and set(calls_lazy) <= {"tensorBringToSet", "mulRelin"}
    assert calls_eager == ["tensorProduct", "tensorProduct"]
    for x, y in ((a1, b1), (a2, b2)):
        assert x._pendT is None and x.primeSet == y.primeSet and x.intFactor == y.intFactor
        assert abs(x.lnNoise - y.lnNoise) < 1e-12
        for h in ("1", "s"):
            assert x.parts[h].getIndexSet() == y.parts[h].getIndexSet()
            assert np.array_equal(x.parts[h].rows, y.parts[h].rows)
    want = B.polymul_mod_phi(ma, mb, m, p)
    assert decrypt(ctx, octx, s, a1, rows) == [int(v) for v in want]


def plain_automorph(msg, m, k, p):
    """F(X) -> F(X^k) mod (Phi_m, p) on coefficient vectors, through the evaluation rows of a
    prime the oracle knows (any prime works: the map is the same permutation of evaluations)."""
    from tests import bgv_ref as B
    return B.automorph_mod_phi(msg, m, k, p)


@pytest.mark.parametrize("m,p,bits,k", [(128, 257, 150, 3), (128, 257, 150, 127), (64, 65537, 250, 5)])
def test_smartAutomorph_decrypts_to_the_rotated_plaintext(m, p, bits, k):
    """Ctxt::smartAutomorph = automorph + reLinearize with the matrix for s(X^k)
    (src/Ctxt.cpp:2437-2515), host logic over the oracle backend; then a multiplication on the
    rotated ciphertext to check the bookkeeping carries on."""
    ctx = hc.ChainContext(m, p, 1, bits=bits, c=3)
    octx = O.Ctx(m)
    for q in ctx.primes:
        octx.add_prime(q)
    s, allp, kb, ka, rows = make_keys(ctx, octx)
    _, _, kbk, kak, _ = make_keys(ctx, octx, auto_k=k)
    ops = OracleOps(octx)
    W, Wk = OKeySwitch(allp, kb, ka), OKeySwitch(allp, kbk, kak)
    rng = np.random.default_rng(21)
    ma, mb = rng.integers(0, p, size=octx.N), rng.integers(0, p, size=octx.N)
    ca = hc.Ctxt.fresh(ctx, ops, *(OPoly(octx, ctx.ctxtPrimes, x) for x in encrypt(ctx, octx, s, ma, 1, rows)), ksw=W)
    cb = hc.Ctxt.fresh(ctx, ops, *(OPoly(octx, ctx.ctxtPrimes, x) for x in encrypt(ctx, octx, s, mb, 2, rows)), ksw=W)
    ca.ksw_auto = {k: Wk}
    with pytest.raises(LookupError):
        ca.clone().smartAutomorph(9 if k != 9 else 11)      # no matrix for that one
    ca.smartAutomorph(k)
    assert set(ca.parts) == {"1", "s"} and ca.primeSet == frozenset(allp)
    rot = plain_automorph(ma, m, k, p)
    assert decrypt(ctx, octx, s, ca, rows) == rot
    ca.multiplyBy(cb)
    from tests import bgv_ref as B
    assert decrypt(ctx, octx, s, ca, rows) == [int(v) for v in B.polymul_mod_phi(rot, mb, m, p)]


# ---- the reference's own Context tests (tests/TestContext.cpp) on the chain builder of this host side ----
def test_buildModChain_throws_when_bits_is_zero():
    """TestContextBGV.buildModChainThrowsWhenBitsIsZero (tests/TestContext.cpp:199-203, m=17 p=2 r=1)"""
    with pytest.raises(ValueError):            # helib::InvalidArgument
        hc.ChainContext(17, 2, 1, bits=0, c=2)


def test_calculate_bit_size_of_Q():
    """TestContextBGV.calculateBitSizeOfQ (:205-225): with bits = 1016, c = 2 the ctxt primes come within 4 % of
    the bits asked for, and bitSizeOfQ is the ceiling of log2 of the product of ALL ctxt and special primes"""
    bits = 1016
    c = hc.ChainContext(17, 2, 1, bits=bits, c=2)
    full = list(c.ctxtPrimes) + list(c.specialPrimes)
    assert abs(math.ceil(c.logOfProduct(c.ctxtPrimes) / math.log(2.0)) - bits) <= 0.04 * bits
    q = 1
    for i in full:
        q *= c.primes[i]
    assert c.bitSizeOfQ() == q.bit_length()     # (exact integer arithmetic; Q is not a power of two)


def test_context_builder_digit_clipping_and_defaults():
    """TestContextBGV.contextBuilderClipsDigitsSizeWithSmallBits / DoesNotClipDigitsSize /
    WithDefaultArguments (:249-306): ContextBuilder<BGV>() defaults m=3, p=2, r=1, bits=300, c=3"""
    assert len(hc.ChainContext(3, 2, 1, bits=100, c=4).digits) < 4      # c clipped: too few ctxt primes
    assert len(hc.ChainContext(3, 2, 1, bits=500, c=5).digits) == 5
    d = hc.ChainContext(3, 2, 1)
    assert len(d.digits) == 3 and len(d.ctxtPrimes) > 0 and len(d.primes) > 0


def test_security_level_formula():
    """Context::securityLevel = lweEstimateSecurity(phi(m), log2(Q/sigma'), hwt) (include/helib/Context.h:875-889,
    src/Context.cpp:34-72): hand-evaluated points of the fit, its lower bound of zero, sparse-key interpolation"""
    assert hc.lweEstimateSecurity(16384, 1000.0, 0) == pytest.approx(3.8 * 16.384 - 20)
    assert hc.lweEstimateSecurity(16384, 1000.0, 120) == pytest.approx(2.4 * 16.384 + 19)
    assert hc.lweEstimateSecurity(16384, 1000.0, 135) == pytest.approx((2.4 + 0.5 * 0.27) * 16.384 + 16.0)
    assert hc.lweEstimateSecurity(16384, 1000.0, 1000) == pytest.approx(3.55 * 16.384 - 12)
    assert hc.lweEstimateSecurity(16, 1000.0, 0) == 0.0                 # never negative
    assert hc.lweEstimateSecurity(16384, 1000.0, 50) == 0.0             # below MIN_SK_HWT
    c = hc.ChainContext(32768, 65537, 1, bits=950, c=3)
    full = list(c.ctxtPrimes) + list(c.specialPrimes)
    want = 3.8 * 16384 / ((c.logOfProduct(full) - math.log(3.2)) / math.log(2.0)) - 20
    assert c.securityLevel() == pytest.approx(want) and 20 < want < 40
    g = hc.ChainContext(1705, 2, 1, bits=200, c=2)                      # general m: sigma' = sigma * sqrt(m)
    full = list(g.ctxtPrimes) + list(g.specialPrimes)
    want = 3.8 * g.phim / ((g.logOfProduct(full) - math.log(3.2 * math.sqrt(1705))) / math.log(2.0)) - 20
    assert g.securityLevel() == pytest.approx(max(0.0, want))
