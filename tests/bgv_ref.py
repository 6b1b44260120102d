"""Minimal BGV scaffolding over the CPU oracle (TEST INFRASTRUCTURE).

Generates valid secret keys, key-switching matrices and ciphertexts so that the
multiply/relinearise data path can be checked end-to-end
(decrypt(c*d) == m_c*m_d), following the reference's algebra:
  key-switch matrix  b_j = P*B_j*s' + p*e_j - s*a_j   (include/helib/keySwitching.h:30-90,
                                                       src/keys.cpp:1159-1255)
  encryption         c0 + c1*s = m + p*e               (src/keys.cpp:39-72, RLWE1)
Distributions need not match NTL's PRG; they are seeded numpy draws.
"""
from functools import reduce

import numpy as np

from oracle import oracle as O


def prod(xs):
    return reduce(lambda a, b: a * b, xs, 1)


class Params:
    def __init__(self, m, p, n_ctxt, n_special, digits, bits=40, sp_bits=None, ctx=None):
        self.m, self.p = m, p
        self.ctx = ctx if ctx is not None else O.Ctx(m)
        if ctx is None:
            g = O.PrimeGen(bits, m)
            for _ in range(n_ctxt):
                self.ctx.add_prime(g.next())
            g2 = O.PrimeGen(sp_bits or bits, m) if sp_bits and sp_bits != bits else g
            for _ in range(n_special):
                self.ctx.add_prime(g2.next())
        self.own = list(range(n_ctxt))
        self.special = list(range(n_ctxt, n_ctxt + n_special))
        self.all = self.own + self.special
        self.digits = digits
        self.N = self.ctx.N
        self.primes = self.ctx.primes

    def to_rows(self, coeffs, idx):
        """signed integer coefficients -> eval rows on primes idx"""
        coef = np.array([[int(c) % self.primes[i] for c in coeffs] for i in idx], dtype=np.uint64)
        return self.ctx.fft(idx, coef)

    def mul(self, a, b, idx):
        return np.stack([O.row_op("mul", a[r], b[r], self.primes[i]) for r, i in enumerate(idx)])

    def add(self, a, b, idx):
        return np.stack([O.row_op("add", a[r], b[r], self.primes[i]) for r, i in enumerate(idx)])

    def sub(self, a, b, idx):
        return np.stack([O.row_op("sub", a[r], b[r], self.primes[i]) for r, i in enumerate(idx)])

    def scalar(self, a, k, idx):
        return np.stack([O.row_op("mul_scalar", a[r], int(k) % self.primes[i], self.primes[i])
                         for r, i in enumerate(idx)])

    def uniform(self, idx, seed):
        return np.stack([O.fill_uniform(self.N, self.primes[i], seed * 7919 + i) for i in idx])


def keygen(P, seed=1):
    rng = np.random.default_rng(seed)
    s = rng.integers(-1, 2, size=P.N)
    return s


def small_noise(P, rng):
    return np.rint(rng.normal(0, 3.2, size=P.N)).astype(np.int64)


def encrypt(P, s, msg, seed):
    """c0 + c1*s = msg + p*e  on the ctxt primes"""
    rng = np.random.default_rng(1000 + seed)
    idx = P.own
    c1 = P.uniform(idx, 50 + seed)
    e = small_noise(P, rng)
    rhs = P.to_rows([int(mm) + P.p * int(ee) for mm, ee in zip(msg, e)], idx)
    s_rows = P.to_rows(s, idx)
    c0 = P.sub(rhs, P.mul(c1, s_rows, idx), idx)
    return c0, c1


def gen_ksk(P, s, seed=77):
    """W[s^2 -> s]: (b_j, a_j) for every digit, on ctxt ∪ special primes."""
    rng = np.random.default_rng(seed)
    idx = P.all
    s_rows = P.to_rows(s, idx)
    s2_rows = P.mul(s_rows, s_rows, idx)
    Pspec = prod(P.primes[i] for i in P.special)
    kb, ka = [], []
    B = 1
    for j, d in enumerate(P.digits):
        a = P.uniform(idx, 900 + seed + j)
        e = small_noise(P, rng)
        pe = P.to_rows([P.p * int(x) for x in e], idx)
        t = P.scalar(s2_rows, Pspec * B, idx)
        b = P.sub(P.add(t, pe, idx), P.mul(s_rows, a, idx), idx)
        kb.append(b)
        ka.append(a)
        B *= prod(P.primes[i] for i in d)
    return np.stack(kb), np.stack(ka)


def decrypt(P, s, c0, c1, idx):
    """[c0 + c1*s centred mod prod(idx)] mod p  (src/keys.cpp:1327-1420 core)"""
    s_rows = P.to_rows(s, idx)
    t = P.add(c0, P.mul(c1, s_rows, idx), idx)
    poly = P.ctx.to_poly(idx, t)
    return [v % P.p for v in poly], max(abs(v) for v in poly)


def polymul_mod_phi(a, b, m, q):
    phi = [int(c) for c in O.phimx(m)]
    n = len(phi) - 1
    out = [0] * (2 * n - 1)
    for i, ai in enumerate(a):
        for j, bj in enumerate(b):
            out[i + j] = (out[i + j] + int(ai) * int(bj)) % q
    for i in range(len(out) - 1, n - 1, -1):
        c = out[i]
        if c:
            for j in range(n + 1):
                out[i - n + j] = (out[i - n + j] - c * phi[j]) % q
    return out[:n]


def automorph_mod_phi(a, m, k, q):
    """F(X) -> F(X^k) mod (Phi_m(X), q) on the coefficient vector a (len phi(m))."""
    phi = [int(c) for c in O.phimx(m)]
    n = len(phi) - 1
    full = [0] * m                       # first modulo X^m - 1
    for i, ai in enumerate(a):
        full[(i * k) % m] = (full[(i * k) % m] + int(ai)) % q
    for i in range(m - 1, n - 1, -1):    # then modulo Phi_m (monic)
        c = full[i]
        if c:
            for j in range(n + 1):
                full[i - n + j] = (full[i - n + j] - c * phi[j]) % q
    return full[:n]
