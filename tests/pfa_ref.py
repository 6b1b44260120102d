"""Python-integer restatement of the Good-Thomas x Rader form of Cmodulus::FFT / iFFT for an m whose prime
factors are distinct Fermat primes (m = 21845 = 5 * 17 * 257 is BASELINE config 5).  TEST INFRASTRUCTURE.

What it restates (the VALUES, not the method: the reference has only Bluestein, src/bluestein.cpp:134-201):
  forward  y[rank(j)] = f(zeta^j),  j in Z_m^* increasing   (src/CModulus.cpp:431-443),  zeta = root^2
  inverse  X[i] = sum_{j in Z_m^*} y_j zeta^(-i j),  i < m   (src/CModulus.cpp:555-563 + BluesteinFFT with rInv);
           the caller still reduces X modulo Phi_m and multiplies by m^-1 (:571-577)

Method.  m = p_1 ... p_r pairwise coprime, M_k = m / p_k.
  input map   i = sum_k i_k M_k  mod m          (Good-Thomas / Ruritanian: no twiddles between the dimensions)
  output map  j = CRT(j_1 .. j_r),  j_k = j mod p_k
  zeta^(i j) = prod_k omega_k^(i_k j_k),  omega_k = zeta^(M_k)  (order p_k)
so the length-m DFT is DFT_p1 (x) ... (x) DFT_pr, and j in Z_m^*  <=>  every j_k != 0: exactly the outputs Rader's
convolution produces.  Per dimension (p prime, n = p - 1 a power of two, g a generator of Z_p^*):
  forward   S[g^-b] = s[0] + sum_a s[g^a] omega^(g^(a-b)) = s[0] + (u * v)_b,    u_a = s[g^a],     v_c = omega^(g^-c)
  inverse   S'[0] = sum_b y[g^-b];  S'[g^a] = sum_b y[g^-b] omega^(-g^(a-b)) = (u' * v')_a,  u'_b = y[g^-b],  v'_c = omega^(-g^c)
with * the cyclic convolution of length n, done as a length-n number-theoretic transform (rho_n of order n exists:
the chain primes are c 2^36 + 1): decimation in frequency forward (bit-reversed spectrum), pointwise product with
the bit-reversed spectrum of v over n, decimation in time back.  The kernels (helib_amd/csrc/pfa_core.h) follow
this file index for index.
"""


def prime_factors(n):
    out, p = [], 2
    while p * p <= n:
        if n % p == 0:
            out.append(p)
            while n % p == 0:
                n //= p
        p += 1
    if n > 1:
        out.append(n)
    return out


def is_fermat_product(m):
    """m odd, squarefree, every prime factor of the form 2^k + 1."""
    if m % 2 == 0 or m < 3:
        return False
    rest = m
    for p in prime_factors(m):
        if (p - 1) & (p - 2):
            return False
        rest //= p
    return rest == 1


def generator(p):
    """smallest generator of Z_p^*"""
    fs = prime_factors(p - 1)
    g = 2
    while any(pow(g, (p - 1) // f, p) == 1 for f in fs):
        g += 1
    return g


def brev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def root_of_order(q, n):
    """an element of order n = 2^k modulo q (any one: a cyclic convolution does not depend on the choice)."""
    assert (q - 1) % n == 0
    h = 2
    while True:
        r = pow(h, (q - 1) // n, q)
        if n == 1 or pow(r, n // 2, q) != 1:
            return r
        h += 1


def ntt_dif(a, rho, q):
    """in place, natural order in, bit-reversed out: position p holds A[brev(p)]"""
    n = len(a)
    half, tw = n // 2, rho
    while half >= 1:
        for base in range(0, n, 2 * half):
            w = 1
            for k in range(half):
                x, y = a[base + k], a[base + k + half]
                a[base + k] = (x + y) % q
                a[base + k + half] = (x - y) * w % q
                w = w * tw % q
        half //= 2
        tw = tw * tw % q
    return a


def intt_dit(a, rho, q):
    """in place, bit-reversed in, natural out, WITHOUT the 1/n (folded into the fixed operand)"""
    n = len(a)
    rinv = pow(rho, q - 2, q)
    half = 1
    while half < n:
        tw = pow(rinv, n // (2 * half), q)
        for base in range(0, n, 2 * half):
            w = 1
            for k in range(half):
                x, y = a[base + k], a[base + k + half] * w % q
                a[base + k] = (x + y) % q
                a[base + k + half] = (x - y) % q
                w = w * tw % q
        half *= 2
    return a


class Dim:
    """one prime factor p of m: generator, index tables, the fixed operands of both directions"""

    def __init__(self, p, omega, q, rho_big, n_big):
        self.p, self.n, self.q = p, p - 1, q
        self.bits = self.n.bit_length() - 1
        self.g = generator(p)
        ginv = pow(self.g, p - 2, p)
        self.gpow = [pow(self.g, a, p) for a in range(self.n)]       # i = g^a
        self.gipow = [pow(ginv, b, p) for b in range(self.n)]        # j = g^-b
        self.rho = pow(rho_big, n_big // self.n, q)                   # order n
        ninv = pow(self.n, q - 2, q)
        oinv = pow(omega, q - 2, q)
        v = [pow(omega, self.gipow[c], q) for c in range(self.n)]     # forward:  v_c  = omega^(g^-c)
        vi = [pow(oinv, self.gpow[c], q) for c in range(self.n)]      # inverse:  v'_c = omega^(-g^c)
        # bit-reversed spectra over n: what the pointwise product multiplies position p by
        self.vhat = [x * ninv % q for x in ntt_dif(v, self.rho, q)]
        self.vihat = [x * ninv % q for x in ntt_dif(vi, self.rho, q)]

    def conv(self, u, hat):
        a = ntt_dif(list(u), self.rho, self.q)
        a = [x * h % self.q for x, h in zip(a, hat)]
        return intt_dit(a, self.rho, self.q)


class Pfa:
    def __init__(self, m, q, root):
        assert is_fermat_product(m)
        self.m, self.q = m, q
        self.ps = prime_factors(m)
        self.M = [m // p for p in self.ps]
        zeta = root * root % q                                        # src/bluestein.cpp: X_k = sum x_i root^(2 i k)
        n_big = max(p - 1 for p in self.ps)
        rho_big = root_of_order(q, n_big)
        self.dims = [Dim(p, pow(zeta, Mk, q), q, rho_big, n_big) for p, Mk in zip(self.ps, self.M)]
        self.zms = [j for j in range(m) if all(j % p for p in self.ps)]
        self.rank = {j: r for r, j in enumerate(self.zms)}
        self.phim = len(self.zms)

    # -- index maps --
    def in_index(self, idx):
        return sum(i * Mk for i, Mk in zip(idx, self.M)) % self.m

    def crt(self, js):
        j = 0
        for jk, p, Mk in zip(js, self.ps, self.M):
            j += jk * Mk * pow(Mk, -1, p)
        return j % self.m

    def _axes(self, shape, k):
        """all index tuples of `shape` with axis k removed (as lists with a hole at k)"""
        import itertools
        rngs = [range(s) if a != k else [None] for a, s in enumerate(shape)]
        return itertools.product(*rngs)

    def forward(self, x):
        """x: phi(m) coefficients -> y[rank(j)] for j in Z_m^*"""
        import itertools
        q, r = self.q, len(self.ps)
        cur = {}
        for idx in itertools.product(*[range(p) for p in self.ps]):
            i = self.in_index(idx)
            cur[idx] = x[i] % q if i < len(x) else 0
        shape = list(self.ps)
        # one dimension after the other: axis k goes from i_k in [0, p) to b_k in [0, p - 1)  (j_k = g^-b_k)
        for k, d in enumerate(self.dims):
            nxt = {}
            for hole in self._axes(shape, k):
                def at(v):
                    return tuple(v if a == k else h for a, h in enumerate(hole))
                s0 = cur[at(0)]
                c = d.conv([cur[at(d.gpow[a])] for a in range(d.n)], d.vhat)
                for b in range(d.n):
                    nxt[at(b)] = (s0 + c[b]) % q
            shape[k] = d.n
            cur = nxt
        y = [0] * self.phim
        for bs, v in cur.items():
            j = self.crt([d.gipow[b] for d, b in zip(self.dims, bs)])
            y[self.rank[j]] = v
        return y

    def inverse_full(self, y):
        """y[rank(j)] -> X[i], i < m  (the length-m inverse DFT of y scattered onto Z_m^*, before rem Phi_m and 1/m)"""
        import itertools
        q = self.q
        cur = {}
        for bs in itertools.product(*[range(d.n) for d in self.dims]):
            j = self.crt([d.gipow[b] for d, b in zip(self.dims, bs)])
            cur[bs] = y[self.rank[j]] % q
        shape = [d.n for d in self.dims]
        # last dimension first (the data grows: p - 1 inputs, p outputs)
        for k in reversed(range(len(self.dims))):
            d = self.dims[k]
            nxt = {}
            for hole in self._axes(shape, k):
                def at(v):
                    return tuple(v if a == k else h for a, h in enumerate(hole))
                u = [cur[at(b)] for b in range(d.n)]
                c = d.conv(u, d.vihat)
                nxt[at(0)] = sum(u) % q
                for a in range(d.n):
                    nxt[at(d.gpow[a])] = c[a]
            shape[k] = d.p
            cur = nxt
        X = [0] * self.m
        for idx, v in cur.items():
            X[self.in_index(idx)] = v
        return X


def rem_phi_times_minv(X, phi, m, q):
    """(X mod Phi_m) * m^-1: what follows the inverse DFT in src/CModulus.cpp:571-577.  phi = coefficients of Phi_m
    (monic, degree phi(m)), plain long division."""
    X = [v % q for v in X]
    n = len(phi) - 1
    for i in range(len(X) - 1, n - 1, -1):
        c = X[i]
        if c:
            for t in range(n + 1):
                X[i - n + t] = (X[i - n + t] - c * phi[t]) % q
    minv = pow(m, q - 2, q)
    return [v * minv % q for v in X[:n]]


def phi_binomials(m):
    """Phi_m = prod_{d | m} (x^d - 1)^mu(m/d) for squarefree m: (numerator ds, denominator ds)"""
    from itertools import combinations
    ps = prime_factors(m)
    num, den = [], []
    for k in range(len(ps) + 1):
        for sub in combinations(ps, k):
            d = 1
            for p in sub:
                d *= p
            (num if (len(ps) - k) % 2 == 0 else den).append(d)
    return sorted(num), sorted(den)


def rem_by_binomials(X, m, q):
    """(X mod Phi_m) * m^-1 with no multiplication but the last one: the fused tail of the inverse kernel
    (helib_amd/csrc/pfa_core.h, inv_rem).  Multiplying by (1 - x^d) is w_i -= w_(i-d); dividing by it is the running
    sum w_i += w_(i-d).  With t = 1/x:  Qr = Xr / Phi_m(t) mod t^(dq+1)  (Phi_m palindromic, 1 / (t^m - 1) = -1 mod t^m),
    W = Q Phi_m mod x^n,  r = X_low - W."""
    num, den = phi_binomials(m)
    num = [d for d in num if d != m]
    n = sum(1 for j in range(m) if all(j % p for p in prime_factors(m)))
    dq = m - 1 - n
    w = [X[m - 1 - k] % q for k in range(dq + 1)]
    for d in den:
        w = [(w[k] - (w[k - d] if k >= d else 0)) % q for k in range(dq + 1)]
    for d in num:
        for k in range(d, dq + 1):
            w[k] = (w[k] + w[k - d]) % q
    W = [w[dq - k] if k <= dq else 0 for k in range(n)]
    for d in num:
        W = [(W[i] - (W[i - d] if i >= d else 0)) % q for i in range(n)]
    for d in den:
        for i in range(d, n):
            W[i] = (W[i] + W[i - d]) % q
    minv = pow(m, q - 2, q)
    return [(X[i] - W[i]) * minv % q for i in range(n)]
