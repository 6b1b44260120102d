"""Host-side parameter helpers mirroring the reference's setup code for this path:
PrimeGenerator (src/PrimeGenerator.h:41-126) and FindPrimitiveRoot
(src/NumbTh.cpp:436-493).  Pure Python integers (one-time setup, not the hot path)."""


def is_prime(n):
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n == p:
            return True
        if n % p == 0:
            return False
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in small:  # deterministic for n < 2^64
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


def _divc(a, b):
    return -(-a // b)


class PrimeGen:
    """PrimeGenerator(len, m): primes p = 2^k*t*m + 1 in [(1-1/8)*2^len, 2^len)."""
    B = 3

    def __init__(self, length, m):
        if not (self.B <= length <= 60):
            raise ValueError("PrimeGenerator: len is not in [B, HELIB_SP_NBITS]")
        self.len, self.m = length, m
        self.k = 0
        while (m << self.k) <= (1 << (length - self.B)):
            self.k += 1
        self.t = _divc((1 << length) - 1, m << self.k)

    def next(self):
        ln, m, B = self.len, self.m, self.B
        upper = _divc((1 << ln) - 1, m << self.k)
        while True:
            self.t += 1
            if self.t >= upper:
                self.k -= 1
                if self.k < (0 if m % 2 == 0 else 1):
                    raise RuntimeError("Prime generator ran out of primes")
                self.t = _divc((1 << ln) - (1 << (ln - B)) - 1, m << self.k)
                upper = _divc((1 << ln) - 1, m << self.k)
            if self.t % 2 == 0:
                continue
            cand = ((self.t * m) << self.k) + 1
            if is_prime(cand):
                return cand


def find_primitive_root(q, e):
    """FindPrimRootT: deterministic e-th root of unity modulo the prime q."""
    if (q - 1) % e:
        raise ValueError("e does not divide q-1")
    facts, x, p = [], e, 2
    while p * p <= x:
        if x % p == 0:
            facts.append(p)
            while x % p == 0:
                x //= p
        p += 1
    if x > 1:
        facts.append(x)
    root = 1
    for p in facts:
        pp, ee = p, e // p
        while ee % p == 0:
            ee //= p
            pp *= p
        s = 1
        while True:
            s += 1
            while not is_prime(s):
                s += 1
            if pow(s, (q - 1) // p, q) != 1:
                break
        root = root * pow(s, (q - 1) // pp, q) % q
    return root


def phimx(m):
    """Coefficients (lowest first) of the m-th cyclotomic polynomial Phi_m(X), by exact division of
    X^m - 1 by the Phi_d of the proper divisors d of m (PAlgebra's PhimX, src/PAlgebra.cpp)."""
    cache = {}

    def phi(n):
        if n in cache:
            return cache[n]
        num = [-1] + [0] * (n - 1) + [1]                     # X^n - 1
        for d in range(1, n):
            if n % d == 0:
                den = phi(d)
                # exact long division of num by the monic den
                q = [0] * (len(num) - len(den) + 1)
                num = num[:]
                for i in range(len(q) - 1, -1, -1):
                    q[i] = num[i + len(den) - 1]
                    if q[i]:
                        for j, c in enumerate(den):
                            num[i + j] -= q[i] * c
                num = q
        cache[n] = num
        return num
    return phi(m)
