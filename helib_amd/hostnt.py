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


# ---------------------------------------------------------------------------------------------
# Z_m^* / <p>: the generators HElib rotates along (src/NumbTh.cpp:276-430, src/PAlgebra.cpp:470-507)
# ---------------------------------------------------------------------------------------------
def _conj_classes(classes, g, m):
    """merge the classes of i and i*g for every i (pivot = smallest element); one pass in
    ascending order, non-pivots pick up their pivot's pivot on the way -- src/NumbTh.cpp:279-304"""
    for i in range(m):
        if classes[i] == 0:
            continue
        if classes[i] < i:
            classes[i] = classes[classes[i]]
            continue
        j = i * g % m
        while classes[j] != i:
            classes[classes[j]] = i
            j = j * g % m


def _comp_order(classes, m):
    """order of every element in the current quotient group -- src/NumbTh.cpp:310-341"""
    orders = [0] * m
    if m > 1:
        orders[1] = 1
    for i in range(2, m):
        if classes[i] <= 1:
            orders[i] = 1 if classes[i] == 1 else 0
            continue
        if classes[i] < i:
            orders[i] = orders[classes[i]]
            continue
        j, o = i * i % m, 2
        while classes[j] != 1:
            j = j * i % m
            o += 1
        orders[i] = o
    return orders


def find_generators(m, p, candidates=()):
    """findGenerators (src/NumbTh.cpp:345-430): -> (gens, ords, ordP).  Generators of Z_m^*/<p>
    picked greedily by largest order in the running quotient, preferring an element whose order
    there equals its order in Z_m^* ("quality 2"), then one whose power lands in <p>; `candidates`
    are tried first.  ords are the quotient orders (positive; ZmStar adds the sign)."""
    import math
    classes = [i if math.gcd(i, m) == 1 else 0 for i in range(m)]
    _conj_classes(classes, p % m, m)
    p_subgp = [1 if classes[i] == 1 else 0 for i in range(m)]
    ordP = sum(p_subgp)
    gens, ords, cand = [], [], 0
    candidates = list(candidates)
    while True:
        orders = _comp_order(classes, m)
        idx = 0
        if cand < len(candidates):
            idx = candidates[cand]
            cand += 1
            if orders[idx] <= 1:
                idx = 0
        if idx == 0:
            largest = max(orders) if orders else 1
            if largest > 1:
                best_q, best = 0, -1
                for i in range(m):
                    if best_q >= 2:
                        break
                    if orders[i] == largest:
                        j = pow(i, largest, m)
                        if j == 1:
                            best, best_q = i, 2
                        elif best_q < 1 and p_subgp[j]:
                            best, best_q = i, 1
                idx = best if best > 0 else 0
        if not idx:
            break
        gens.append(idx)
        ords.append(orders[idx])
        _conj_classes(classes, idx, m)
    return gens, ords, ordP


class ZmStar:
    """What the key-switching-matrix families need of PAlgebra: gens, |ords|, native (SameOrd),
    ordP and genToPow (src/PAlgebra.cpp:470-507, 619-637)."""

    def __init__(self, m, p, gens=(), ords=()):
        self.m, self.p = m, p
        gens, ords = list(gens), list(ords)
        if gens and len(gens) == len(ords):       # externally supplied generators and orders
            self.gens, o = gens, ords
            self.ordP = 1
            x = p % m
            while x != 1:
                x = x * p % m
                self.ordP += 1
        else:                                     # (if any) treated as candidates
            self.gens, o, self.ordP = find_generators(m, p, gens)
        self.ords = [abs(int(v)) for v in o]      # a user-supplied negative sign is ignored (:497-501)
        self.native = [pow(g, d, m) == 1 for g, d in zip(self.gens, self.ords)]

    def numOfGens(self):
        return len(self.gens)

    def OrderOf(self, i):
        return self.ords[i]

    def SameOrd(self, i):
        return self.native[i]

    def signedOrds(self):
        """the form Context::writeTo stores: bad dimensions as a negated order"""
        return [d if nat else -d for d, nat in zip(self.ords, self.native)]

    def getNSlots(self):
        n = 1
        for d in self.ords:
            n *= d
        return n

    def genToPow(self, i, j):
        """g_i^j mod m; i == -1: the Frobenius p^j; negative j through the inverse"""
        if i == len(self.gens):
            if j != 0:
                raise ValueError("PAlgebra::genToPow: i == sz but j != 0")
            return 1
        if not -1 <= i < len(self.gens):
            raise ValueError("PAlgebra::genToPow: bad dim")
        base = self.p % self.m if i == -1 else self.gens[i]
        return pow(base, j, self.m)
