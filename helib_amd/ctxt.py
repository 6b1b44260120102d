"""Host-side mirror of the reference's ciphertext bookkeeping for this path (BGV):

  ChainContext  -- Context::buildModChain (src/Context.cpp:728-1073): small / ctxt / special
                   primes, digits, and ModuliSizes (src/primeChain.cpp:68-335)
  Ctxt          -- the orchestration rows of SURVEY.md 8(a): modUpToSet / modDownToSet /
                   bringToSet (src/Ctxt.cpp:346-562), dropSmallAndSpecialPrimes (:589-662),
                   tensorProduct bookkeeping (:1563-1608), computeIntervalForMul (:1610-1656),
                   multLowLvl / multiplyBy (:1681-1774), reLinearize / keySwitchPart (:720-842),
                   addCtxt for equal prime sets (:1540-1553)

The polynomial work is delegated to a backend (`ops`): on the GPU that is helib_amd.capi
(DoubleCRT parts resident in HBM, possibly a batch of independent ciphertexts sharing this
bookkeeping).  The class itself is pure control flow + floating-point noise estimates, exactly the
part SURVEY.md keeps on the host (risk R3).  Added noise follows the reference's default
branches when the backend can measure it (Ctxt.measure, power-of-two m on the GPU): the mod-switch
noise is sum_parts embeddingLargestCoeff(delta/diffProd) * skBound^power (src/Ctxt.cpp:466-530)
and the key-switch noise is W.noiseBound * sum_digits embeddingLargestCoeff(digit)
(src/DoubleCRT.cpp:530-545, src/Ctxt.cpp:828-841), both norms evaluated on the device (SURVEY
row N1).  A batch of ciphertexts behind one Ctxt shares one bound: the maximum over the batch.
Otherwise (general m, or Ctxt.measure = False) the reference's own alternative branches are used:
modSwitchAddedNoiseBound() (src/Ctxt.cpp:546-559 `#else`) and noiseBoundForUniform per digit
(src/DoubleCRT.cpp:520-529 `#if 0`).  Noise bounds are kept as natural logarithms (NTL::xdouble
in the reference) -- only their logs are ever compared.
"""
import math
from functools import reduce
from helib_amd import timing

from . import hostnt

HELIB_SP_NBITS = 60
LN2 = math.log(2.0)


def phi(m):
    r, x, p = m, m, 2
    while p * p <= x:
        if x % p == 0:
            while x % p == 0:
                x //= p
            r -= r // p
        p += 1
    if x > 1:
        r -= r // x
    return r


def _divc(a, b):
    return -(-a // b)


def handle_powers(handle):
    """(powerOfS, powerOfX) of a part key: "1", "s", "s2", ("s", k) = s(X^k), ("s^", r) = s^r,
    ("s^", r, k) = s^r(X^k)  (SKHandle, include/helib/Ctxt.h:74-170)."""
    if handle == "1":
        return 0, 1
    if handle == "s":
        return 1, 1
    if handle == "s2":
        return 2, 1
    if handle[0] == "s":
        return 1, int(handle[1])
    return int(handle[1]), (int(handle[2]) if len(handle) > 2 else 1)


def handle_of(powerOfS, powerOfX=1):
    if powerOfS == 0:
        return "1"
    if powerOfX == 1:
        return {1: "s", 2: "s2"}.get(powerOfS, ("s^", powerOfS))
    return ("s", powerOfX) if powerOfS == 1 else ("s^", powerOfS, powerOfX)


def handle_mul(h1, h2):
    """SKHandle::mul (include/helib/Ctxt.h:140-165): powers of s add; the automorphism amounts must
    agree unless one side is the constant handle."""
    (s1, x1), (s2, x2) = handle_powers(h1), handle_powers(h2)
    if s1 == 0:
        return h2
    if s2 == 0:
        return h1
    if x1 != x2:
        raise ValueError("cannot multiply parts under different automorphisms")
    return handle_of(s1 + s2, x1)


def _powerOfS(handle):
    return handle_powers(handle)[0]


def polyNormBnd(m):
    """calcPolyNormBnd (src/PAlgebra.cpp:215-434): the ring constant c_M.  1 for a power of two,
    2 cot(pi/(2u))/u when the odd part of m is a power of one prime u; otherwise, with m replaced by
    the radical of its odd part, the maximal absolute row sum of the inverse of the Vandermonde matrix
    of the primitive m-th roots x_j: row i of column j is q_i(x_j) / Phi_m'(x_j) with the Horner
    prefixes q_0 = 1, q_i = q_(i-1) x_j + a_(n-i) of Phi_m (:360-372).  |Phi_m'(x_j)| = prod_i |x_i -
    x_j| is taken as exp of a sum of logarithms (the reference keeps a frexp-normalised running
    product, :300-357); the sums over i for all j at once are one cyclic correlation."""
    import numpy as np
    while m % 2 == 0:
        m //= 2
    if m == 1:
        return 1.0
    fac, r, d = [], m, 3
    while r > 1:
        if r % d == 0:
            fac.append(d)
            while r % d == 0:
                r //= d
        d += 2
    if len(fac) == 1:
        u = fac[0]
        return 2.0 / math.tan(math.pi / (2.0 * u)) / u
    m = 1
    for u in fac:
        m *= u
    phi_coef = np.array(hostnt.phimx(m)[:-1], dtype=np.float64)      # a_0 .. a_(n-1): without the leading 1
    n = len(phi_coef)
    res = np.array([i for i in range(1, m) if math.gcd(i, m) == 1])
    assert len(res) == n
    k = np.arange(m, dtype=np.longdouble)
    x = (np.cos(2 * np.pi * k / m)[res] + 1j * np.sin(2 * np.pi * k / m)[res]).astype(np.complex128)
    # ln |Phi'(x_j)| = sum over units i != j of ln(2 sin(pi |i - j| / m)): correlation of the unit
    # indicator with the table of logarithms (dist 0 counts as 1, as dist_tab[0] = 1 does)
    logd = np.zeros(m)
    logd[1:] = np.log(2.0 * np.sin(np.pi * k[1:] / m).astype(np.float64))
    ind = np.zeros(m)
    ind[res] = 1.0
    corr = np.fft.irfft(np.fft.rfft(logd) * np.conj(np.fft.rfft(ind)), m)   # corr[t] = sum_s ind[s] logd[s+t]
    # sum_i logd[(res_i - res_j) mod m] = sum_s ind[s] logd[(s - res_j) mod m] = corr[-res_j]
    inv_prod = np.exp(-corr[(-res) % m])
    norm_col = np.empty(n)
    q = np.ones(n, dtype=np.complex128)
    norm_col[0] = inv_prod.sum()
    for i in range(1, n):
        q = q * x + phi_coef[n - i]
        norm_col[i] = float(np.dot(np.abs(q), inv_prod))
    return float(norm_col.max())


def _ln(x):
    return math.log(x) if x > 0 else -math.inf


def logaddexp(a, b):
    if a == -math.inf:
        return b
    if b == -math.inf:
        return a
    hi, lo = (a, b) if a >= b else (b, a)
    return hi + math.log1p(math.exp(lo - hi))


class ModuliSizes:
    """src/primeChain.cpp:68-335"""

    def __init__(self, context):
        self.iFFT_cost = 0 if context.pow2 else 20
        sizes = [(0.0, frozenset())]
        for i in context.smallPrimes:
            sq = math.log(context.primes[i])
            sizes += [(s + sq, st | {i}) for s, st in sizes]
        base = list(sizes)
        interval, isz = set(), 0.0
        for i in context.ctxtPrimes:
            interval.add(i)
            isz += math.log(context.primes[i])
            fs = frozenset(interval)
            sizes += [(s + isz, st | fs) for s, st in base]
        sizes.sort(key=lambda e: (e[0], sorted(e[1])))
        self.sizes = sizes
        self._keys = [s for s, _ in sizes]

    def _cost(self, frm, to):
        if self.iFFT_cost == 0:
            return 100 * len(to - frm)
        return 100 * len(to - frm) + self.iFFT_cost * len(frm - to)

    def getSet4Size(self, low, high, from1, from2=None, reverse=False):
        import bisect
        from1 = frozenset(from1)
        from2 = frozenset(from2) if from2 is not None else None
        cost = (lambda s: self._cost(from1, s)) if from2 is None else \
               (lambda s: self._cost(from1, s) + self._cost(from2, s))
        idx = bisect.bisect_left(self._keys, low)
        best, best_cost = -1, None
        ii = idx
        n = len(self.sizes)
        while ii < n and self.sizes[ii][0] <= high:
            c = cost(self.sizes[ii][1])
            if best_cost is None or c <= best_cost:
                best, best_cost = ii, c
            ii += 1
        w = "window1" if from2 is None else "window2"         # src/primeChain.cpp:207-208, 288-289
        timing.STATS_UPDATE(w + "-out", best == -1)
        timing.STATS_UPDATE(w + "-nchoices", ii - idx)
        if best == -1:
            if reverse:
                if ii < n:
                    ub = self.sizes[ii][0] + LN2
                    i = ii
                    while i < n and self.sizes[i][0] <= ub:
                        c = cost(self.sizes[i][1])
                        if best_cost is None or c < best_cost:
                            best, best_cost = i, c
                        i += 1
            elif idx > 0:
                lb = self.sizes[idx - 1][0] - LN2
                i = idx - 1
                while i >= 0 and self.sizes[i][0] >= lb:
                    c = cost(self.sizes[i][1])
                    if best_cost is None or c < best_cost:
                        best, best_cost = i, c
                    i -= 1
        if best == -1:
            return frozenset()
        return self.sizes[best][1]


MIN_SK_HWT = 120     # include/helib/Context.h:34


def lweEstimateSecurity(n, log2AlphaInv, hwt):
    """src/Context.cpp:34-72: security ~ slope * n / log2(1/alpha) + const, slope and constant interpolated
    between the fitted Hamming weights 120, 150, ..., 450 (dense keys: 3.8, -20); never negative"""
    if hwt < 0 or 0 < hwt < MIN_SK_HWT:
        return 0.0
    hw = [120, 150, 180, 210, 240, 270, 300, 330, 360, 390, 420, 450]
    sl = [2.4, 2.67, 2.83, 3.0, 3.1, 3.3, 3.3, 3.35, 3.4, 3.45, 3.5, 3.55]
    cn = [19, 13, 10, 6, 3, 1, -3, -4, -5, -7, -10, -12]
    if hwt == 0:
        slope, const = 3.8, -20.0
    else:
        i = (hwt - 120) // 30
        if i < len(hw) - 1:
            a = (hwt - hw[i]) / (hw[i + 1] - hw[i])
            slope, const = sl[i] + a * (sl[i + 1] - sl[i]), cn[i] + a * (cn[i + 1] - cn[i])
        else:
            slope, const = sl[-1], float(cn[-1])
    return max(0.0, slope * n / log2AlphaInv + const)


class ChainContext:
    """ContextBuilder<BGV>().m(m).p(p).r(r).bits(bits).c(c) -> buildModChain; with ckks=True
    ContextBuilder<CKKS>().m(m).precision(r).bits(bits).c(c): p = -1, plaintext space 1, r = the
    precision in bits (include/helib/Context.h:1040-1130)."""

    def __init__(self, m, p, r=1, bits=300, c=3, stdev=3.2, scale=10.0, skHwt=0, resolution=3,
                 bitsInSpecialPrimes=0, ckks=False):
        self.ckks = bool(ckks)
        if self.ckks:
            p = -1
        self.m, self.p, self.r = m, p, r
        self.ptxtSpace = 1 if self.ckks else p ** r
        self.phim = phi(m)
        self.pow2 = (m & (m - 1)) == 0
        self.stdev, self.scale, self.hwt = stdev, scale, skHwt
        if bits <= 0:      # Context::buildModChain (src/Context.cpp:1044-1046): InvalidArgument
            raise ValueError("Cannot initialise modulus chain with nBits < 1")
        if skHwt < 0:
            raise ValueError("invalid skHwt parameter")
        self.primes = []
        self.smallPrimes, self.ctxtPrimes, self.specialPrimes = [], [], []
        pSize = self._ctxtPrimeSize(bits)
        self._addSmallPrimes(resolution, pSize)
        self._addCtxtPrimes(bits, pSize)
        self._addSpecialPrimes(c, bitsInSpecialPrimes)
        self.modSizes = ModuliSizes(self)

    # ---- size and security of the chain (include/helib/Context.h:857-889, src/Context.cpp:34-72) ----
    def bitSizeOfQ(self):
        """ceil(log2 of the product of the ctxt and special primes)"""
        return int(math.ceil(self.logOfProduct(list(self.ctxtPrimes) + list(self.specialPrimes)) / LN2))

    def securityLevel(self):
        """Context::securityLevel: lweEstimateSecurity(phi(m), log2(Q / stdev'), hwt) -- the reference's
        affine fits to the LWE estimator, by Hamming weight of the secret key (0 = dense)."""
        s = self.stdev if self.pow2 else self.stdev * math.sqrt(self.m)
        log2AlphaInv = (self.logOfProduct(list(self.ctxtPrimes) + list(self.specialPrimes)) - math.log(s)) / LN2
        return lweEstimateSecurity(self.phim, log2AlphaInv, self.hwt)

    # ---- chain construction (src/Context.cpp:728-1035) ----
    @staticmethod
    def _bit_loss():
        return -math.log1p(-1.0 / (1 << hostnt.PrimeGen.B)) / LN2

    def _ctxtPrimeSize(self, nBits):
        bit_loss = self._bit_loss()
        nPrimes = int(math.ceil(nBits / (HELIB_SP_NBITS - bit_loss)))
        t = HELIB_SP_NBITS
        while 10 * (t - 1) >= 9 * HELIB_SP_NBITS and (t - 1) >= 30 and \
                ((t - 1) - bit_loss) * nPrimes >= nBits:
            t -= 1
        return t

    def _add(self, q, where):
        assert q not in self.primes, "Prime q is already in the prime chain"
        self.primes.append(q)
        where.append(len(self.primes) - 1)

    def _addSmallPrimes(self, resolution, cpSize):
        if resolution < 1 or resolution > 10:
            resolution = 3
        sizes = []
        if cpSize >= 54:
            smallest = _divc(2 * cpSize, 3)
        elif cpSize >= 45:
            smallest = _divc(7 * cpSize, 10)
        else:
            smallest = _divc(11 * cpSize, 15)
            sizes.append(smallest)
        sizes += [smallest, smallest]
        delta = resolution
        while cpSize - delta > smallest:
            sizes.append(cpSize - delta)
            delta *= 2
        if cpSize - 3 * resolution > smallest:
            sizes.append(cpSize - 3 * resolution)
        if resolution == 1 and cpSize - 11 > smallest:
            sizes.append(cpSize - 11)
        sizes.sort()
        last, gen = 0, None
        for sz in sizes:
            if sz != last:
                gen = hostnt.PrimeGen(sz, self.m)
            self._add(gen.next(), self.smallPrimes)
            last = sz

    def _addCtxtPrimes(self, nBits, targetSize):
        gen = hostnt.PrimeGen(targetSize, self.m)
        bitlen = 0.0
        while bitlen < nBits - 0.5:
            q = gen.next()
            self._add(q, self.ctxtPrimes)
            bitlen += math.log2(q)

    def _addSpecialPrimes(self, nDgts, bitsInSpecialPrimes):
        n = len(self.ctxtPrimes)
        nDgts = max(1, min(nDgts, n))
        digits = []
        if nDgts > 1:
            remaining = list(self.ctxtPrimes)
            for dgt in range(nDgts - 1):
                card = _divc(len(remaining), nDgts - dgt)
                digits.append(remaining[:card])
                remaining = remaining[card:]
            if remaining:
                digits.append(remaining)
        else:
            digits = [list(self.ctxtPrimes)]
        self.digits = digits
        maxDigitLog = max(self.logOfProduct(d) for d in digits)
        nDgts = len(digits)
        if bitsInSpecialPrimes:
            nBits = bitsInSpecialPrimes
        else:
            h = self.phim / 2.0 if self.hwt == 0 else self.hwt
            log_phim = max(math.log(self.phim), 1.0)
            p2e = self.ptxtSpace
            if self.ckks:   # a smaller noise estimate, to protect precision (src/Context.cpp:957-965)
                nBits = (maxDigitLog + math.log(self.stdev) + math.log(nDgts) - 0.5 * math.log(h)) / LN2
            elif self.pow2:
                nBits = (maxDigitLog + math.log(p2e) + math.log(self.stdev) + 0.5 * math.log(12.0) +
                         math.log(nDgts) - 0.5 * math.log(log_phim) - 2 * math.log(self.p) -
                         math.log(h)) / LN2
            else:
                nBits = (maxDigitLog + math.log(self.m) + math.log(p2e) + math.log(self.stdev) +
                         0.5 * math.log(12.0) + math.log(nDgts) - 0.5 * log_phim -
                         0.5 * math.log(log_phim) - 2 * math.log(self.p) - math.log(h)) / LN2
        nBits = max(nBits, 1.0)
        bit_loss = self._bit_loss()
        nPrimes = int(math.ceil(nBits / (HELIB_SP_NBITS - bit_loss)))
        t = HELIB_SP_NBITS
        while (t - 1) >= 0.55 * HELIB_SP_NBITS and (t - 1) >= 30 and \
                ((t - 1) - bit_loss) * nPrimes >= nBits:
            t -= 1
        gen = hostnt.PrimeGen(t, self.m)
        while nPrimes > 0:
            q = gen.next()
            if q in self.primes:
                continue
            self._add(q, self.specialPrimes)
            nPrimes -= 1

    # ---- helpers ----
    def logOfPrime(self, i):
        return math.log(self.primes[i])

    def logOfProduct(self, s):
        return sum(math.log(self.primes[i]) for i in s)

    def productOfPrimes(self, s):
        return reduce(lambda a, b: a * b, (self.primes[i] for i in s), 1)

    def noiseBoundForUniform(self, magBound, degBound):
        return self.scale * math.sqrt(degBound / 3.0) * magBound

    def encodeRoundingError(self):
        """EncryptedArrayCx::encodeRoundingError (include/helib/EncryptedArray.h:1287-1297)"""
        return self.noiseBoundForUniform(0.5, self.phim)

    def encodeScalingFactor(self, precision=-1, roundErr=-1.0):
        """EncryptedArrayCx::encodeScalingFactor (include/helib/EncryptedArray.h:1299-1312): the factor CKKS
        plaintexts are scaled by before rounding -- ceil(precision * roundErr) rounded up to a power of two, with
        precision defaulting to 2^r of ContextBuilder<CKKS>::precision(r).  (m = 65536: 2^11 at precision(1),
        2^30 at precision(20).)"""
        if precision <= 0:
            precision = 1 << self.r
        if roundErr < 0:
            roundErr = self.encodeRoundingError()
        f = int(math.ceil(precision * roundErr))
        return 1 << max(f - 1, 0).bit_length()       # NTL::NextPowerOfTwo

    def noiseBoundForMod(self, modulus, degBound):
        var = modulus * modulus / 12.0 + (1.0 / 6.0 if modulus % 2 == 0 else 0.0)
        return self.scale * math.sqrt(degBound * var)

    # sample*Bounded return values (src/sample.cpp:260-267, 342-396, 445-463)
    def skBound(self):
        if self.hwt > 0:
            return math.sqrt(self.hwt * math.log(self.phim))
        return math.sqrt(self.phim * math.log(self.phim) / 2.0)

    def gaussBound(self):
        eff = math.sqrt(self.phim * math.log(self.phim)) if self.pow2 else \
            math.sqrt(self.m * math.log(self.phim))
        st = self.stdev if self.pow2 else self.stdev * math.sqrt(self.m)
        return st * eff

    def freshNoiseBound(self):
        """noiseBound of PubKey::Encrypt output (src/keys.cpp:395-475)."""
        p = self.ptxtSpace
        e = self.gaussBound() * p
        pk_noise = e                                   # RLWE1: bound *= p
        r_bound = math.sqrt(self.phim * math.log(self.phim) / 2.0)
        return r_bound * pk_noise + e + e * self.skBound() + self.noiseBoundForMod(p, self.phim)


class Ctxt:
    """BGV ciphertext: parts keyed by secret-key handle ("1", "s", "s2")."""
    safety = LN2  # src/Ctxt.cpp:39
    measure = True  # measured added noise (the reference's default) when the backend supports it
    lazyTensor = True  # multiplyBy leaves the tensor product to the mod-switch / key-switch kernels (tests A/B this)

    def __init__(self, context, ops, ksw=None, ksw_ptxtSpace=None, ksw_noise=None):
        self.context, self.ops = context, ops
        self._meas = bool(Ctxt.measure and hasattr(ops, "supportsNorms") and ops.supportsNorms(context.m))
        self._ln = -math.inf
        self._pending = []     # deferred noise updates waiting for norms still on the device
        self.ksw_auto = {}     # k -> key-switching matrix from s(X^k) to s (PubKey::getKeySWmatrix)
        self.ksw_pow = {}      # r -> key-switching matrix from s^r to s, r >= 3 (r = 2 is self.ksw)
        self.ksw_map = None    # PubKey::keySwitchMap: k -> first step n on the way to X -> X^k (0: none)
        self.parts = {}
        # multiplyBy only (as the C++ host's Ctxt::pendingTensor): the tensor product's bookkeeping is done, its
        # data not yet -- the four operand parts wait here while `parts` holds the three product handles with no
        # storage; the mod-switch or the key switch that follows forms the product inside its own kernels
        # (ops.tensorBringToSet / ops.mulRelin), anything else calls _materializeTensor() first
        self._pendT = None
        self.primeSet = frozenset()
        self.ptxtSpace = context.ptxtSpace
        self.intFactor = 1
        # CKKS (src/Ctxt.h: ptxtMag, ratFactor): |plaintext| bound and ln of the scaling factor
        self.ptxtMag, self.lnRatFactor = 1.0, 0.0
        self.ksw, self.ksw_ptxtSpace = ksw, ksw_ptxtSpace or context.ptxtSpace
        self.ksw_lnNoise = ksw_noise if ksw_noise is not None else \
            math.log(context.gaussBound() * context.ptxtSpace)

    @classmethod
    def fresh(cls, context, ops, c0, c1, ksw=None, **kw):
        c = cls(context, ops, ksw, **kw)
        c.parts = {"1": c0, "s": c1}
        c.primeSet = frozenset(context.ctxtPrimes)
        c.lnNoise = math.log(context.freshNoiseBound())
        return c

    def clone(self):
        self._materializeTensor()
        c = Ctxt(self.context, self.ops, self.ksw, self.ksw_ptxtSpace, self.ksw_lnNoise)
        c.parts = {h: p.copy() for h, p in self.parts.items()}
        c.primeSet, c.ptxtSpace = self.primeSet, self.ptxtSpace
        c.lnNoise, c.intFactor = self.lnNoise, self.intFactor
        c.ptxtMag, c.lnRatFactor = self.ptxtMag, self.lnRatFactor
        c.ksw_auto = self.ksw_auto
        c.ksw_pow = self.ksw_pow
        c.ksw_map = self.ksw_map
        return c

    # ---- noise estimate: ln(noiseBound) ----
    # Measured norms are read back lazily: an operation enqueues its kernels, registers how the
    # norms will enter the estimate, and returns; the estimate is completed (waiting only for the
    # norm kernels, hx_norms_flush) when somebody reads it -- normally the next prime-set
    # decision, by which time more work is already queued behind on the GPU.
    @property
    def lnNoise(self):
        if self._pending:
            if hasattr(self.ops, "normsFlush") and self.parts:
                some = self._pendT[0] if self._pendT is not None else next(iter(self.parts.values()))
                self.ops.normsFlush(some)
            todo, self._pending = self._pending, []
            for fn in todo:
                fn()
        return self._ln

    @lnNoise.setter
    def lnNoise(self, v):
        if self._pending:
            self.lnNoise  # noqa: B018 -- complete earlier updates first, in order
        self._ln = v

    def _defer(self, fn):
        if self._meas:
            self._pending.append(fn)
        else:
            fn()

    def _materializeTensor(self):
        """the pending tensor product's data, now (Ctxt::tensorProduct's DoubleCRT work, src/Ctxt.cpp:1576-1597)"""
        if self._pendT is None:
            return
        t, self._pendT = self._pendT, None
        t0, t1, t2 = self.ops.tensorProduct(*t)
        self.parts = {"1": t0, "s": t1, "s2": t2}

    # ---- bookkeeping ----
    def logOfPrimeSet(self):
        return self.context.logOfProduct(self.primeSet)

    def modSwitchAddedNoiseBound(self):
        h = self.context.skBound()
        added = sum(h ** _powerOfS(k) for k in self.parts)
        return added * self.context.noiseBoundForUniform(self.ptxtSpace / 2.0, self.context.phim)

    # ---- prime-set maintenance ----
    @timing.timed
    def modUpToSet(self, s):
        diff = sorted(frozenset(s) - self.primeSet)
        if not diff:
            return
        self._materializeTensor()
        for p in self.parts.values():
            p.addPrimesAndScale(diff)
        self.lnNoise += self.context.logOfProduct(diff)
        self.lnRatFactor += self.context.logOfProduct(diff)   # "If CKKS, the rational factor grows" (:366)
        self.primeSet = self.primeSet | frozenset(diff)

    @timing.timed
    def modDownToSet(self, s):
        inter = self.primeSet & frozenset(s)
        if not inter:
            raise RuntimeError(f"modDownToSet called from {sorted(self.primeSet)} to {sorted(s)}")
        diff = self.primeSet - inter
        if not diff:
            return
        added = Ctxt._modDownParts([self], sorted(inter))[0]
        logdiff = self.context.logOfProduct(diff)
        self.lnRatFactor -= logdiff                              # ratFactor /= f (:533, :553)
        bound = math.log(self.modSwitchAddedNoiseBound()) if timing.fhe_stats else None

        def update():
            add = _ln(added())
            if bound is not None:   # src/Ctxt.cpp:535-537: added noise over its a-priori bound
                timing.STATS_UPDATE("mod-switch-added-noise", math.exp(add - bound))
            self._ln = logaddexp(self._ln - logdiff, add)
        self._defer(update)
        self.primeSet = inter

    @staticmethod
    def _modDownParts(cts, keep, add=()):
        """The polynomial work of modDownToSet (after a mod-up by `add`, if given) on all parts of
        ciphertexts that share one prime set, in one backend call where the backend has one.
        Returns, per ciphertext, a function giving its added noise: measured, sum over parts of
        embeddingLargestCoeff(fdelta) * h^power (src/Ctxt.cpp:495-527) -- to be called once the
        norms have been read back (Ctxt.lnNoise does) -- else the bound."""
        a = cts[0]
        ops, meas, ptxt = a.ops, a._meas, a.ptxtSpace
        kw = {"norms": True, "defer": True} if meas else {}
        pend = None
        if len(cts) == 1 and a._pendT is not None:   # the tensor product is formed inside this mod-switch
            pend, a._pendT = a._pendT, None
        else:
            for c in cts:
                c._materializeTensor()
        parts = [p for c in cts for p in c.parts.values()]
        if pend is not None:
            res = ops.tensorBringToSet(*pend, list(add), keep, ptxt, **kw)
            outs, norms = res if meas else (res, None)
            a.parts = {"1": outs[0], "s": outs[1], "s2": outs[2]}
        elif add and hasattr(ops, "bringToSetMulti"):
            norms = ops.bringToSetMulti(parts, add, keep, ptxt, **kw)
        elif add:
            for p in parts:
                p.addPrimesAndScale(list(add))
            norms = [p.scaleDownToSet(keep, ptxt, **({"norms": True} if meas else {})) for p in parts]
        elif hasattr(ops, "scaleDownToSetMulti"):
            norms = ops.scaleDownToSetMulti(parts, keep, ptxt, **kw)
        else:
            norms = [p.scaleDownToSet(keep, ptxt, **({"norms": True} if meas else {})) for p in parts]
        if not meas:
            return [c.modSwitchAddedNoiseBound for c in cts]
        h = a.context.skBound()
        out, k = [], 0
        for c in cts:
            keys = list(c.parts)
            out.append(lambda k0=k, keys=keys: sum(float(max(norms[k0 + i])) * h ** _powerOfS(key)
                                                   for i, key in enumerate(keys)))
            k += len(keys)
        return out

    @timing.timed
    def bringToSet(self, s):
        s = frozenset(s) if s else frozenset([self.context.ctxtPrimes[0]])
        if hasattr(self.ops, "bringToSetMulti") or self._pendT is not None:
            Ctxt._bringManyToSet([self], s)
            return
        self.modUpToSet(s)
        self.modDownToSet(s)

    @staticmethod
    def _bringManyToSet(cts, s):
        """bringToSet(s) = modUpToSet(s); modDownToSet(s) for ciphertexts that sit on one prime
        set, with the polynomial work of all their parts in one backend call (the mod-up scaling
        is folded into the mod-down kernels on the GPU).  Bookkeeping as in src/Ctxt.cpp:346-562."""
        a = cts[0]
        add = sorted(s - a.primeSet)
        up = a.primeSet | frozenset(add)
        inter = up & s
        if not inter:
            raise RuntimeError(f"modDownToSet called from {sorted(up)} to {sorted(s)}")
        diff = up - inter
        if not add and not diff:
            return
        if not diff:        # a pure mod-up: no mod-switch kernel to form a pending product in
            for c in cts:
                c._materializeTensor()
        added = Ctxt._modDownParts(cts, sorted(inter), add=add or ())
        for c, ad in zip(cts, added):
            c.lnNoise += c.context.logOfProduct(add)
            c.lnRatFactor += c.context.logOfProduct(add) - c.context.logOfProduct(diff)
            c.primeSet = up
            if diff:
                logdiff = c.context.logOfProduct(diff)
                c._defer(lambda c=c, ad=ad, logdiff=logdiff:
                         setattr(c, "_ln", logaddexp(c._ln - logdiff, _ln(ad()))))
                c.primeSet = inter

    @staticmethod
    def _bringBothToSet(a, b, s):
        """bringToSet(s) on two ciphertexts that sit on the same prime set: identical arithmetic
        and bookkeeping to calling it on each, with one batched mod-switch for all parts."""
        s = frozenset(s) if s else frozenset([a.context.ctxtPrimes[0]])
        a.modUpToSet(s)
        b.modUpToSet(s)
        inter = a.primeSet & s
        if not inter:
            raise RuntimeError(f"modDownToSet called from {sorted(a.primeSet)} to {sorted(s)}")
        diff = a.primeSet - inter
        if not diff:
            return
        added = Ctxt._modDownParts([a, b], sorted(inter))
        logdiff = a.context.logOfProduct(diff)
        for c, ad in zip((a, b), added):
            c.lnRatFactor -= logdiff
            c._defer(lambda c=c, ad=ad: setattr(c, "_ln", logaddexp(c._ln - logdiff, _ln(ad()))))
            c.primeSet = inter

    @timing.timed
    def dropSmallAndSpecialPrimes(self):
        ctx = self.context
        small, ctp = frozenset(ctx.smallPrimes), frozenset(ctx.ctxtPrimes)
        if not (self.primeSet & small):
            self.modDownToSet(ctp)
            return
        target = set(self.primeSet & ctp)
        dropping = self.primeSet - target
        log_dropping = ctx.logOfProduct(dropping)
        log_msn = math.log(self.modSwitchAddedNoiseBound()) + 3 * LN2
        comp = 0.0
        if self.lnNoise - log_dropping + comp < log_msn:
            for i in sorted(ctp - target):
                target.add(i)
                comp += ctx.logOfPrime(i)
                if self.lnNoise - log_dropping + comp >= log_msn:
                    break
        self.bringToSet(target)

    # ---- arithmetic ----
    # ---- size / correctness accessors (include/helib/Ctxt.h:1291-1325, src/Ctxt.cpp:116-127) ----
    def lnTotalNoiseBound(self):
        """ln of totalNoiseBound(): for CKKS ptxtMag*ratFactor + noiseBound, else noiseBound"""
        if self.context.ckks:
            return logaddexp(_ln(self.ptxtMag) + self.lnRatFactor, self.lnNoise)
        return self.lnNoise

    def capacity(self):
        """log2 of the modulus over the TOTAL noise bound (at least 1)"""
        return (self.logOfPrimeSet() - max(self.lnTotalNoiseBound(), 0.0)) / LN2

    def bitCapacity(self):
        return int(self.capacity())

    def isCorrect(self):
        """totalNoiseBound * polyNormBnd <= 0.48 Q: would this ciphertext decrypt without errors?"""
        return self.lnTotalNoiseBound() + math.log(polyNormBnd(self.context.m)) <= \
            math.log(0.48) + self.logOfPrimeSet()

    def frobeniusAutomorph(self, j):
        """Ctxt::frobeniusAutomorph (src/Ctxt.cpp:2526-2545): X -> X^(p^j) for BGV (j mod ord(p));
        for CKKS complex conjugation when j is odd (X -> X^(m-1))."""
        if not self.parts or j == 0:
            return
        ctx = self.context
        if ctx.ckks:
            if j & 1:
                self.smartAutomorph(ctx.m - 1)
            return
        d, x = 1, ctx.p % ctx.m
        while x != 1:
            x = x * ctx.p % ctx.m
            d += 1
        j %= d
        if j:
            self.smartAutomorph(pow(ctx.p, j, ctx.m))

    def mulIntFactor(self, e):
        """Ctxt::mulIntFactor (src/Ctxt.cpp:331-340)"""
        if e == 1:
            return
        self.intFactor = self.intFactor * e % self.ptxtSpace
        bal = e - self.ptxtSpace if e > self.ptxtSpace // 2 else e
        for p in self.parts.values():
            p.mulConstant(bal)
        self.lnNoise = self.lnNoise + math.log(abs(bal))

    def negate(self):
        for p in self.parts.values():
            p.Negate()

    @staticmethod
    def equalizeRationalFactors(c1, c2):
        """Ctxt::equalizeRationalFactors (src/Ctxt.cpp:1212-1356): scale both CKKS ciphertexts by
        small integers (continued-fraction convergents of the ratio of their factors) until they
        share one factor, stopping as soon as the discretisation error is within sqrt(2) of the
        error the sum has anyway.  Computed relative to the smaller factor, so plain doubles do
        (the reference's xdouble carries the same 53 bits)."""
        big, small = (c1, c2) if c1.lnRatFactor > c2.lnRatFactor else (c2, c1)
        base = small.lnRatFactor
        x = math.exp(big.lnRatFactor - base)
        r = c1.context.r
        denomBound = 1 << (r + 1)
        epsilon = 0.125 / denomBound
        a = int(math.floor(x + epsilon))
        xi = x - a
        prevDenom, denom = 0, 1
        numer = int(math.floor(denom * x + 0.5))
        m1, of1, oe1 = big.ptxtMag, x, math.exp(big.lnNoise - base)
        m2, of2, oe2 = small.ptxtMag, 1.0, math.exp(small.lnNoise - base)
        target = oe1 / of1 + oe2 / of2

        def calc_err(f, f1, e1, f2, e2):
            return m1 * abs(f1 / f - 1.0) + m2 * abs(f2 / f - 1.0) + (e1 + e2) / f
        while True:
            f1, e1 = of1 * denom, oe1 * denom
            f2, e2 = of2 * numer, oe2 * numer
            err1, err2 = calc_err(f1, f1, e1, f2, e2), calc_err(f2, f1, e1, f2, e2)
            if err1 < err2:
                f, fe1, fe2, err = f1, e1, e2 + m2 * abs(f2 - f1), err1
            else:
                f, fe1, fe2, err = f2, e1 + m1 * abs(f2 - f1), e2, err2
            if err < math.sqrt(2.0) * target or xi <= 0:
                break
            xi = 1.0 / xi
            ai = int(math.floor(xi + epsilon))
            xi -= ai
            tmpDenom = denom * ai + prevDenom
            if tmpDenom > denomBound:
                break
            prevDenom, denom = denom, tmpDenom
            numer = int(math.floor(denom * x + 0.5))
        if denom != 1:
            for p in big.parts.values():
                p.mulConstant(denom)
        if numer != 1:
            for p in small.parts.values():
                p.mulConstant(numer)
        big.lnRatFactor = small.lnRatFactor = math.log(f) + base
        big.lnNoise, small.lnNoise = _ln(fe1) + base, _ln(fe2) + base

    @timing.timed
    def addCtxt(self, other, negative=False):
        """Ctxt::addCtxt (src/Ctxt.cpp:1405-1556): plaintext spaces reduced to their gcd (BGV),
        both operands mod-switched UP to the union of their prime sets, CKKS factors equalised,
        BGV intFactors harmonised by the (e1, e2) of least noise along the extended Euclidean
        sequence, then the parts added handle by handle."""
        ctx = self.context
        if not other.parts:
            return
        if not self.parts:
            c = other.clone()
            self.__dict__.update(c.__dict__)
            if negative:
                self.negate()
            return
        o = other
        owned = False

        def own():
            nonlocal o, owned
            if not owned:
                o, owned = other.clone(), True
            return o
        if ctx.ckks:
            if self.ptxtSpace != 1 or other.ptxtSpace != 1:
                raise ValueError("Plaintext spaces incompatible")
        else:
            g = math.gcd(self.ptxtSpace, other.ptxtSpace)
            if g <= 1:
                raise ValueError("New and old plaintext spaces are coprime")
            self.ptxtSpace = g
            self.intFactor %= g
            if other.ptxtSpace != g:
                own().ptxtSpace = g
                o.intFactor %= g
        if o.primeSet - self.primeSet:
            self.modUpToSet(self.primeSet | o.primeSet)
        if self.primeSet - o.primeSet:
            own().modUpToSet(self.primeSet)
        if ctx.ckks:
            Ctxt.equalizeRationalFactors(self, own())
        e1 = e2 = 1
        if not ctx.ckks and self.intFactor != o.intFactor:
            P = self.ptxtSpace
            ratio = o.intFactor * pow(self.intFactor, -1, P) % P
            bal = lambda e: abs(e - P if e > P // 2 else e)                 # noqa: E731
            norm = lambda a, b: logaddexp(self.lnNoise + _ln(bal(a)), o.lnNoise + _ln(bal(b)))  # noqa: E731
            r0, t0, r1, t1 = P, 0, ratio, 1
            e1, e2 = r1, t1
            best = norm(e1, e2)
            while r1 != 0:
                q = r0 // r1
                r0, r1, t0, t1 = r1, r0 % r1, t1, t0 - t1 * q
                a, b = r1 % P, t1 % P
                if a % ctx.p != 0:
                    cand = norm(a, b)
                    if cand < best:
                        e1, e2, best = a, b, cand
            assert e1 * self.intFactor % P == e2 * o.intFactor % P
            assert math.gcd(e1, P) == 1 and math.gcd(e2, P) == 1
        if e2 != 1:
            own().mulIntFactor(e2)
        if e1 != 1:
            self.mulIntFactor(e1)
        for h, p in o.parts.items():
            if h in self.parts:
                if negative:
                    self.parts[h] -= p
                else:
                    self.parts[h] += p
            else:
                self.parts[h] = p.copy()
                if negative:
                    self.parts[h].Negate()
        self.ptxtMag += o.ptxtMag
        self.lnNoise = logaddexp(self.lnNoise, o.lnNoise)

    @staticmethod
    def computeIntervalForMul(c1, c2):
        cap1 = c1.logOfPrimeSet() - max(c1.lnNoise, 0.0)
        cap2 = c2.logOfPrimeSet() - max(c2.lnNoise, 0.0)
        adn1 = math.log(c1.modSwitchAddedNoiseBound())
        adn2 = math.log(c2.modSwitchAddedNoiseBound())
        if c1.context.ckks:     # the opposite end: keep n*q'/q above the added noise (:1637-1651)
            lo = max(cap1 + adn1, cap2 + adn2) + Ctxt.safety
            return lo, lo + 4 * LN2
        hi = min(cap1 + adn1, cap2 + adn2) - Ctxt.safety
        return hi - 4 * LN2, hi

    @timing.timed
    def multLowLvl(self, other, destructive=False, lazyTensor=False):
        o = other if destructive else other.clone()
        ckks = self.context.ckks
        if ckks:
            assert self.ptxtSpace == 1 and o.ptxtSpace == 1, "Plaintext spaces incompatible"
        else:
            g = math.gcd(self.ptxtSpace, o.ptxtSpace)
            assert g > 1, "Plaintext spaces are co-prime"
            self.ptxtSpace = o.ptxtSpace = g
            self.intFactor %= g
            o.intFactor %= g
        lo, hi = Ctxt.computeIntervalForMul(self, o)
        common = self.context.modSizes.getSet4Size(lo, hi, self.primeSet, o.primeSet, ckks)
        if self.primeSet == o.primeSet and hasattr(self.ops, "bringToSetMulti"):
            Ctxt._bringManyToSet([self, o], frozenset(common) if common else
                                 frozenset([self.context.ctxtPrimes[0]]))
        elif self.primeSet == o.primeSet and hasattr(self.ops, "scaleDownToSetMulti"):
            Ctxt._bringBothToSet(self, o, common)   # same result, the 4 parts share the launches
        else:
            self.bringToSet(common)
            o.bringToSet(common)
        self._tensorProduct(o, lazyTensor)

    def _tensorProduct(self, o, lazy=False):
        if self.ptxtSpace > 2:
            q = self.context.productOfPrimes(self.primeSet) % self.ptxtSpace
            self.intFactor = self.intFactor * o.intFactor % self.ptxtSpace * q % self.ptxtSpace
        if set(self.parts) == {"1", "s"} and set(o.parts) == {"1", "s"}:
            four = (self.parts["1"], self.parts["s"], o.parts["1"], o.parts["s"])
            if lazy and Ctxt.lazyTensor and hasattr(self.ops, "tensorBringToSet"):
                # the product's data is left to the kernels that consume it (include/helib_amd_ctxt.hpp, pendingTensor)
                self._pendT = four
                self.parts = {"1": None, "s": None, "s2": None}
            else:
                t0, t1, t2 = self.ops.tensorProduct(*four)
                self.parts = {"1": t0, "s": t1, "s2": t2}
        else:   # any parts (src/Ctxt.cpp:1576-1597): all pairwise products, accumulated by handle
            new = {}
            for h1, p1 in self.parts.items():
                for h2, p2 in o.parts.items():
                    h = handle_mul(h1, h2)
                    t = p1.copy()
                    t *= p2
                    if h in new:
                        new[h] += t
                    else:
                        new[h] = t
            self.parts = new
        if self.context.ckks:   # totalNoiseBound = factor*ptxt + noiseBound on both sides (:1600-1606)
            n1, n2 = self.lnNoise, o.lnNoise
            self.lnNoise = logaddexp(logaddexp(n1 + _ln(o.ptxtMag) + o.lnRatFactor,
                                               n2 + _ln(self.ptxtMag) + self.lnRatFactor), n1 + n2)
            self.lnRatFactor += o.lnRatFactor
            self.ptxtMag *= o.ptxtMag
        else:
            self.lnNoise = self.lnNoise + o.lnNoise

    @timing.timed
    def reLinearize(self):
        """Ctxt::reLinearize (src/Ctxt.cpp:720-786) for the shapes of this path: (1, s, s^2) after a
        multiplication, or (1, [s,] s(X^k)) after an automorphism."""
        other = [h for h in self.parts if h not in ("1", "s")]
        if not other:
            return
        if len(other) > 1:
            self._materializeTensor()
            return self._reLinearizeMany(other)
        hnd = other[0]
        W = self._matrixFor(hnd)
        ctx = self.context
        self.dropSmallAndSpecialPrimes()
        self._relin_CKKS_adjust()
        sp = list(ctx.specialPrimes)
        # No mod-switch consumed a pending tensor product (a fresh CKKS product, a product at a level that needs
        # none): at the full level the key switch takes the operands themselves (ops.mulRelin = hx_mul_relin);
        # otherwise the product is formed now.
        pend = None
        if self._pendT is not None:
            cp = list(ctx.ctxtPrimes)
            if (hasattr(self.ops, "mulRelin") and self._pendT[0].getIndexSet() == cp and frozenset(cp) == self.primeSet
                    and list(getattr(W, "row_idx", ())) == cp + sp):
                pend, self._pendT = self._pendT, None
            else:
                self._materializeTensor()
        logProd = ctx.logOfProduct(sp)
        self.lnRatFactor += logProd                              # CKKS factor after mod-up (:757)
        # digits of the context restricted to the current prime set (src/DoubleCRT.cpp:485-493)
        digits = [[i for i in d if i in self.primeSet] for d in ctx.digits]
        digits = [d for d in digits if d]
        if self.ptxtSpace > 1:   # g == 1 for CKKS
            self.ptxtSpace = math.gcd(self.ptxtSpace, self.ksw_ptxtSpace)
            self.intFactor %= self.ptxtSpace
        kw = {"norms": True, "defer": True} if self._meas else {}
        if pend is not None:
            res = self.ops.mulRelin(*pend, W, digits, **kw)
        else:
            res = self.ops.reLinearize(self.parts["1"], self.parts.get("s"), self.parts[hnd], W, digits, sp, **kw)
        o0, o1 = res[0], res[1]

        # noise: scaled parts + key-switch added noise (src/Ctxt.cpp:746, 827-841)
        def update():
            added = -math.inf
            for k, d in enumerate(digits):
                bnd = math.log(ctx.noiseBoundForUniform(0.5, ctx.phim)) + ctx.logOfProduct(d)
                if self._meas:   # norm_val = embeddingLargestCoeff(digit) (src/DoubleCRT.cpp:538-545)
                    nb = _ln(float(max(res[2][k]))) + ctx.logOfProduct(d)
                    if timing.fhe_stats:   # (src/DoubleCRT.cpp:547-548)
                        timing.STATS_UPDATE("break-into-digits-ratio", math.exp(nb - bnd))
                else:            # high-probability bound (src/DoubleCRT.cpp:520-529)
                    nb = bnd
                added = logaddexp(added, nb + self.ksw_lnNoise)
            if timing.fhe_stats:           # (src/Ctxt.cpp:833-835): added noise over the ciphertext's own
                timing.STATS_UPDATE("KS-noise-ratio", math.exp(added - (self._ln + logProd)))
            self._ln = logaddexp(self._ln + logProd, added)

        self.parts = {"1": o0, "s": o1}
        self.primeSet = self.primeSet | frozenset(sp)
        self._defer(update)

    def _matrixFor(self, hnd):
        """PubKey::getKeySWmatrix for the part's handle; LogicError in the reference when absent"""
        sp_, xp_ = handle_powers(hnd)
        if (sp_, xp_) == (2, 1):
            W = self.ksw
        elif sp_ == 1:
            W = self.ksw_auto.get(xp_)
        elif xp_ == 1:
            W = self.ksw_pow.get(sp_)
        else:
            W = None
        if W is None:
            raise LookupError(f"no key-switching matrices for s^{sp_}(X^{xp_})")
        return W

    def _reLinearizeMany(self, other):
        """Ctxt::reLinearize with several non-canonical parts (after multiplyBy2: s^2 and s^3):
        parts 1 and s are scaled by P, every other part goes through keySwitchPart -- break into
        digits, key-switch with its own matrix, accumulate (src/Ctxt.cpp:720-842)."""
        ctx = self.context
        mats = {h: self._matrixFor(h) for h in other}
        self.dropSmallAndSpecialPrimes()
        self._relin_CKKS_adjust()
        sp = list(ctx.specialPrimes)
        logProd = ctx.logOfProduct(sp)
        self.lnRatFactor += logProd
        digits = [[i for i in d if i in self.primeSet] for d in ctx.digits]
        digits = [d for d in digits if d]
        if self.ptxtSpace > 1:
            self.ptxtSpace = math.gcd(self.ptxtSpace, self.ksw_ptxtSpace)
            self.intFactor %= self.ptxtSpace
        part0 = self.parts["1"]
        part0.addPrimesAndScale(sp)
        if "s" in self.parts:
            part1 = self.parts["s"]
            part1.addPrimesAndScale(sp)
        else:
            part1 = self.ops.zerosLike(part0)
        added = -math.inf
        for h in other:
            if self._meas:
                dg, nrm = self.ops.breakIntoDigits(self.parts[h], digits, sp, norms=True)
            else:
                dg, nrm = self.ops.breakIntoDigits(self.parts[h], digits, sp), None
            self.ops.keySwitchDigits(dg, mats[h], part0, part1)
            for k, d in enumerate(digits):
                nb = _ln(float(max(nrm[k]))) if nrm is not None else \
                    math.log(ctx.noiseBoundForUniform(0.5, ctx.phim))
                added = logaddexp(added, nb + ctx.logOfProduct(d) + self.ksw_lnNoise)
        self.lnNoise = logaddexp(self.lnNoise + logProd, added)
        self.parts = {"1": part0, "s": part1}
        self.primeSet = self.primeSet | frozenset(sp)

    @timing.timed
    def multiplyBy2(self, other1, other2):
        """Ctxt::multiplyBy2 (src/Ctxt.cpp:1776-1828): the product of three ciphertexts with ONE
        relinearisation at the end (parts up to s^3), multiplying in the order of their capacities."""
        if not self.parts:
            return
        for o in (other1, other2):
            if not o.parts:
                c = o.clone()
                self.__dict__.update(c.__dict__)
                return
        cap, cap1, cap2 = self.capacity(), other1.capacity(), other2.capacity()
        if cap < cap1 and cap < cap2:
            tmp = other1.clone()
            tmp.multLowLvl(other2)
            self.multLowLvl(tmp, destructive=True)
            self.reLinearize()
            return
        first, second = (other2, other1) if (cap < cap2 or cap1 < cap2) else (other1, other2)
        second = second.clone() if second is self else second
        self.multLowLvl(first if first is not self else first.clone())
        self.multLowLvl(second)
        self.reLinearize()

    @timing.timed
    def square(self):
        """Ctxt::square: multiplyBy(*this)"""
        self.multiplyBy(self.clone())

    def cube(self):
        """Ctxt::cube: multiplyBy2(*this, *this)"""
        self.multiplyBy2(self.clone(), self.clone())

    def power(self, e):
        """Ctxt::power (src/polyEval.cpp:392-414): repeated squaring for a power of two, otherwise
        DynamicCtxtPowers (:18-29): X^e = X^(e-k) * X^k with k the largest power of two below e, every
        power computed once -- the multiplication depth stays at ceil(log2 e)."""
        if e < 1:
            raise ValueError("Cannot raise a ctxt to a non positive exponent")
        if e == 1:
            return self
        if e & (e - 1) == 0:
            for _ in range(e.bit_length() - 1):
                self.square()
            return self
        powers = {1: self.clone()}

        def get(n):
            if n not in powers:
                k = 1 << ((n - 1).bit_length() - 1)      # NextPowerOfTwo(n) - 1
                c = get(n - k).clone()
                c.multiplyBy(get(k))
                powers[n] = c
            return powers[n]
        r = get(e)
        self.__dict__.update(r.__dict__)
        return self

    def _relin_CKKS_adjust(self):
        """Ctxt::relin_CKKS_adjust (src/Ctxt.cpp:664-717): if the noise is below what the special
        primes were sized for, scale the ciphertext (and its factor) up by an integer."""
        ctx = self.context
        if not ctx.ckks:
            return
        h = ctx.phim / 2.0 if ctx.hwt == 0 else float(ctx.hwt)
        log_phim = max(math.log(ctx.phim), 1.0)
        gamma = 8.0 * int(ctx.scale) * math.sqrt(ctx.phim * log_phim * h / 12.0)
        if math.log(gamma) > self.lnNoise:
            self._materializeTensor()
            xf = int(math.ceil(math.exp(math.log(gamma) - self.lnNoise)))
            for p in self.parts.values():
                p.mulConstant(xf)
            self.lnNoise = self.lnNoise + math.log(xf)
            self.lnRatFactor += math.log(xf)

    # ---- rotations ----
    def automorph(self, k):
        """Ctxt::automorph (src/Ctxt.cpp:2437-2457): F(X) -> F(X^k) on every part; the part that
        pointed at s now points at s(X^k).  No change in the noise bound."""
        m = self.context.m
        k %= m
        if math.gcd(k, m) != 1:
            raise ValueError("k must be in Zm*")
        if k == 1:
            return self
        assert "s2" not in self.parts, "relinearise before an automorphism"
        new = {}
        for h, p in self.parts.items():
            p.automorph(k)
            if h == "1":
                new[h] = p
            else:
                j = (1 if h == "s" else h[1]) * k % m
                new["s" if j == 1 else ("s", j)] = p
        self.parts = new
        return self

    def _firstStep(self, k):
        """PubKey::getNextKSWmatrix(k).fromKey.getPowerOfX(): k itself when there is a matrix for it
        and no map was set, otherwise the first edge of the BFS path (PubKey::setKeySwitchMap)."""
        if self.ksw_map is not None:
            amt = self.ksw_map[k]
            if amt == 0 or amt not in self.ksw_auto:
                raise LookupError(f"no key-switching matrices for k={k}")   # LogicError in the reference
            return amt
        if k not in self.ksw_auto:
            raise LookupError(f"no key-switching matrices for k={k}")
        return k

    @timing.timed
    def smartAutomorph(self, k):
        """Ctxt::smartAutomorph (src/Ctxt.cpp:2462-2515): re-linearise, then walk the path of
        available matrices -- automorph(amt), reLinearize, k <- k * amt^-1 -- until k = 1."""
        m = self.context.m
        k %= m
        if k == 1 or not self.parts:
            return self
        if math.gcd(k, m) != 1:
            raise ValueError("k must be in Zm*")
        self._firstStep(k)          # isReachable, before anything is touched
        self.reLinearize()          # canonical form first
        while k != 1:
            amt = self._firstStep(k)
            self.automorph(amt)
            self.reLinearize()
            k = k * pow(amt, -1, m) % m
        return self

    @timing.timed
    def multiplyBy(self, other):
        self.multLowLvl(other, lazyTensor=True)
        self.reLinearize()
        self._materializeTensor()
        return self

    # ---- plaintext constants ----
    def multByConstant(self, dcrt, size=-1.0):
        """Ctxt::multByConstant(const DoubleCRT&, double size) (src/Ctxt.cpp:1832-1856), BGV: every
        part times the constant (Mul with matchIndexSets=false: dcrt may live on more primes);
        size < 0: the bound for coefficients uniform in [-ptxtSpace/2, ptxtSpace/2]."""
        if not self.parts:
            return self
        if size < 0.0:
            size = self.context.noiseBoundForMod(self.ptxtSpace, self.context.phim)
        for p in self.parts.values():
            p *= dcrt
        self.lnNoise = self.lnNoise + _ln(size)
        return self

    def multByScalar(self, c):
        """Ctxt::multByConstant(const ZZ& / long / double / xdouble) (src/Ctxt.cpp:2033-2110).
        BGV: c mod ptxtSpace = c1 * d with d = gcd(c, ptxtSpace); the ciphertext is multiplied by the
        balanced d only and the unit c1 goes into intFactor (its inverse).  CKKS: no polynomial work
        at all -- ptxtMag *= |c|, ratFactor /= |c|, a sign flips the parts."""
        if not self.parts:
            return self
        if self.context.ckks:
            c = float(c)
            if c == 1.0:
                return self
            if c == 0.0:
                self.parts = {}
                return self
            self.ptxtMag *= abs(c)
            self.lnRatFactor -= math.log(abs(c))
            if c < 0:
                self.negate()
            return self
        if isinstance(c, float):
            raise TypeError("multByConstant(double) not supported for BGV")     # LogicError in the reference
        P = self.ptxtSpace
        c0 = int(c) % P
        if c0 == 1:
            return self
        if c0 == 0:
            self.parts = {}
            return self
        d = math.gcd(c0, P)
        self.intFactor = self.intFactor * pow(c0 // d, -1, P) % P
        if d == 1:
            return self
        cc = d - P if d > P // 2 else d
        self.lnNoise = self.lnNoise + math.log(abs(cc))
        for p in self.parts.values():
            p.mulConstant(cc)
        return self

    def __iadd__(self, other):       # Ctxt::operator+=
        self.addCtxt(other)
        return self

    def __isub__(self, other):       # Ctxt::operator-=
        self.addCtxt(other, negative=True)
        return self

    def __imul__(self, other):       # Ctxt::operator*= (ciphertext): multiplyBy
        self.multiplyBy(other)
        return self

    def multByConstantCKKS(self, dcrt, size, factor, roundingErr):
        """Ctxt::multByConstantCKKS(const DoubleCRT&, size, factor, roundingErr) (src/Ctxt.cpp:1905-
        1938): dcrt encodes a constant of magnitude <= size scaled by `factor` with encoding error
        <= roundingErr (the reference takes absent values from EncryptedArrayCx, which is out of
        scope here: they are arguments).  noise' = noise*factor*size + roundingErr*ratFactor*ptxtMag +
        noise*roundingErr; ptxtMag *= size; ratFactor *= factor."""
        if not self.parts:
            return self
        if size <= 0:
            size = 1.0
        if factor <= 0 or roundingErr < 0:
            raise ValueError("factor and roundingErr are the encoder's: pass them")
        n = self.lnNoise
        self.lnNoise = logaddexp(logaddexp(n + math.log(factor) + math.log(size),
                                           _ln(roundingErr) + self.lnRatFactor + _ln(self.ptxtMag)),
                                 n + _ln(roundingErr))
        self.ptxtMag *= size
        self.lnRatFactor += math.log(factor)
        for p in self.parts.values():
            p *= dcrt
        return self

    def addConstantCKKS(self, dcrt, size, factor):
        """Ctxt::addConstantCKKS(const DoubleCRT&, size, factor) (src/Ctxt.cpp:951-1045): the constant
        (scaled by `factor`) is multiplied by round(ratFactor / factor) and added to the part of 1;
        ptxtMag += size, noiseBound += 0.5.  The reference mod-switches up (addSomePrimes) when the
        rounded ratio is off by more than 2^-precision; here that case is an error."""
        if size <= 0:
            size = 1.0
        if factor <= 0:
            raise ValueError("factor is the encoder's: pass it")
        x = math.exp(self.lnRatFactor - math.log(factor))
        ratio = int(math.floor(x + 0.5))
        if ratio < 1 or abs(ratio / x - 1.0) * (1 << self.context.r) > 1.0:
            raise RuntimeError("addConstantCKKS: ratFactor / factor is too far from an integer "
                               "(the reference would call addSomePrimes here)")
        self.ptxtMag += size
        self.lnNoise = logaddexp(self.lnNoise, math.log(0.5))
        if "1" not in self.parts:
            raise RuntimeError("Ctxt::addPart: no part pointing at 1")
        if ratio == 1:
            self.parts["1"] += dcrt
        else:
            tmp = dcrt.copy()
            tmp.mulConstant(ratio)
            self.parts["1"] += tmp
        return self

    def addConstant(self, dcrt, size=-1.0):
        """Ctxt::addConstant(const DoubleCRT&, double size) (src/Ctxt.cpp:896-935), BGV: the
        constant is scaled by f = balRem(intFactor * Q mod ptxtSpace) and added to the part of 1."""
        ctx = self.context
        if size < 0.0:
            size = ctx.noiseBoundForMod(self.ptxtSpace, ctx.phim)
        f = 1
        if self.ptxtSpace > 2:
            p = self.ptxtSpace
            f = ctx.productOfPrimes(self.primeSet) % p * self.intFactor % p
            if f > p // 2:                      # balRem: into (-p/2, p/2]
                f -= p
        self.lnNoise = logaddexp(self.lnNoise, _ln(size * abs(f)))
        if "1" not in self.parts:
            raise RuntimeError("Ctxt::addPart: no part pointing at 1")   # (an empty ctxt in the reference)
        if f == 1:
            self.parts["1"] += dcrt
        else:
            tmp = dcrt.copy()
            tmp.mulConstant(f)
            self.parts["1"] += tmp
        return self

    def cleanUp(self):
        """Ctxt::cleanUp (src/Ctxt.cpp:788-797)"""
        self.reLinearize()
        ctx = self.context
        if self.primeSet & (frozenset(ctx.specialPrimes) | frozenset(ctx.smallPrimes)):
            self.dropSmallAndSpecialPrimes()
        return self


class BasicAutomorphPrecon:
    """Hoisting (src/matmul.cpp:48-184): break the `s` part of a ciphertext into digits ONCE, then
    every automorphism rotates the digits (a permutation of evaluation rows, hx_automorph on the
    whole digit block) and key-switches them with the matrix of that automorphism -- no inverse
    transform, no basis extension and no forward transforms per rotation.  As in the reference the
    matrix for k itself must be available (the generator-tree walk of PubKey::getNextKSWmatrix is
    key management and stays with the caller)."""

    def __init__(self, ct):
        self.ctxt = ct.clone()
        self.polyDigits, self.lnNoise = None, 0.0
        c = self.ctxt
        if len(c.parts) <= 1:
            return
        c.cleanUp()
        assert set(c.parts) == {"1", "s"}, "Ciphertext is not in canonical form"
        ctx = c.context
        sp = list(ctx.specialPrimes)
        digits = [[i for i in d if i in c.primeSet] for d in ctx.digits]
        self.digits = [d for d in digits if d]
        if c._meas:
            self.polyDigits, nrm = c.ops.breakIntoDigits(c.parts["s"], self.digits, sp, norms=True)
        else:
            self.polyDigits, nrm = c.ops.breakIntoDigits(c.parts["s"], self.digits, sp), None
        # addedNoise = breakIntoDigits' return value * max over the matrices' noise bounds
        # (src/matmul.cpp:91-97); noise = ctxt.noise * P + addedNoise (:99-112)
        added = -math.inf
        for k, d in enumerate(self.digits):
            nb = _ln(float(max(nrm[k]))) if nrm is not None else math.log(ctx.noiseBoundForUniform(0.5, ctx.phim))
            added = logaddexp(added, nb + ctx.logOfProduct(d))
        added += c.ksw_lnNoise
        self.lnNoise = logaddexp(c.lnNoise + ctx.logOfProduct(sp), added)

    def automorph(self, k):
        c = self.ctxt
        ctx = c.context
        k %= ctx.m
        if k == 1 or not c.parts:
            return c.clone()
        if math.gcd(k, ctx.m) != 1:
            raise ValueError("k must be in Zm*")
        sp = list(ctx.specialPrimes)
        res = Ctxt(ctx, c.ops, c.ksw, c.ksw_ptxtSpace, c.ksw_lnNoise)
        res.ksw_auto = c.ksw_auto
        res.ptxtSpace, res.intFactor = c.ptxtSpace, c.intFactor
        res.primeSet = c.primeSet | frozenset(sp)
        part0 = c.parts["1"].copy()
        part0.automorph(k)
        part0.addPrimesAndScale(sp)
        res.ksw_map = c.ksw_map
        if len(c.parts) == 1:        # only the constant part: nothing to key-switch (:145-151)
            res.parts = {"1": part0}
            res.lnNoise = c.lnNoise + ctx.logOfProduct(sp)
            return res
        amt = c._firstStep(k)          # first key-switching matrix on the way to k (:153-162)
        W = c.ksw_auto[amt]
        if amt != k:                   # the constant part was rotated by k above: redo it by amt
            part0 = c.parts["1"].copy()
            part0.automorph(amt)
            part0.addPrimesAndScale(sp)
        dg = self.polyDigits.copy()
        dg.automorph(amt)
        part1 = c.ops.zerosLike(part0)
        c.ops.keySwitchDigits(dg, W, part0, part1)
        res.parts = {"1": part0, "s": part1}
        res.lnNoise = self.lnNoise
        res.ksw_map = c.ksw_map
        if amt != k:                   # more automorphisms to do: the usual smartAutomorph (:177-181)
            res.smartAutomorph(k * pow(amt, -1, ctx.m) % ctx.m)
        return res


# ---------------------------------------------------------------------------------------------
# products of many ciphertexts (src/Ctxt.cpp:2803-2904)
# ---------------------------------------------------------------------------------------------
def _split(n):
    """n1 = the highest power of two below n (n/2 <= n1 < n): NumBits(n - 1) = l, n1 = 2^(l-1)"""
    return 1 << ((n - 1).bit_length() - 1)


def incrementalProduct(v):
    """For i = n-1 .. 0: v[i] = prod_{j <= i} v[j], depth log n and (n log n)/2 products, in place."""
    def rec(lo, n):
        if n <= 1:
            return
        n1 = _split(n)
        rec(lo, n1)
        rec(lo + n1, n - n1)
        for i in range(lo + n1, lo + n):
            v[i].multiplyBy(v[lo + n1 - 1])
    rec(0, len(v))


def totalProduct(v):
    """prod_i v[i] in depth log n with n-1 products (triples through multiplyBy2); a new Ctxt."""
    def rec(lo, n):
        out = v[lo].clone()
        if n == 2:
            out.multiplyBy(v[lo + 1])
        elif n == 3:
            out.multiplyBy2(v[lo + 1], v[lo + 2])
        elif n > 3:
            n1 = _split(n)
            out = rec(lo, n1)
            out.multiplyBy(rec(lo + n1, n - n1))
        return out
    if not v:
        raise ValueError("totalProduct of an empty vector")
    return rec(0, len(v))


def innerProduct(v1, v2):
    """sum_i v1[i] * v2[i] with the low-level product and ONE relinearisation at the end"""
    n = min(len(v1), len(v2))
    if n <= 0:
        raise ValueError("innerProduct of empty vectors")
    result = v1[0].clone()
    result.multLowLvl(v2[0])
    for i in range(1, n):
        tmp = v1[i].clone()
        tmp.multLowLvl(v2[i])
        result.addCtxt(tmp)
    result.reLinearize()
    return result
