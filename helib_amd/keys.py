"""Host-side mirror of the reference's key material and encryption for the BGV path (SURVEY.md
row N2): every polynomial operation is a DoubleCRT operation of the engine (rows resident on the
GPU through helib_amd.capi); the samplers and the control flow stay on the host, as in the
reference.

  Sampler                 -- src/sample.cpp: sampleSmall :67-112, sampleHWt :29-65, sampleGaussian
                             :114-199 and their *Bounded forms (:269-304, :342-396, :459-509:
                             redraw until embeddingLargestCoeff <= the high-probability bound)
  RLWE1 / RLWE            -- src/keys.cpp:39-85     c0 = p*e - c1*s
  SecKey.GenSecKey        -- src/keys.cpp:1139-1157 (ImportSecKey :1099-1137: public encryption
                             key = one RLWE instance over the ctxt primes, s^e -> s matrices)
  SecKey.GenKeySWmatrix   -- src/keys.cpp:1161-1255 b_i = p*e_i - a_i*s' + P*B_i*s^r(X^t),
                             P = prod(special primes), B_i = prod(digits before i)
  PubKey.Encrypt          -- src/keys.cpp:358-488   r*pk + p*(e0,e1) + (ptxt*Q mod p balanced, 0)
  SecKey.skEncrypt        -- src/keys.cpp:1425-1520 (BGV branch)
  SecKey.Decrypt          -- src/keys.cpp:1327-1420 sum_parts part*s^r(X^t), toPoly (centred CRT,
                             src/DoubleCRT.cpp:925-1113), PolyRed mod ptxtSpace, times
                             (intFactor*Q)^-1 mod p

The random streams are numpy's, not NTL's (the distributions are what matters; NTL's
PRG is not available to pin a stream against).  The backend object supplies the polynomial side:

  be.fromCoeffs(idx, coeffs)     DoubleCRT = zzX / ZZX  (reduce mod each prime, forward transform)
  be.randomize(idx, rng)         DoubleCRT::randomize   (uniform rows)
  be.toPoly(poly)                DoubleCRT::toPoly      (centred big integers)
  be.toPolyMod(poly, t)          toPoly + PolyRed(t)    (hx_poly_rem: exact, no big integers)
  be.embeddingLargestCoeff(f)    norms.cpp              (canonical-embedding l-infinity norm)
  be.keySwitch(row_idx, b, a)    KeySwitch storage for Ctxt.reLinearize / smartAutomorph
  be.ops                         the tensorProduct / reLinearize entry points (helib_amd.capi)

`HxBackend` below is the GPU one.  tests/ drive the same classes over the CPU oracle's backend and
compare bit for bit.
"""
import math

import numpy as np

from . import ctxt as hc
from . import hostnt


# ---------------------------------------------------------------------------------------------
# samplers (host, as in the reference)
# ---------------------------------------------------------------------------------------------
class Sampler:
    """seed=None (the default): every draw comes from a ChaCha20 stream keyed with 256 bits of OS
    entropy, as the reference seeds NTL's PRG.  seed=<int>: a key derived from the seed --
    DETERMINISTIC, for tests and benchmarks only (helib_amd.prg)."""

    def __init__(self, context, backend, seed=None):
        from .prg import ChaChaRng
        self.cc, self.be = context, backend
        self.rng = seed if isinstance(seed, ChaChaRng) else ChaChaRng(seed)

    # -- unbounded draws: power-of-two m takes n = phi(m) coefficients directly; general m samples m
    #    coefficients and reduces modulo Phi_m (src/sample.cpp:240-256, 321-341, 420-440) --
    def _n(self):
        return self.cc.phim if self.cc.pow2 else self.cc.m

    def _reduce(self, poly):
        """reduceModPhimX (src/sample.cpp:216-226): remainder modulo the monic Phi_m(X)"""
        if self.cc.pow2:
            return poly
        if not hasattr(self, "_phimx"):
            from . import hostnt
            self._phimx = np.array(hostnt.phimx(self.cc.m), dtype=np.int64)
        phi, n = self._phimx, self.cc.phim
        a = poly.astype(np.int64).copy()
        for i in range(len(a) - 1, n - 1, -1):
            c = a[i]
            if c:
                a[i - n:i + 1] -= c * phi
        return a[:n]

    def sampleSmall(self, prob=None):
        """each coefficient 0 with probability 1-prob, +-1 with probability prob/2 each (prob = 1/2 for
        power-of-two m, phi(m)/(2m) otherwise: src/sample.cpp:327-339)"""
        if prob is None:
            prob = 0.5 if self.cc.pow2 else self.cc.phim / (2.0 * self.cc.m)
        n = self._n()
        nz = self.rng.random(n) < prob
        sign = self.rng.integers(0, 2, size=n) * 2 - 1
        return self._reduce((nz * sign).astype(np.int64))

    def sampleHWt(self, hwt):
        n = self._n()
        hwt = min(hwt, n)
        out = np.zeros(n, dtype=np.int64)
        pos = self.rng.choice(n, size=hwt, replace=False)
        out[pos] = self.rng.integers(0, 2, size=hwt) * 2 - 1
        return self._reduce(out)

    def sampleGaussian(self, stdev):
        return self._reduce(np.rint(self.rng.normal(0.0, stdev, size=self._n())).astype(np.int64))

    def _bounded(self, draw, bound, what):
        for _ in range(1000):           # "while (++count < 1000 && val > bound)"
            f = draw()
            if self.be.embeddingLargestCoeff(f) <= bound:
                return f, bound
        raise RuntimeError(f"Error: {what}, after 1000 trials, still val > bound={bound}")

    def sampleSmallBounded(self):
        n = self.cc.phim
        return self._bounded(self.sampleSmall, math.sqrt(n * math.log(n) / 2.0), "sampleSmallBounded")

    def sampleHWtBounded(self, hwt):
        bound = math.sqrt(hwt * math.log(self.cc.phim))   # sampleHWtBoundedEffectiveBound
        return self._bounded(lambda: self.sampleHWt(hwt), bound, "sampleHWtBounded")

    def sampleGaussianBounded(self, stdev=0.0):
        stdev = stdev or self.cc.stdev
        eff = math.sqrt(self.cc.phim * math.log(self.cc.phim)) if self.cc.pow2 else \
            math.sqrt(self.cc.m * math.log(self.cc.phim))
        return self._bounded(lambda: self.sampleGaussian(stdev), stdev * eff, "sampleGaussianBounded")


def RLWE1(be, sampler, idx, c1, s_coeffs, p):
    """c0 = p*e - c1*s over the primes idx (those of c1); returns (c0, bound)."""
    cc = sampler.cc
    stdev = cc.stdev if cc.pow2 else cc.stdev * math.sqrt(cc.m)
    e, bound = sampler.sampleGaussianBounded(stdev)
    c0 = be.fromCoeffs(idx, e)
    if p > 1:
        c0.mulConstant(p)
        bound *= p
    tmp = c1.copy()
    tmp *= be.fromCoeffs(idx, s_coeffs)   # Mul(s, matchIndexSets=false): s restricted to c1's primes
    c0 -= tmp
    return c0, bound


# ---------------------------------------------------------------------------------------------
# keys
# ---------------------------------------------------------------------------------------------
class KeySwitchInfo:
    """include/helib/keySwitching.h:86-101 bookkeeping around the backend's storage."""

    def __init__(self, fromSPower, fromXPower, W, ptxtSpace, noiseBound, b=None, a=None):
        self.fromSPower, self.fromXPower = fromSPower, fromXPower
        self.W, self.ptxtSpace, self.noiseBound = W, ptxtSpace, noiseBound
        self.b, self.a = b, a             # host copies of the rows [D][rows][N] (wire format, tests)


class PubKey:
    def __init__(self, context, backend, seed=None):
        self.cc, self.be = context, backend
        self.sampler = Sampler(context, backend, seed)
        self.pubEncrKey = None            # (part "1", part "s") over the ctxt primes
        self.pubEncrKeyNoise = 0.0
        self.ptxtSpace = context.ptxtSpace
        self.skBounds = []
        self.keySwitching = {}            # (fromSPower, fromXPower) -> KeySwitchInfo

    # -- PubKey::getKeySWmatrix --
    def haveKeySWmatrix(self, fromSPower, fromXPower):
        return (fromSPower, fromXPower) in self.keySwitching

    def getKeySWmatrix(self, fromSPower, fromXPower):
        return self.keySwitching[(fromSPower, fromXPower)]

    def getSKeyBound(self):
        return self.skBounds[0]

    # -- PubKey::setKeySwitchMap / isReachable / getNextKSWmatrix (src/keys.cpp:122-172, 226-250) --
    def setKeySwitchMap(self):
        """BFS over Zm* from 1 along the available matrices W[s(X^n) -> s]: keySwitchMap[k] = the n
        of the matrix to use as the FIRST step of the automorphism X -> X^k (0: unreachable)."""
        m = self.cc.m
        edges = [xp for (sp, xp) in self.keySwitching if sp == 1 and xp > 1]   # insertion order, as the reference's list
        kmap = [0] * m
        queue, head = [1], 0
        while head < len(queue):
            cur = queue[head]
            head += 1
            for n in edges:
                nxt = cur * n % m
                if kmap[nxt] == 0 and nxt != 1:
                    kmap[nxt] = n
                    queue.append(nxt)
        self.keySwitchMap = kmap
        return kmap

    def isReachable(self, k):
        km = getattr(self, "keySwitchMap", None)
        k %= self.cc.m
        return k == 1 or (km is not None and km[k] != 0)

    def getNextKSWmatrix(self, k):
        return self.keySwitching[(1, self.keySwitchMap[k % self.cc.m])]

    def _newCtxt(self, c0, c1, noiseBound, ptxtSpace):
        relin = self.keySwitching.get((2, 1))
        ct = hc.Ctxt(self.cc, self.be.ops, relin.W if relin else None,
                     relin.ptxtSpace if relin else None,
                     math.log(relin.noiseBound) if relin else None)
        ct.parts = {"1": c0, "s": c1}
        ct.primeSet = frozenset(self.cc.ctxtPrimes)
        ct.ptxtSpace, ct.intFactor = ptxtSpace, 1
        ct.lnNoise = math.log(noiseBound)
        for (sp, xp), ks in self.keySwitching.items():
            if sp == 1 and xp > 1:
                ct.ksw_auto[xp] = ks.W
            elif sp > 2 and xp == 1:
                ct.ksw_pow[sp] = ks.W
        ct.ksw_map = getattr(self, "keySwitchMap", None)
        return ct

    def CKKSencrypt(self, ptxt, ptxtSize=1.0, scaling=0.0):
        """PubKey::CKKSencrypt (src/keys.cpp:501-581): ptxt is an integer polynomial already scaled
        by `scaling`; ctxt = r*pk + (e0, e1) + (ef*ptxt, 0) with ef = ceil(error_bound*prec /
        (scaling*ptxtSize)), prec = 2^precision.  Decrypts to ptxt*ef + noise; ratFactor =
        scaling*ef."""
        cc, be = self.cc, self.be
        if not cc.ckks:
            raise RuntimeError("CKKSencrypt on a BGV context")
        if ptxtSize <= 0:
            ptxtSize = 1.0
        prec = 1 << cc.r
        if scaling <= 0:
            scaling = float(prec) / ptxtSize
        idx = list(cc.ctxtPrimes)
        parts = [self.pubEncrKey[0].copy(), self.pubEncrKey[1].copy()]
        r, r_bound = self.sampler.sampleSmallBounded()
        rr = be.fromCoeffs(idx, r)
        error_bound = r_bound * self.pubEncrKeyNoise
        stdev = cc.stdev if cc.pow2 else cc.stdev * math.sqrt(cc.m)
        for i in range(2):
            parts[i] *= rr
            e, e_bound = self.sampler.sampleGaussianBounded(stdev)
            parts[i] += be.fromCoeffs(idx, e)
            if i == 1:
                e_bound *= self.getSKeyBound()
            error_bound += e_bound
        ef = int(math.ceil(error_bound * prec / (scaling * ptxtSize)))
        coeffs = np.zeros(cc.phim, dtype=object)
        for i, v in enumerate(ptxt):
            coeffs[i] = int(v) * ef if ef > 1 else int(v)
        if ef > 1:
            scaling *= ef
        parts[0] += be.fromCoeffs(idx, coeffs)
        ct = self._newCtxt(parts[0], parts[1], error_bound, 1)
        # EncryptedArrayCx::roundedSize: the next power of two, so as not to leak the size
        ct.ptxtMag = 1.0 if ptxtSize <= 1 else float(1 << (int(math.ceil(ptxtSize)) - 1).bit_length())
        ct.lnRatFactor = math.log(scaling)
        return ct

    def _ptxt_fixed(self, ptxt, primeSet, ptxtSpace):
        """balanced_MulMod(ptxt, Q mod p, p) (src/NumbTh.cpp:876-891; the coin for c == p/2 at even
        p is the sampler's)."""
        QmodP = self.cc.productOfPrimes(primeSet) % ptxtSpace
        out = np.zeros(self.cc.phim, dtype=np.int64)
        for i, v in enumerate(ptxt):
            c = (int(v) % ptxtSpace) * QmodP % ptxtSpace
            if c > ptxtSpace // 2 or (ptxtSpace % 2 == 0 and c == ptxtSpace // 2
                                      and self.sampler.rng.integers(0, 2)):
                c -= ptxtSpace
            out[i] = c
        return out

    def Encrypt(self, ptxt, ptxtSpace=0):
        """PubKey::Encrypt (BGV): returns a helib_amd.ctxt.Ctxt over the ctxt primes."""
        cc, be = self.cc, self.be
        if self.pubEncrKey is None:
            raise RuntimeError("no public encryption key")
        ptxtSpace = ptxtSpace or self.ptxtSpace
        if ptxtSpace != self.ptxtSpace:
            ptxtSpace = math.gcd(ptxtSpace, self.ptxtSpace)
            if ptxtSpace <= 1:
                raise RuntimeError("Plaintext-space mismatch on encryption")
        idx = list(cc.ctxtPrimes)
        parts = [self.pubEncrKey[0].copy(), self.pubEncrKey[1].copy()]
        r, r_bound = self.sampler.sampleSmallBounded()
        rr = be.fromCoeffs(idx, r)
        noise = r_bound * self.pubEncrKeyNoise
        stdev = cc.stdev if cc.pow2 else cc.stdev * math.sqrt(cc.m)
        for i in range(2):
            parts[i] *= rr
            e, e_bound = self.sampler.sampleGaussianBounded(stdev)
            ee = be.fromCoeffs(idx, e)
            ee.mulConstant(ptxtSpace)
            e_bound *= ptxtSpace
            if i == 1:
                e_bound *= self.getSKeyBound()
            parts[i] += ee
            noise += e_bound
        parts[0] += be.fromCoeffs(idx, self._ptxt_fixed(ptxt, idx, ptxtSpace))
        noise += cc.noiseBoundForMod(ptxtSpace, cc.phim)
        return self._newCtxt(parts[0], parts[1], noise, ptxtSpace)


class SecKey(PubKey):
    def __init__(self, context, backend, seed=None):
        super().__init__(context, backend, seed)
        self.sKeys = []                   # secret keys as small coefficient vectors (the DoubleCRT
                                          # over any prime set is be.fromCoeffs(set, coeffs))

    def GenSecKey(self, ptxtSpace=0, maxDegKswitch=3):
        if self.cc.hwt > 0:
            s, bound = self.sampler.sampleHWtBounded(self.cc.hwt)
        else:
            s, bound = self.sampler.sampleSmallBounded()
        return self.ImportSecKey(s, bound, ptxtSpace, maxDegKswitch)

    def ImportSecKey(self, s_coeffs, bound, ptxtSpace=0, maxDegKswitch=3):
        cc, be = self.cc, self.be
        s_coeffs = np.asarray(s_coeffs, dtype=np.int64)
        if not self.sKeys:
            if ptxtSpace < 2:
                ptxtSpace = cc.ptxtSpace
            idx = list(cc.ctxtPrimes)
            c1 = be.randomize(idx, self.sampler.rng)
            c0, nb = RLWE1(be, self.sampler, idx, c1, s_coeffs, ptxtSpace)
            self.pubEncrKey, self.pubEncrKeyNoise, self.ptxtSpace = (c0, c1), nb, ptxtSpace
        self.skBounds.append(bound)
        self.sKeys.append(s_coeffs)
        keyID = len(self.sKeys) - 1
        for e in range(2, maxDegKswitch + 1):
            self.GenKeySWmatrix(e, 1, keyID, keyID)
        return keyID

    def _keyRows(self, idx, sPower=1, xPower=1, keyID=0):
        key = self.be.fromCoeffs(idx, self.sKeys[keyID])
        if xPower > 1:
            key.automorph(xPower)         # s(X^t)
        if sPower > 1:
            key.Exp(sPower)               # s^r(X^t), computed modulo every prime (:1189-1192)
        return key

    def GenKeySWmatrix(self, fromSPower, fromXPower, fromIdx=0, toIdx=0, p=0):
        cc, be = self.cc, self.be
        if fromSPower <= 0 or fromXPower <= 0:
            return None
        if fromSPower == 1 and fromXPower == 1 and fromIdx == toIdx:
            return None
        if self.haveKeySWmatrix(fromSPower, fromXPower):
            return self.keySwitching[(fromSPower, fromXPower)]
        idx = list(cc.ctxtPrimes) + list(cc.specialPrimes)
        fromKey = self._keyRows(idx, fromSPower, fromXPower, fromIdx)
        n = len(cc.digits)
        a = [be.randomize(idx, self.sampler.rng) for _ in range(n)]
        if p < 2:
            p = self.ptxtSpace
        b, noise = [], 0.0
        for i in range(n):
            bi, noise = RLWE1(be, self.sampler, idx, a[i], self.sKeys[toIdx], p)
            b.append(bi)
        fromKey.mulConstant(cc.productOfPrimes(cc.specialPrimes))
        for i in range(n):
            b[i] += fromKey
            fromKey.mulConstant(cc.productOfPrimes(cc.digits[i]))
        hb = np.stack([x.download()[:, 0] for x in b])
        ha = np.stack([x.download()[:, 0] for x in a])
        ks = KeySwitchInfo(fromSPower, fromXPower, be.keySwitch(idx, hb, ha), p, noise, hb, ha)
        self.keySwitching[(fromSPower, fromXPower)] = ks
        return ks

    def skEncrypt(self, ptxt, ptxtSpace=0, skIdx=0):
        cc, be = self.cc, self.be
        if ptxtSpace < 2:
            ptxtSpace = self.ptxtSpace
        idx = list(cc.ctxtPrimes)
        c1 = be.randomize(idx, self.sampler.rng)
        c0, noise = RLWE1(be, self.sampler, idx, c1, self.sKeys[skIdx], ptxtSpace)
        c0 += be.fromCoeffs(idx, self._ptxt_fixed(ptxt, idx, ptxtSpace))
        noise += cc.noiseBoundForMod(ptxtSpace, cc.phim)
        return self._newCtxt(c0, c1, noise, ptxtSpace)

    def Decrypt(self, ct, raw=False):
        """SecKey::Decrypt: plaintext coefficients in [0, ptxtSpace) (raw=True: the centred
        integers f before the modular reduction)."""
        cc, be = self.cc, self.be
        idx = sorted(ct.primeSet)
        acc = None
        for handle, part in ct.parts.items():
            if handle == "1":
                term = part.copy()
            else:
                sPower, xPower = hc.handle_powers(handle)
                term = self._keyRows(part.getIndexSet(), sPower, xPower)
                term *= part
            if acc is None:
                acc = term
            else:
                acc += term
        if raw or cc.ckks:                # "if (isCKKS()) return;" -- the caller divides by ratFactor
            return be.toPoly(acc)
        p = ct.ptxtSpace
        out = be.toPolyMod(acc, p)        # toPoly + PolyRed(p, abs=true), on the device
        if p > 2:
            factor = cc.productOfPrimes(idx) % p * ct.intFactor % p
            if factor != 1:
                inv = pow(factor, -1, p)
                out = [v * inv % p for v in out]
        return out


# ---------------------------------------------------------------------------------------------
# Families of key-switching matrices (src/keySwitching.cpp:297-700)
# ---------------------------------------------------------------------------------------------
HELIB_KSS_UNKNOWN, HELIB_KSS_FULL, HELIB_KSS_BSGS, HELIB_KSS_MIN = 0, 1, 2, 3
HELIB_KEYSWITCH_THRESH, HELIB_KEYSWITCH_MIN_THRESH = 50, 8
LONG_MAX = (1 << 63) - 1


def KSGiantStepSize(D):
    if D <= 0:
        raise ValueError("Step size must be positive")
    g = math.isqrt(D)
    return g + 1 if g * g < D else g


def _zmstar(sk):
    z = getattr(sk, "zMStar", None)
    if z is None:
        z = sk.zMStar = hostnt.ZmStar(sk.cc.m, sk.cc.p)     # CKKS: p = -1, the quotient by <-1>
    return z


def _setKSStrategy(sk, dim, val):
    ks = sk.__dict__.setdefault("KS_strategy", [])
    while len(ks) <= dim + 1:
        ks.append(HELIB_KSS_UNKNOWN)
    ks[dim + 1] = val


def getKSStrategy(sk, dim):
    ks = getattr(sk, "KS_strategy", [])
    return ks[dim + 1] if 0 <= dim + 1 < len(ks) else HELIB_KSS_UNKNOWN


def _dim(z, i):
    return (z.ordP, True) if i == -1 else (z.OrderOf(i), z.SameOrd(i))


def _add1Dmats4dim(sk, i, keyID):
    z = _zmstar(sk)
    ord_, native = _dim(z, i)
    for j in range(1, ord_):
        sk.GenKeySWmatrix(1, z.genToPow(i, j), keyID, keyID)
    if not native:
        sk.GenKeySWmatrix(1, z.genToPow(i, -ord_), keyID, keyID)
    _setKSStrategy(sk, i, HELIB_KSS_FULL)


def _addSome1Dmats4dim(sk, i, keyID):
    z = _zmstar(sk)
    ord_, native = _dim(z, i)
    g = KSGiantStepSize(ord_)
    for j in range(1, g):                     # baby steps
        sk.GenKeySWmatrix(1, z.genToPow(i, j), keyID, keyID)
    for j in range(g, ord_, g):               # giant steps
        sk.GenKeySWmatrix(1, z.genToPow(i, j), keyID, keyID)
    if not native:
        sk.GenKeySWmatrix(1, z.genToPow(i, -ord_), keyID, keyID)
    _setKSStrategy(sk, i, HELIB_KSS_BSGS)


def _addMinimal1Dmats4dim(sk, i, keyID):
    z = _zmstar(sk)
    ord_, native = _dim(z, i)
    sk.GenKeySWmatrix(1, z.genToPow(i, 1), keyID, keyID)
    if not native:
        sk.GenKeySWmatrix(1, z.genToPow(i, -ord_), keyID, keyID)
    if ord_ > HELIB_KEYSWITCH_MIN_THRESH:
        sk.GenKeySWmatrix(1, z.genToPow(i, KSGiantStepSize(ord_)), keyID, keyID)
    _setKSStrategy(sk, i, HELIB_KSS_MIN)


def addSome1DMatrices(sk, bound=HELIB_KEYSWITCH_THRESH, keyID=0):
    """all powers for generators of order <= bound, baby/giant steps for the others"""
    z = _zmstar(sk)
    for i in range(z.numOfGens()):
        if bound >= z.OrderOf(i):
            _add1Dmats4dim(sk, i, keyID)
        else:
            _addSome1Dmats4dim(sk, i, keyID)
    sk.setKeySwitchMap()


def add1DMatrices(sk, keyID=0):
    addSome1DMatrices(sk, LONG_MAX, keyID)


def addBSGS1DMatrices(sk, keyID=0):
    addSome1DMatrices(sk, 0, keyID)


def addSomeFrbMatrices(sk, bound=HELIB_KEYSWITCH_THRESH, keyID=0):
    if bound >= _zmstar(sk).ordP:
        _add1Dmats4dim(sk, -1, keyID)
    else:
        _addSome1Dmats4dim(sk, -1, keyID)
    sk.setKeySwitchMap()


def addFrbMatrices(sk, keyID=0):
    addSomeFrbMatrices(sk, LONG_MAX, keyID)


def addBSGSFrbMatrices(sk, keyID=0):
    addSomeFrbMatrices(sk, 0, keyID)


def addMinimal1DMatrices(sk, keyID=0):
    for i in range(_zmstar(sk).numOfGens()):
        _addMinimal1Dmats4dim(sk, i, keyID)
    sk.setKeySwitchMap()


def addMinimalFrbMatrices(sk, keyID=0):
    _addMinimal1Dmats4dim(sk, -1, keyID)
    sk.setKeySwitchMap()


def addAllMatrices(sk, keyID=0):
    m = sk.cc.m
    for i in range(m):
        if math.gcd(i, m) == 1:
            sk.GenKeySWmatrix(1, i, keyID, keyID)
    sk.setKeySwitchMap()


def addTheseMatrices(sk, automVals, keyID=0):
    for k in sorted(set(automVals)):
        sk.GenKeySWmatrix(1, k, keyID, keyID)
    sk.setKeySwitchMap()


# ---------------------------------------------------------------------------------------------
# GPU backend
# ---------------------------------------------------------------------------------------------
def crt_centred(primes, rows):
    """DoubleCRT::toPoly's CRT (src/DoubleCRT.cpp:992-1112): rows[i][j] = value_j mod primes[i]
    -> centred integers in (-Q/2, Q/2]."""
    Q = 1
    for q in primes:
        Q *= q
    acc = np.zeros(rows.shape[1], dtype=object)
    for q, row in zip(primes, rows):
        Qi = Q // q
        c = Qi * pow(Qi % q, -1, q)
        acc = (acc + row.astype(object) * c) % Q
    half = Q // 2
    return [int(v) - Q if int(v) > half else int(v) for v in acc]


class HxBackend:
    def __init__(self, hxctx, context):
        from . import capi
        self.ops, self.hx, self.gctx, self.cc = capi, capi, hxctx, context

    def _rows(self, idx, coeffs):
        primes = self.cc.primes
        c = np.asarray(coeffs)
        if c.dtype == object:
            return np.array([[int(v) % primes[i] for v in c] for i in idx], dtype=np.uint64)
        c = c.astype(np.int64)
        out = np.empty((len(idx), len(c)), dtype=np.uint64)
        for r, i in enumerate(idx):
            out[r] = np.mod(c, np.int64(primes[i])).astype(np.uint64)   # q < 2^60 fits int64
        return out

    def fromCoeffs(self, idx, coeffs):
        idx = list(idx)
        rows = self._rows(idx, coeffs)
        return self.hx.DoubleCRT(self.gctx, idx, 1, rows[:, None, :]).FFT()

    def randomize(self, idx, rng):
        """DoubleCRT::randomize on the device: no host fill, no upload (hx_randomize; one fresh
        ChaCha20 stream of the sampler's key per call)."""
        return self.hx.DoubleCRT(self.gctx, list(idx), 1, zero=False).randomize(rng.key, rng.next_stream())

    def toPoly(self, poly):
        idx = poly.getIndexSet()
        rows = poly.copy().iFFT().download()[:, 0]
        return crt_centred([self.cc.primes[i] for i in idx], rows)

    def toPolyMod(self, poly, t):
        return [int(v) for v in poly.toPolyMod(t)[0]]

    def embeddingLargestCoeff(self, f):
        return float(self.hx.embeddingLargestCoeff(self.gctx, np.asarray(f, dtype=np.float64))[0])

    def keySwitch(self, row_idx, b, a):
        return self.hx.KeySwitch(self.gctx, list(row_idx), b, a)
