"""Oracle-backed stand-in for helib_amd.capi so that helib_amd.ctxt.Ctxt (the host control flow)
can be driven on the CPU in tests: same method names as capi.DoubleCRT / capi module functions.
TEST INFRASTRUCTURE: lives under oracle/ so that only tests/, smoke() and the cpu_baseline leg of
bench.py can reach it."""
import numpy as np

from oracle import oracle as O


class OPoly:
    def __init__(self, octx, idx, rows):
        self.o, self.idx = octx, list(idx)
        self.rows = np.ascontiguousarray(rows, dtype=np.uint64)   # [nrows, N]

    def getIndexSet(self):
        return list(self.idx)

    def copy(self):
        return OPoly(self.o, self.idx, self.rows.copy())

    def download(self):
        return self.rows[:, None, :]

    def __iadd__(self, other):
        for r, i in enumerate(self.idx):
            self.rows[r] = O.row_op("add", self.rows[r], other.rows[other.idx.index(i)], self.o.primes[i])
        return self

    def _binary(self, name, other):
        for r, i in enumerate(self.idx):
            self.rows[r] = O.row_op(name, self.rows[r], other.rows[other.idx.index(i)], self.o.primes[i])
        return self

    def __isub__(self, other):
        return self._binary("sub", other)

    def __imul__(self, other):
        return self._binary("mul", other)

    def mulConstant(self, num):
        for r, i in enumerate(self.idx):
            q = self.o.primes[i]
            self.rows[r] = O.row_op("mul_scalar", self.rows[r], int(num) % q, q)
        return self

    def Negate(self):
        return self.mulConstant(-1)

    def Exp(self, e):
        """DoubleCRT::Exp (src/DoubleCRT.cpp:1142-1156): entry-wise PowerMod by square and multiply"""
        base, acc = self.copy(), None
        while e:
            if e & 1:
                acc = base.copy() if acc is None else acc._binary("mul", base)
            e >>= 1
            if e:
                base._binary("mul", base.copy())
        if acc is None:
            for r, i in enumerate(self.idx):
                self.rows[r] = 1 % self.o.primes[i]
        else:
            self.rows = acc.rows
        return self

    def automorph(self, k):
        zms = O.zmstar(self.o.m)
        self.rows = np.stack([O.automorph(r, self.o.m, zms, k) for r in self.rows])
        return self

    def addPrimesAndScale(self, s):
        s = list(s)
        self.rows = np.vstack([self.o.scale_by_primes(self.idx, self.rows, s),
                               np.zeros((len(s), self.o.N), dtype=np.uint64)])
        self.idx += s
        return self

    def scaleDownToSet(self, keep, ptxt, norms=False):
        """norms=True: returns [embeddingLargestCoeff(delta/diffProd)] (src/Ctxt.cpp:466-507)."""
        keep = set(keep)
        drop = [i for i in self.idx if i not in keep]
        if not drop:
            return np.zeros(1) if norms else self
        if norms:
            self.rows, fd = self.o.scale_down(self.idx, self.rows, drop, ptxt, want_fdelta=True)
        else:
            self.rows = self.o.scale_down(self.idx, self.rows, drop, ptxt)
        self.idx = [i for i in self.idx if i in keep]
        return np.array([O.embedding_largest_coeff(self.o.m, fd)]) if norms else self


class OKeySwitch:
    def __init__(self, row_idx, b, a):
        self.row_idx, self.b, self.a = list(row_idx), b, a


class OracleOps:
    def __init__(self, octx):
        self.o = octx

    def tensorProduct(self, c0, c1, d0, d1):
        assert c0.idx == c1.idx == d0.idx == d1.idx
        t = self.o.tensor(c0.idx, c0.rows, c1.rows, d0.rows, d1.rows)
        return [OPoly(self.o, c0.idx, x) for x in t]

    def tensorBringToSet(self, c0, c1, d0, d1, add_set, keep_set, ptxtSpace, norms=False, defer=False):
        """The reference's own sequence for what the device fuses: Ctxt::tensorProduct (src/Ctxt.cpp:1563-1608), then
        per product part addPrimesAndScale(add) and scaleDownToSet(keep) (Ctxt::bringToSet, :346-562)."""
        outs = self.tensorProduct(c0, c1, d0, d1)
        add = [i for i in add_set if i not in c0.idx]
        nrm = []
        for t in outs:
            if add:
                t.addPrimesAndScale(add)
            r = t.scaleDownToSet(keep_set, ptxtSpace, norms=norms)
            if norms:
                nrm.append(r)
        return (outs, np.array(nrm).reshape(3, 1)) if norms else outs

    def mulRelin(self, c0, c1, d0, d1, W, digits, norms=False, defer=False):
        """Ctxt::tensorProduct followed by Ctxt::reLinearize at the full level (src/Ctxt.cpp:1563-1608, :720-786)"""
        t = self.tensorProduct(c0, c1, d0, d1)
        sp = W.row_idx[len(c0.idx):]
        return self.reLinearize(t[0], t[1], t[2], W, digits, sp, norms=norms)

    @staticmethod
    def supportsNorms(m):
        return True

    def breakIntoDigits(self, part, digits, special, norms=False):
        allp = part.idx + list(special)
        if norms:
            dg, nrm = self.o.break_into_digits(part.idx, part.rows, digits, allp, want_norms=True)
        else:
            dg, nrm = self.o.break_into_digits(part.idx, part.rows, digits, allp), None
        blk = OPoly(self.o, allp * len(digits), dg.reshape(-1, self.o.N))
        return (blk, nrm.reshape(-1, 1)) if norms else blk

    def keySwitchDigits(self, dg, W, out0, out1):
        allp = out0.idx
        D = dg.rows.shape[0] // len(allp)
        sel = [W.row_idx.index(i) for i in allp]
        kb, ka = np.ascontiguousarray(W.b[:D][:, sel]), np.ascontiguousarray(W.a[:D][:, sel])
        out0.rows, out1.rows = self.o.key_switch_digits(allp, dg.rows.reshape(D, len(allp), -1), kb, ka,
                                                        out0.rows, out1.rows)

    @staticmethod
    def zerosLike(poly):
        return OPoly(poly.o, poly.idx, np.zeros_like(poly.rows))

    def reLinearize(self, t0, t1, t2, W, digits, special, norms=False, defer=False):
        own, sp = t0.idx, list(special)
        allp = own + sp
        sel = [W.row_idx.index(i) for i in allp]
        D = len(digits)
        kb, ka = W.b[:D][:, sel], W.a[:D][:, sel]
        s0 = self.o.scale_by_primes(own, t0.rows, sp)
        s1 = self.o.scale_by_primes(own, t1.rows, sp) if t1 is not None else np.zeros_like(s0)
        z = np.zeros((len(sp), self.o.N), dtype=np.uint64)
        if norms:
            dg, nrm = self.o.break_into_digits(own, t2.rows, digits, allp, want_norms=True)
        else:
            dg = self.o.break_into_digits(own, t2.rows, digits, allp)
        o0, o1 = self.o.key_switch_digits(allp, dg, np.ascontiguousarray(kb), np.ascontiguousarray(ka),
                                          np.vstack([s0, z]), np.vstack([s1, z]))
        if norms:
            return OPoly(self.o, allp, o0), OPoly(self.o, allp, o1), nrm.reshape(-1, 1)
        return OPoly(self.o, allp, o0), OPoly(self.o, allp, o1)


class OracleBackend:
    """The polynomial side of helib_amd.keys over the CPU oracle (same interface as
    helib_amd.keys.HxBackend)."""

    def __init__(self, octx, context):
        self.o, self.cc = octx, context
        self.ops = OracleOps(octx)

    def fromCoeffs(self, idx, coeffs):
        idx = list(idx)
        coef = np.array([[int(c) % self.o.primes[i] for c in coeffs] for i in idx], dtype=np.uint64)
        return OPoly(self.o, idx, self.o.fft(idx, coef))

    def randomize(self, idx, rng):
        """DoubleCRT::randomize over the same ChaCha20 stream the device kernel uses"""
        idx, stream = list(idx), rng.next_stream()
        rows = np.stack([O.randomize_row(self.o.N, self.o.primes[i], rng.key, stream, i)[0] for i in idx])
        return OPoly(self.o, idx, rows)

    def toPoly(self, poly):
        return self.o.to_poly(poly.idx, poly.rows)

    def toPolyMod(self, poly, t):
        return [int(v) % t for v in self.toPoly(poly)]

    def embeddingLargestCoeff(self, f):
        return O.embedding_largest_coeff(self.o.m, np.asarray(f, dtype=np.float64))

    def keySwitch(self, row_idx, b, a):
        return OKeySwitch(row_idx, b, a)
