"""ctypes binding of the C++17 host library (include/helib_amd_host.h, helib_amd/csrc/host_session.cpp):
the reference's benchmark loops -- keys, encryptions, `copy = ctxt1; copy.multiplyBy(ctxt2)` -- run by the
C++ Ctxt / DoubleCRT / SecKey of include/helib_amd_ctxt.hpp and helib_amd_keys.hpp; python only starts
them, synchronises and checks the decrypted results."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "lib", "libhelib_amd_host.so")
_libs = {}


class HostError(RuntimeError):
    pass


def lib(path=None):
    path = path or os.environ.get("HX_HOST_LIB") or _SO   # (HX_HOST_LIB + HX_LIB: a variant build, tools/build_variant.sh)
    if path not in _libs:
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `python -m helib_amd.build`")
        if path == _SO:
            from . import capi
            capi.lib()            # libhelib_amd.so (and the HIP runtime) first
        L = C.CDLL(path)
        L.hxh_last_error.restype = C.c_char_p
        vp, ip, lg = C.c_void_p, C.c_int, C.c_long
        sig = {
            "hxh_session_create": [vp, ip, vp, ip, lg, lg, lg, lg, ip, C.c_uint64],
            "hxh_session_destroy": [vp], "hxh_session_info": [vp, vp],
            "hxh_multiply": [vp, ip, ip, ip], "hxh_multiply_single": [vp, ip],
            "hxh_plaintext": [vp, ip, vp], "hxh_decrypt": [vp, ip, ip, vp, vp],
            "hxh_result_primes": [vp, ip, vp, ip, vp],
            "hxh_export_keys": [vp, vp, C.c_size_t, vp],
            "hxh_session_create_with_keys": [vp, ip, vp, ip, lg, lg, lg, lg, ip, C.c_uint64, vp, C.c_size_t],
            "hxh_chain_primes": [vp, vp, ip, vp], "hxh_ctxt_info": [vp, ip, ip, vp],
            "hxh_ctxt_rows": [vp, ip, ip, ip, vp, vp, ip, vp],
            "hxh_relin_matrix": [vp, vp, vp, vp, ip, vp, vp], "hxh_arena_stats": [vp, vp],
            "hxh_encrypt_decrypt_batch": [vp, ip, ip, vp],
            "hxh_session_create_source": [vp, ip, vp, ip, lg, lg, lg, lg, ip, C.c_uint64],
            "hxh_export_public_keys": [vp, vp, C.c_size_t, vp],
            "hxh_export_ctxts": [vp, ip, ip, ip, ip, vp, C.c_size_t, vp],
            "hxh_session_create_from_ctxts": [vp, ip, vp, ip, lg, lg, lg, lg, ip, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t],
            "hxh_decrypt_wire": [vp, vp, C.c_size_t, vp, vp, vp],
        }
        for name, args in sig.items():
            f = getattr(L, name)
            f.argtypes = args
            f.restype = C.c_int
        _libs[path] = L
    return _libs[path]


SYMBOLS = ["hxh_session_create", "hxh_session_destroy", "hxh_session_info", "hxh_multiply", "hxh_multiply_single",
           "hxh_plaintext", "hxh_decrypt", "hxh_result_primes", "hxh_last_error", "hxh_export_keys",
           "hxh_session_create_with_keys", "hxh_chain_primes", "hxh_ctxt_info", "hxh_ctxt_rows", "hxh_relin_matrix",
           "hxh_arena_stats", "hxh_encrypt_decrypt_batch", "hxh_session_create_source", "hxh_export_public_keys",
           "hxh_export_ctxts", "hxh_session_create_from_ctxts", "hxh_decrypt_wire"]


def ckks_correlation(got, want):
    """<got, want> / (|got| |want|) over the coefficients of a decoded CKKS product (0 when either is all zero)."""
    g, w = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    d = float(np.linalg.norm(g) * np.linalg.norm(w))
    return float(np.dot(g, w) / d) if d > 0 else 0.0


class Session:
    """One benchmark session of the C++ host: scheme "bgv" (ContextBuilder<BGV>().m(m).p(p).r(r).bits(bits))
    or "ckks" (ContextBuilder<CKKS>().m(m).precision(r).bits(bits)), a key pair with its relinearisation
    matrix and `batch` pairs of fresh encryptions of seeded random plaintexts."""

    def __init__(self, scheme, m, p, r, bits, batch, device=0, stream=0, seed=7, lib_path=None, keys=None, source=False,
                 operands=None):
        """keys: key material exported by another session of the same parameters (export_keys(): a numpy uint64
        array, e.g. as received from rank 0) -- the session then holds THAT key pair and `seed` drives only its own
        encryptions and plaintexts; None: the session generates its own key pair from `seed` (0 = OS entropy).
        source=True: a session that only HOLDS its `batch` pairs (the rank that scatters them; no arena reservation
        for a multiply loop).  operands=(a, b): two blobs of `batch` wire ciphertexts each (export_ctxts of another
        session) -- a WORKER: with keys = public material it multiplies what it was handed and cannot decrypt."""
        self.L = lib(lib_path)
        self.scheme = scheme
        self.h = C.c_void_p()
        sc = 1 if scheme == "ckks" else 0
        if operands is not None:
            if keys is None:
                raise HostError("a worker session needs key material (export_public_keys of the source)")
            keys = np.ascontiguousarray(keys, dtype=np.uint64)
            a, b = (np.ascontiguousarray(x, dtype=np.uint8) for x in operands)
            self._chk(self.L.hxh_session_create_from_ctxts(C.byref(self.h), device, C.c_void_p(stream), sc, m, p, r, bits, batch,
                                                           keys.ctypes.data_as(C.c_void_p), keys.size,
                                                           a.ctypes.data_as(C.c_void_p), a.size, b.ctypes.data_as(C.c_void_p), b.size))
        elif source:
            self._chk(self.L.hxh_session_create_source(C.byref(self.h), device, C.c_void_p(stream), sc, m, p, r, bits, batch, seed))
        elif keys is None:
            self._chk(self.L.hxh_session_create(C.byref(self.h), device, C.c_void_p(stream), sc, m, p, r, bits, batch, seed))
        else:
            keys = np.ascontiguousarray(keys, dtype=np.uint64)
            self._chk(self.L.hxh_session_create_with_keys(C.byref(self.h), device, C.c_void_p(stream), sc, m, p, r, bits,
                                                          batch, seed, keys.ctypes.data_as(C.c_void_p), keys.size))
        info = (C.c_long * 8)()
        self._chk(self.L.hxh_session_info(self.h, info))
        (self.phim, self.L_ctxt, self.K, self.D, self.n_small, self.ctxt_bits, self.special_bits, self.batch) = \
            (int(v) for v in info)
        self.p, self.r, self.m = p, r, m

    def _chk(self, rc):
        if rc != 0:
            raise HostError(self.L.hxh_last_error().decode())

    def close(self):
        if self.h:
            self.L.hxh_session_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def multiply(self, level, k, measure=True):
        """k x [copy(a); copy.multiplyBy(b)] enqueued back to back (hxh_multiply)."""
        self._chk(self.L.hxh_multiply(self.h, level, k, 1 if measure else 0))

    def multiply_single(self, measure=True):
        self._chk(self.L.hxh_multiply_single(self.h, 1 if measure else 0))

    def plaintext(self, which):
        out = np.empty((self.batch, self.phim), dtype=np.float64)
        self._chk(self.L.hxh_plaintext(self.h, which, out.ctypes.data_as(C.c_void_p)))
        return out

    def decrypt(self, level, b):
        out = np.empty(self.phim, dtype=np.float64)
        bound = C.c_double()
        self._chk(self.L.hxh_decrypt(self.h, level, b, out.ctypes.data_as(C.c_void_p), C.byref(bound)))
        return out, bound.value

    def result_primes(self, level):
        out = (C.c_int * 512)()
        n = C.c_int()
        self._chk(self.L.hxh_result_primes(self.h, level, out, 512, C.byref(n)))
        return [int(out[i]) for i in range(min(n.value, 512))]

    # ---- one key pair, many GPUs; and what a checker needs of a session ----
    def export_keys(self):
        """the key pair with its relinearisation matrix as uint64 words (hxh_export_keys)"""
        need = C.c_size_t()
        self._chk(self.L.hxh_export_keys(self.h, None, 0, C.byref(need)))
        out = np.empty(need.value, dtype=np.uint64)
        self._chk(self.L.hxh_export_keys(self.h, out.ctypes.data_as(C.c_void_p), out.size, None))
        return out

    def export_public_keys(self):
        """the same without the secret polynomial (SecKey::exportKeys(false)): what a worker needs"""
        need = C.c_size_t()
        self._chk(self.L.hxh_export_public_keys(self.h, None, 0, C.byref(need)))
        out = np.empty(need.value, dtype=np.uint64)
        self._chk(self.L.hxh_export_public_keys(self.h, out.ctypes.data_as(C.c_void_p), out.size, None))
        return out

    def export_ctxts(self, level, which=0, first=0, count=None):
        """batch elements [first, first + count) of fresh operand `which` (level 0) or of the kept product of level
        1 / 2, each in the reference's binary ciphertext format (Ctxt::writeTo, src/Ctxt.cpp:2584-2611), concatenated:
        a numpy uint8 array"""
        count = self.batch - first if count is None else count
        need = C.c_size_t()
        self._chk(self.L.hxh_export_ctxts(self.h, level, which, first, count, None, 0, C.byref(need)))
        out = np.empty(need.value, dtype=np.uint8)
        self._chk(self.L.hxh_export_ctxts(self.h, level, which, first, count, out.ctypes.data_as(C.c_void_p), out.size, None))
        return out

    def decrypt_wire(self, blob, offset=0):
        """ONE wire ciphertext at blob[offset:] under this session's secret key -> (values, bound, bytes consumed)"""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        out = np.empty(self.phim, dtype=np.float64)
        bound, used = C.c_double(), C.c_size_t()
        self._chk(self.L.hxh_decrypt_wire(self.h, C.c_void_p(blob.ctypes.data + offset), blob.size - offset,
                                          out.ctypes.data_as(C.c_void_p), C.byref(bound), C.byref(used)))
        return out, bound.value, int(used.value)

    def verify_blob(self, blob, level, first, count):
        """decrypt(each of the `count` products of a blob) == the plaintext product of THIS session's elements
        first .. first + count - 1 (the session that encrypted the operands and holds the secret key: rank 0 of a
        scatter / gather).  Returns the number checked; raises HostError on a mismatch or on trailing bytes."""
        off = 0
        for b in range(first, first + count):
            got, bound, used = self.decrypt_wire(blob, off)
            off += used
            self._check_one(level, b, got, bound)
        if off != len(blob):
            raise HostError(f"verify_blob: {len(blob) - off} trailing bytes after {count} ciphertexts")
        return count

    def chain_primes(self):
        n = C.c_int()
        self._chk(self.L.hxh_chain_primes(self.h, None, 0, C.byref(n)))
        out = np.empty(n.value, dtype=np.uint64)
        self._chk(self.L.hxh_chain_primes(self.h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return [int(q) for q in out]

    def ctxt_info(self, level, which=0):
        """dict of the bookkeeping of fresh operand `which` (level 0) or the kept product of level 1 / 2"""
        info = (C.c_double * 8)()
        self._chk(self.L.hxh_ctxt_info(self.h, level, which, info))
        keys = ("lnNoise", "lnRatFactor", "ptxtMag", "intFactor", "ptxtSpace", "nparts", "key_lnNoise", "key_ptxtSpace")
        return dict(zip(keys, (float(v) for v in info)))

    def ctxt_rows(self, level, which, part):
        """(prime indices by row, rows [nrows, batch, phi(m)]) of part 0 / 1 of a ciphertext the session holds"""
        n = C.c_int()
        self._chk(self.L.hxh_ctxt_rows(self.h, level, which, part, None, None, 0, C.byref(n)))
        idx = (C.c_int * n.value)()
        out = np.empty((n.value, self.batch, self.phim), dtype=np.uint64)
        self._chk(self.L.hxh_ctxt_rows(self.h, level, which, part, out.ctypes.data_as(C.c_void_p), idx, n.value, C.byref(n)))
        return [int(i) for i in idx], out

    def relin_matrix(self):
        """(row primes, b, a) of the relinearisation matrix, b / a = [ndig, nrows, phi(m)]"""
        nd, nr = C.c_int(), C.c_int()
        self._chk(self.L.hxh_relin_matrix(self.h, None, None, None, 0, C.byref(nd), C.byref(nr)))
        idx = (C.c_int * nr.value)()
        b = np.empty((nd.value, nr.value, self.phim), dtype=np.uint64)
        a = np.empty_like(b)
        self._chk(self.L.hxh_relin_matrix(self.h, b.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), idx, nr.value,
                                          C.byref(nd), C.byref(nr)))
        return [int(i) for i in idx], b, a

    def arena_stats(self):
        out = (C.c_uint64 * 4)()
        self._chk(self.L.hxh_arena_stats(self.h, out))
        return {"reserved_bytes": int(out[0]), "in_use_bytes": int(out[1]), "hipMalloc_calls": int(out[2]), "parked_blocks": int(out[3])}

    def encrypt_decrypt_batch(self, batch, reps=3):
        """SecKey::EncryptBatch / DecryptBatch timed in the C++ host: ms per ciphertext, and whether all round-tripped"""
        out = (C.c_double * 4)()
        self._chk(self.L.hxh_encrypt_decrypt_batch(self.h, batch, reps, out))
        return {"encrypt_ms_per_ciphertext": round(out[0], 4), "decrypt_ms_per_ciphertext": round(out[1], 4),
                "batch": int(out[2]), "all_elements_round_trip": bool(out[3] == 1.0)}

    # ---- the checks bench.py and the tests apply to a kept product ----
    @staticmethod
    def _negacyclic_mod(x, y, P):
        """x * y mod (X^n + 1, P), exactly.  P < 2^18 (the benchmark's 65537): four floating-point FFT
        convolutions of 9-bit halves -- every partial sum stays below 2^32, far inside what a double FFT rounds
        exactly; larger P: direct integer convolution (python integers when n * P^2 does not fit 62 bits)."""
        n = len(x)

        def fold(full):
            return full[:n] - np.append(full[n:], 0)
        if P < (1 << 18):
            from scipy.signal import fftconvolve
            x, y = np.asarray(x, dtype=np.int64), np.asarray(y, dtype=np.int64)
            xs = [(x & 511).astype(np.float64), (x >> 9).astype(np.float64)]
            ys = [(y & 511).astype(np.float64), (y >> 9).astype(np.float64)]
            acc = np.zeros(n, dtype=np.int64)
            for i in range(2):
                for j in range(2):
                    c = np.rint(fold(fftconvolve(xs[i], ys[j]))).astype(np.int64)
                    acc = (acc + (c % P) * ((1 << (9 * (i + j))) % P)) % P
            return acc
        if n * P * P < 1 << 62:
            return np.mod(fold(np.convolve(np.asarray(x, dtype=np.int64), np.asarray(y, dtype=np.int64))), P)
        full = fold(np.convolve(np.asarray(x).astype(object), np.asarray(y).astype(object)))
        return np.array([int(v) % P for v in full], dtype=object)

    def expected(self, level, b):
        """the plaintext the kept product of `level` must decrypt to (element b; level 0: the first operand)"""
        a, c = self.plaintext(0)[b], self.plaintext(1)[b]
        n = self.phim
        if self.scheme == "ckks":
            from scipy.signal import fftconvolve

            def nega(x, y):
                full = fftconvolve(x, y)
                return full[:n] - np.append(full[n:], 0)
            if level == 0:
                return a
            w = nega(a, c)
            return w if level == 1 else nega(w, w)
        P = self.p ** self.r
        ai, ci = a.astype(np.int64), c.astype(np.int64)
        if level == 0:
            return ai
        if (self.m & (self.m - 1)) == 0:
            mul = self._negacyclic_mod
        elif self.phim == self.m - 1:
            mul = self._mul_mod_phi_prime_m
        else:
            mul = self._mul_mod_phi_squarefree_m
        w = mul(ai, ci, P)
        return w if level == 1 else mul(w, w, P)

    def _mul_mod_phi_squarefree_m(self, x, y, P):
        """x * y mod (Phi_m, P) for an odd SQUAREFREE m (BASELINE config 5's ring, m = 21845 = 5 * 17 * 257): the product
        folded modulo X^m - 1, then rem Phi_m by binomial passes -- Phi_m = prod_{d | m} (X^d - 1)^mu(m/d), multiplying by
        1 - X^d is w_i -= w_(i-d), dividing the running sum of stride d (exact over the integers, here modulo P) -- the
        same identity the inverse transform's fused tail uses on the device (helib_amd/csrc/pfa_core.h), in numpy."""
        from itertools import combinations
        m, n = self.m, self.phim
        ps, r = [], m
        f = 3
        while f * f <= r:
            if r % f == 0:
                ps.append(f)
                r //= f
                if r % f == 0:
                    raise NotImplementedError("Session.expected: m not squarefree")
            f += 2
        if r > 1:
            ps.append(r)
        if m % 2 == 0:
            raise NotImplementedError("Session.expected: even general m")
        num, den = [], []
        for k in range(len(ps) + 1):
            for sub in combinations(ps, k):
                d = int(np.prod(sub)) if sub else 1
                (num if (len(ps) - k) % 2 == 0 else den).append(d)
        num = [d for d in num if d != m]
        from scipy.signal import fftconvolve
        x, y = np.asarray(x, dtype=np.int64), np.asarray(y, dtype=np.int64)
        if P < (1 << 10):
            full = np.rint(fftconvolve(x.astype(np.float64), y.astype(np.float64))).astype(np.int64)   # exact: sums < 2^35
        else:
            full = np.array([int(v) for v in np.convolve(x.astype(object), y.astype(object))], dtype=object)
        X = np.zeros(m, dtype=np.int64)
        X[:min(len(full), m)] += np.asarray(full[:m] % P, dtype=np.int64)
        if len(full) > m:
            X[:len(full) - m] += np.asarray(full[m:] % P, dtype=np.int64)
        X %= P
        dq = m - 1 - n

        def times(w, d):          # w (1 - t^d)
            out = w.copy()
            out[d:] -= w[:-d]
            return out % P

        def over(w, d):           # w / (1 - t^d): running sums along the d chains
            L = len(w)
            pad = (-L) % d
            v = np.concatenate([w, np.zeros(pad, dtype=np.int64)]).reshape(-1, d)
            return (np.cumsum(v % P, axis=0) % P).reshape(-1)[:L]
        w = X[::-1][:dq + 1].copy()
        for d in den:
            if d <= dq:
                w = times(w, d)
        for d in num:
            if d <= dq:
                w = over(w, d)
        W = np.zeros(n, dtype=np.int64)
        W[:dq + 1] = w[::-1]
        for d in num:
            W = times(W, d)
        for d in den:
            W = over(W, d)
        return (X[:n] - W) % P

    def _mul_mod_phi_prime_m(self, x, y, P):
        """x * y mod (Phi_m, P) for a PRIME m (Phi_m = 1 + X + ... + X^(m-1); the reference's general-m benchmark
        ring, m = 32003): fold modulo X^m - 1, then subtract the coefficient of X^(m-1) from every other one."""
        m = self.m
        if any(m % d == 0 for d in range(2, int(m ** 0.5) + 1)) or self.phim != m - 1:
            raise NotImplementedError("Session.expected: m a power of two or a prime")
        from scipy.signal import fftconvolve
        x, y = np.asarray(x, dtype=np.int64), np.asarray(y, dtype=np.int64)
        if P < (1 << 10):
            full = np.rint(fftconvolve(x.astype(np.float64), y.astype(np.float64))).astype(np.int64)   # exact: sums < 2^35
        else:
            full = np.convolve(x.astype(object), y.astype(object))
        c = np.zeros(m, dtype=full.dtype)
        c[:min(len(full), m)] += full[:m]
        if len(full) > m:
            c[:len(full) - m] += full[m:]
        out = (c[:m - 1] - c[m - 1])
        return np.array([int(v) % P for v in out], dtype=np.int64)

    def verify(self, level, elements=None):
        """decrypt(product) == plaintext product for the listed batch elements (all by default); returns the
        number checked, raises HostError on a mismatch.  CKKS: within the bound the ciphertext reports (and, from
        precision(10) on, 1e-3 of the largest coefficient)."""
        todo = range(self.batch) if elements is None else elements
        for b in todo:
            got, bound = self.decrypt(level, b)
            self._check_one(level, b, got, bound)
        return len(todo)

    def _check_one(self, level, b, got, bound):
        want = self.expected(level, b)
        if self.scheme == "ckks":
            # precision(r) promises 2^-r.  From 10 bits on the product must be right to 1e-3 of its size.  At the
            # reference's benchmark setting precision(1) the reported bound (O(1)) exceeds the product's own
            # coefficients (~1e-4): "within the bound" alone would accept an all-zero or a foreign product.  The
            # signal is still there, in all N coefficients at once: the normalised correlation of the decoded
            # product with the expected one is 1/sqrt(1 + noise^2/signal^2), while a product of other operands (or
            # zeros) correlates like N(0, 1/N) -- it must clear 8 standard deviations of that.
            err = float(np.max(np.abs(got - want)))
            ok = err <= bound
            if ok and self.r >= 10:
                ok = err < 1e-3 * float(np.max(np.abs(want)))
            elif ok:
                corr = ckks_correlation(got, want)
                self.min_ckks_correlation = min(getattr(self, "min_ckks_correlation", 1.0), corr)
                ok = corr >= 8.0 / np.sqrt(len(want))
            if not ok:
                raise HostError(f"CKKS level {level} element {b}: decode error {err} (bound {bound})"
                                + (f", correlation with the expected product {ckks_correlation(got, want):.4f}" if self.r < 10 else ""))
        elif not np.array_equal(got.astype(np.int64).astype(object), np.asarray(want).astype(object)):
            raise HostError(f"decrypt(multiplyBy(a, b)) != a*b at level {level}, batch element {b}")
