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
        }
        for name, args in sig.items():
            f = getattr(L, name)
            f.argtypes = args
            f.restype = C.c_int
        _libs[path] = L
    return _libs[path]


SYMBOLS = ["hxh_session_create", "hxh_session_destroy", "hxh_session_info", "hxh_multiply", "hxh_multiply_single",
           "hxh_plaintext", "hxh_decrypt", "hxh_result_primes", "hxh_last_error"]


class Session:
    """One benchmark session of the C++ host: scheme "bgv" (ContextBuilder<BGV>().m(m).p(p).r(r).bits(bits))
    or "ckks" (ContextBuilder<CKKS>().m(m).precision(r).bits(bits)), a key pair with its relinearisation
    matrix and `batch` pairs of fresh encryptions of seeded random plaintexts."""

    def __init__(self, scheme, m, p, r, bits, batch, device=0, stream=0, seed=7, lib_path=None):
        self.L = lib(lib_path)
        self.scheme = scheme
        self.h = C.c_void_p()
        self._chk(self.L.hxh_session_create(C.byref(self.h), device, C.c_void_p(stream), 1 if scheme == "ckks" else 0,
                                            m, p, r, bits, batch, seed))
        info = (C.c_long * 8)()
        self._chk(self.L.hxh_session_info(self.h, info))
        (self.phim, self.L_ctxt, self.K, self.D, self.n_small, self.ctxt_bits, self.special_bits, self.batch) = \
            (int(v) for v in info)
        self.p, self.r = p, r

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
        w = self._negacyclic_mod(ai, ci, P)
        return w if level == 1 else self._negacyclic_mod(w, w, P)

    def verify(self, level, elements=None):
        """decrypt(product) == plaintext product for the listed batch elements (all by default); returns the
        number checked, raises HostError on a mismatch.  CKKS: within the bound the ciphertext reports and
        1e-3 of the largest coefficient."""
        todo = range(self.batch) if elements is None else elements
        for b in todo:
            got, bound = self.decrypt(level, b)
            want = self.expected(level, b)
            if self.scheme == "ckks":
                err = float(np.max(np.abs(got - want)))
                if not (err <= bound and err < 1e-3 * float(np.max(np.abs(want)))):
                    raise HostError(f"CKKS level {level} element {b}: decode error {err} (bound {bound})")
            elif not np.array_equal(got.astype(np.int64).astype(object), np.asarray(want).astype(object)):
                raise HostError(f"decrypt(multiplyBy(a, b)) != a*b at level {level}, batch element {b}")
        return len(todo)
