"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under helib_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "hx_oracle.c")
    hdr = os.path.join(_HERE, "hx_oracle.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(
            os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("HX_ORACLE_SO") or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            so = build()
        L = C.CDLL(so)
        L.ho_mulmod.restype = C.c_uint64
        L.ho_mulmod.argtypes = [C.c_uint64] * 3
        L.ho_powmod.restype = C.c_uint64
        L.ho_powmod.argtypes = [C.c_uint64] * 3
        L.ho_invmod.restype = C.c_uint64
        L.ho_invmod.argtypes = [C.c_uint64] * 2
        L.ho_is_prime.restype = C.c_int
        L.ho_is_prime.argtypes = [C.c_uint64]
        L.ho_find_prim_root.restype = C.c_uint64
        L.ho_find_prim_root.argtypes = [C.c_uint64, C.c_uint64]
        L.ho_hexl_minimal_primitive_root.restype = C.c_uint64
        L.ho_hexl_minimal_primitive_root.argtypes = [C.c_uint64, C.c_uint64]
        for f in (L.ho_hexl_forward, L.ho_hexl_inverse):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_uint64]
        L.ho_zmstar.restype = C.c_long
        L.ho_zmstar.argtypes = [C.c_uint64, C.c_void_p, C.c_long]
        L.ho_phimx.argtypes = [C.c_uint64, C.c_void_p]
        L.ho_primegen_init.argtypes = [C.c_void_p, C.c_long, C.c_long]
        L.ho_primegen_next.restype = C.c_long
        L.ho_primegen_next.argtypes = [C.c_void_p]
        L.ho_cmod_create.restype = C.c_void_p
        L.ho_cmod_create.argtypes = [C.c_uint64] * 3
        L.ho_cmod_destroy.argtypes = [C.c_void_p]
        L.ho_cmod_root.restype = C.c_uint64
        L.ho_cmod_root.argtypes = [C.c_void_p]
        L.ho_cmod_phim.restype = C.c_long
        L.ho_cmod_phim.argtypes = [C.c_void_p]
        for f in (L.ho_cmod_fft, L.ho_cmod_ifft):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ho_cmod_eval_naive.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long]
        for f in (L.ho_row_add, L.ho_row_sub, L.ho_row_mul):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_uint64]
        L.ho_row_neg.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_uint64]
        for f in (L.ho_row_add_scalar, L.ho_row_sub_scalar, L.ho_row_mul_scalar):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_long, C.c_uint64]
        L.ho_row_automorph.restype = C.c_int
        L.ho_row_automorph.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_long, C.c_uint64]
        L.ho_ctx_create.restype = C.c_void_p
        L.ho_ctx_create.argtypes = [C.c_uint64]
        L.ho_ctx_destroy.argtypes = [C.c_void_p]
        L.ho_ctx_add_prime.restype = C.c_int
        L.ho_ctx_add_prime.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.ho_ctx_phim.restype = C.c_long
        L.ho_ctx_phim.argtypes = [C.c_void_p]
        L.ho_ctx_prime.restype = C.c_uint64
        L.ho_ctx_prime.argtypes = [C.c_void_p, C.c_int]
        L.ho_ctx_root.restype = C.c_uint64
        L.ho_ctx_root.argtypes = [C.c_void_p, C.c_int]
        L.ho_ctx_zms.restype = C.c_void_p
        L.ho_ctx_zms.argtypes = [C.c_void_p]
        for f in (L.ho_dcrt_fft, L.ho_dcrt_ifft):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ho_dcrt_add_primes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ho_dcrt_scale_by_primes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_int]
        L.ho_dcrt_break_into_digits.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_int, C.c_void_p]
        L.ho_dcrt_break_into_digits_norms.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                      C.c_int, C.c_void_p, C.c_void_p]
        L.ho_embedding_largest_coeff.restype = C.c_double
        L.ho_embedding_largest_coeff.argtypes = [C.c_uint64, C.c_void_p, C.c_long]
        L.ho_dcrt_scale_down.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
        L.ho_tensor.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7
        L.ho_key_switch_digits.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.ho_mul_relin.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8
        L.ho_dcrt_to_poly_limbs.restype = C.c_int
        L.ho_dcrt_to_poly_limbs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_int, C.c_void_p]
        L.ho_fill_uniform.argtypes = [C.c_void_p, C.c_long, C.c_uint64, C.c_uint64]
        L.ho_chacha20_block.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.ho_randomize_row.restype = C.c_long
        L.ho_randomize_row.argtypes = [C.c_void_p, C.c_long, C.c_uint64, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class PrimeGen:
    """PrimeGenerator(len, m) -- src/PrimeGenerator.h:41-126."""

    def __init__(self, length, m):
        self._st = (C.c_long * 4)()
        lib().ho_primegen_init(C.byref(self._st), length, m)

    def next(self):
        q = lib().ho_primegen_next(C.byref(self._st))
        if q == 0:
            raise RuntimeError("Prime generator ran out of primes")
        return int(q)


def zmstar(m):
    n = lib().ho_zmstar(m, None, 0)
    out = np.zeros(n, dtype=np.uint32)
    lib().ho_zmstar(m, _p(out), n)
    return out


def phimx(m):
    n = lib().ho_zmstar(m, None, 0)
    out = np.zeros(n + 1, dtype=np.int64)
    lib().ho_phimx(m, _p(out))
    return out


def embedding_largest_coeff(m, f):
    """embeddingLargestCoeff (src/norms.cpp:480-493) of the real polynomial f."""
    f = np.ascontiguousarray(f, dtype=np.float64)
    return float(lib().ho_embedding_largest_coeff(m, _p(f), len(f)))


def fill_uniform(n, q, seed):
    out = np.zeros(n, dtype=np.uint64)
    lib().ho_fill_uniform(_p(out), n, q, seed)
    return out


def chacha20_block(key, counter, nonce):
    """RFC 8439 block function: key = 32 bytes, nonce = 12 bytes -> 64 bytes."""
    k = np.frombuffer(bytes(key), dtype="<u4").astype(np.uint32)
    nn = np.frombuffer(bytes(nonce), dtype="<u4").astype(np.uint32)
    out = np.zeros(64, dtype=np.uint8)
    lib().ho_chacha20_block(_p(k), counter, _p(nn), _p(out))
    return out.tobytes()


def randomize_row(n, q, key, stream, prime_index, batch_element=0):
    """One row of DoubleCRT::randomize over the ChaCha20 stream (key, nonce = stream_lo, stream_hi,
    prime_index | batch_element << 16).  Returns (row, buffers consumed)."""
    k = np.frombuffer(bytes(key), dtype="<u4").astype(np.uint32)
    nonce = np.array([stream & 0xffffffff, (stream >> 32) & 0xffffffff,
                      (prime_index & 0xffff) | ((batch_element & 0xffff) << 16)], dtype=np.uint32)
    out = np.zeros(n, dtype=np.uint64)
    nbuf = lib().ho_randomize_row(_p(out), n, q, _p(k), _p(nonce))
    return out, int(nbuf)


def hexl_minimal_primitive_root(q, e):
    """hexl::MinimalPrimitiveRoot(e, q): the root hexl::NTT(n, q) uses is this at e = 2n."""
    return int(lib().ho_hexl_minimal_primitive_root(q, e))


def hexl_forward(x, q):
    """intel::FFTFwd = hexl::NTT(n, q).ComputeForward (src/intelExt.cpp:76-84): bit-reversed output."""
    x = _u64(x)
    y = np.zeros_like(x)
    lib().ho_hexl_forward(_p(y), _p(x), len(x), q)
    return y


def hexl_inverse(y, q):
    """intel::FFTRev1 = hexl::NTT(n, q).ComputeInverse (src/intelExt.cpp:87-98): bit-reversed input."""
    y = _u64(y)
    x = np.zeros_like(y)
    lib().ho_hexl_inverse(_p(x), _p(y), len(y), q)
    return x


def bit_reverse_copy(a):
    """BitReverseCopy (src/CModulus.cpp:284-353): B[rev(i)] = A[i]."""
    a = np.asarray(a)
    n = len(a)
    bits = n.bit_length() - 1
    rev = np.array([int(format(i, "0%db" % bits)[::-1], 2) if bits else 0 for i in range(n)])
    out = np.empty_like(a)
    out[rev] = a
    return out


class Cmod:
    """One Cmodulus (src/CModulus.cpp)."""

    def __init__(self, m, q, root=0):
        self.h = lib().ho_cmod_create(m, q, root)
        self.m, self.q = m, q
        self.phim = lib().ho_cmod_phim(self.h)
        self.root = int(lib().ho_cmod_root(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ho_cmod_destroy(self.h)
            self.h = None

    def fft(self, x):
        x = _u64(x)
        y = np.zeros(self.phim, dtype=np.uint64)
        lib().ho_cmod_fft(self.h, _p(x), _p(y))
        return y

    def ifft(self, y):
        y = _u64(y)
        x = np.zeros(self.phim, dtype=np.uint64)
        lib().ho_cmod_ifft(self.h, _p(y), _p(x))
        return x

    def eval_naive(self, x, j0=0, j1=None):
        x = _u64(x)
        j1 = self.phim if j1 is None else j1
        y = np.zeros(j1 - j0, dtype=np.uint64)
        lib().ho_cmod_eval_naive(self.h, _p(x), _p(y), j0, j1)
        return y


class Ctx:
    """Prime chain + transforms; rows are numpy uint64 [nrows, N]."""

    def __init__(self, m):
        self.h = lib().ho_ctx_create(m)
        self.m = m
        self.N = lib().ho_ctx_phim(self.h)
        self.primes = []
        self.roots = []

    def __del__(self):
        if getattr(self, "h", None):
            lib().ho_ctx_destroy(self.h)
            self.h = None

    def add_prime(self, q, root=0):
        i = lib().ho_ctx_add_prime(self.h, q, root)
        self.primes.append(int(q))
        self.roots.append(int(lib().ho_ctx_root(self.h, i)))
        return i

    def zms(self):
        return zmstar(self.m)

    def fft(self, idx, coef):
        idx, coef = _i32(idx), _u64(coef)
        out = np.zeros_like(coef)
        lib().ho_dcrt_fft(self.h, _p(idx), len(idx), _p(coef), _p(out))
        return out

    def ifft(self, idx, ev):
        idx, ev = _i32(idx), _u64(ev)
        out = np.zeros_like(ev)
        lib().ho_dcrt_ifft(self.h, _p(idx), len(idx), _p(ev), _p(out))
        return out

    def add_primes(self, from_idx, rows, to_idx, want_poly=False):
        from_idx, to_idx, rows = _i32(from_idx), _i32(to_idx), _u64(rows)
        out = np.zeros((len(to_idx), self.N), dtype=np.uint64)
        pf = np.zeros(self.N, dtype=np.float64) if want_poly else None
        lib().ho_dcrt_add_primes(self.h, _p(from_idx), len(from_idx), _p(rows),
                                 _p(to_idx), len(to_idx), _p(out),
                                 _p(pf) if want_poly else None)
        return (out, pf) if want_poly else out

    def scale_by_primes(self, from_idx, rows, add_idx):
        from_idx, add_idx = _i32(from_idx), _i32(add_idx)
        rows = _u64(rows).copy()
        lib().ho_dcrt_scale_by_primes(self.h, _p(from_idx), len(from_idx), _p(rows),
                                      _p(add_idx), len(add_idx))
        return rows

    def break_into_digits(self, own_idx, rows, digits, all_idx, want_norms=False):
        """want_norms: also embeddingLargestCoeff(digit_d)/P_d per digit (the pieces of the
        reference's return value, src/DoubleCRT.cpp:538-545)."""
        own_idx, all_idx, rows = _i32(own_idx), _i32(all_idx), _u64(rows)
        dig_idx = _i32([p for d in digits for p in d])
        dig_off = _i32(np.concatenate([[0], np.cumsum([len(d) for d in digits])]))
        out = np.zeros((len(digits), len(all_idx), self.N), dtype=np.uint64)
        nrm = np.zeros(len(digits), dtype=np.float64) if want_norms else None
        lib().ho_dcrt_break_into_digits_norms(self.h, _p(own_idx), len(own_idx), _p(rows),
                                              _p(dig_idx), _p(dig_off), len(digits),
                                              _p(all_idx), len(all_idx), _p(out),
                                              _p(nrm) if want_norms else None)
        return (out, nrm) if want_norms else out

    def scale_down(self, own_idx, rows, drop_idx, ptxt_space, want_fdelta=False):
        own_idx, drop_idx, rows = _i32(own_idx), _i32(drop_idx), _u64(rows)
        nkeep = len(own_idx) - len(drop_idx)
        out = np.zeros((nkeep, self.N), dtype=np.uint64)
        fd = np.zeros(self.N, dtype=np.float64) if want_fdelta else None
        lib().ho_dcrt_scale_down(self.h, _p(own_idx), len(own_idx), _p(rows),
                                 _p(drop_idx), len(drop_idx), ptxt_space, _p(out),
                                 _p(fd) if want_fdelta else None)
        return (out, fd) if want_fdelta else out

    def tensor(self, idx, c0, c1, d0, d1):
        idx = _i32(idx)
        c0, c1, d0, d1 = map(_u64, (c0, c1, d0, d1))
        o = [np.zeros_like(c0) for _ in range(3)]
        lib().ho_tensor(self.h, _p(idx), len(idx), _p(c0), _p(c1), _p(d0), _p(d1),
                        _p(o[0]), _p(o[1]), _p(o[2]))
        return o

    def key_switch_digits(self, all_idx, digits, ksk_b, ksk_a, out0, out1):
        all_idx = _i32(all_idx)
        digits, ksk_b, ksk_a = map(_u64, (digits, ksk_b, ksk_a))
        out0, out1 = _u64(out0).copy(), _u64(out1).copy()
        lib().ho_key_switch_digits(self.h, _p(all_idx), len(all_idx), digits.shape[0],
                                   _p(digits), _p(ksk_b), _p(ksk_a), _p(out0), _p(out1))
        return out0, out1

    def mul_relin(self, own_idx, sp_idx, digits, c0, c1, d0, d1, ksk_b, ksk_a):
        own_idx, sp_idx = _i32(own_idx), _i32(sp_idx)
        dig_idx = _i32([p for d in digits for p in d])
        dig_off = _i32(np.concatenate([[0], np.cumsum([len(d) for d in digits])]))
        c0, c1, d0, d1, ksk_b, ksk_a = map(_u64, (c0, c1, d0, d1, ksk_b, ksk_a))
        nall = len(own_idx) + len(sp_idx)
        o0 = np.zeros((nall, self.N), dtype=np.uint64)
        o1 = np.zeros((nall, self.N), dtype=np.uint64)
        lib().ho_mul_relin(self.h, _p(own_idx), len(own_idx), _p(sp_idx), len(sp_idx),
                           _p(dig_idx), _p(dig_off), len(digits), _p(c0), _p(c1), _p(d0),
                           _p(d1), _p(ksk_b), _p(ksk_a), _p(o0), _p(o1))
        return o0, o1

    def to_poly(self, idx, rows, positive=False):
        """toPoly -> list of python ints (src/DoubleCRT.cpp:925-1113)."""
        idx, rows = _i32(idx), _u64(rows)
        bits = sum(int(self.primes[i]).bit_length() for i in idx)
        nl = bits // 64 + 3
        mag = np.zeros((self.N, nl), dtype=np.uint64)
        sign = np.zeros(self.N, dtype=np.int8)
        rc = lib().ho_dcrt_to_poly_limbs(self.h, _p(idx), len(idx), _p(rows),
                                         1 if positive else 0, _p(mag), nl, _p(sign))
        assert rc == 0
        out = []
        for h in range(self.N):
            v = 0
            for i in range(nl - 1, -1, -1):
                v = (v << 64) | int(mag[h, i])
            out.append(v * int(sign[h]))
        return out


def row_op(name, a, b, q):
    a = _u64(a)
    r = np.zeros_like(a)
    f = getattr(lib(), "ho_row_" + name)
    if name in ("add", "sub", "mul"):
        b = _u64(b)
        f(_p(r), _p(a), _p(b), a.size, q)
    elif name == "neg":
        f(_p(r), _p(a), a.size, q)
    else:
        f(_p(r), _p(a), int(b), a.size, q)
    return r


def automorph(row, m, zms, k):
    row, zms = _u64(row), np.ascontiguousarray(zms, dtype=np.uint32)
    out = np.zeros_like(row)
    rc = lib().ho_row_automorph(_p(out), _p(row), m, _p(zms), row.size, k)
    if rc != 0:
        raise RuntimeError("DoubleCRT::automorph: k not in Zm*")
    return out
