"""ctypes binding of the C ABI (include/helib_amd.h) + a thin host-side mirror of
the reference's DoubleCRT interface for this path.

The HIP extension is the only compute path: importing works without a GPU (so
that the symbol table can be checked), but every compute call needs a gfx950
device and raises otherwise.  Nothing here imports oracle/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HX_LIB selects an alternative build of the same extension (kernel A/B experiments)
_SO = os.environ.get("HX_LIB") or os.path.join(_HERE, "lib", "libhelib_amd.so")
_lib = None

HX_OK = 0
HX_ERR_INVALID, HX_ERR_DEVICE, HX_ERR_PRIMESET = -1, -2, -3
HX_ERR_NOT_IN_ZMSTAR, HX_ERR_UNSUPPORTED, HX_ERR_NOMEM = -4, -5, -6


class HxError(RuntimeError):
    """helib::RuntimeError / LogicError analogue (include/helib/exceptions.h)."""

    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class InvalidArgument(HxError):
    pass


# every symbol include/helib_amd.h declares (checked by tests without a GPU)
SYMBOLS = [
    "hx_last_error", "hx_version", "hx_device_count",
    "hx_ctx_create", "hx_ctx_destroy", "hx_ctx_phim", "hx_ctx_set_stream", "hx_ctx_sync",
    "hx_ctx_add_prime", "hx_ctx_num_primes", "hx_ctx_prime",
    "hx_poly_create", "hx_poly_create_uninit", "hx_poly_wrap", "hx_poly_destroy", "hx_poly_shape", "hx_poly_primes",
    "hx_poly_device_ptr", "hx_poly_upload", "hx_poly_download", "hx_poly_copy",
    "hx_poly_set_zero", "hx_poly_remove_primes",
    "hx_ntt_forward", "hx_ntt_inverse",
    "hx_add", "hx_sub", "hx_mul", "hx_negate", "hx_add_scalar", "hx_sub_scalar", "hx_mul_scalar",
    "hx_set_scalar", "hx_exp",
    "hx_automorph", "hx_complex_conj",
    "hx_add_primes_and_scale", "hx_add_primes", "hx_poly_rem", "hx_scale_down", "hx_scale_down_multi",
    "hx_bring_to_set_multi", "hx_break_into_digits",
    "hx_ksk_create", "hx_ksk_destroy", "hx_ksk_shape", "hx_ksk_download", "hx_tensor", "hx_key_switch_digits", "hx_mul_relin",
    "hx_relinearize",
    "hx_ctx_defer_norms", "hx_norms_flush",
    "hx_embedding_norm", "hx_scale_down_multi_norms", "hx_bring_to_set_multi_norms",
    "hx_break_into_digits_norms", "hx_relinearize_norms",
    "hx_intel_FFTFwd", "hx_intel_FFTRev1", "hx_intel_EltwiseAddMod", "hx_intel_EltwiseAddModScalar",
    "hx_intel_EltwiseSubMod", "hx_intel_EltwiseSubModScalar", "hx_intel_EltwiseMultMod",
    "hx_intel_EltwiseMultModScalar",
    "hx_time_ntt", "hx_ctx_timer_begin", "hx_ctx_timer_end", "hx_randomize",
    "hx_ctx_graph_begin", "hx_ctx_graph_end", "hx_graph_launch", "hx_graph_destroy",
    "hx_profile_begin", "hx_profile_end", "hx_ctx_arena_stats", "hx_ctx_reserve",
    "hx_tensor_bring_to_set", "hx_tensor_bring_to_set_norms", "hx_mul_relin_norms",
]


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so.7 and dlopens it by
    path; if this library (linked against /opt/rocm's copy of the same SONAME) was loaded first, the
    process ends up with two runtimes and the first one loses its devices (hipGetDeviceCount = 0).
    Loading torch's copy first -- without importing torch -- makes both resolve to one runtime
    whatever the import order.  No torch installed: nothing to do."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load the HIP extension; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError(
                f"{_SO} is missing: build it with `python -m helib_amd.build` "
                "(there is no CPU fallback)")
        _preload_hip_runtime()
        L = C.CDLL(_SO)
        L.hx_last_error.restype = C.c_char_p
        L.hx_version.restype = C.c_char_p
        L.hx_poly_device_ptr.restype = C.c_void_p
        L.hx_poly_device_ptr.argtypes = [C.c_void_p]
        vp, ip, u64 = C.c_void_p, C.c_int, C.c_uint64
        sig = {
            "hx_device_count": [vp],
            "hx_ctx_create": [vp, ip, u64], "hx_ctx_destroy": [vp], "hx_ctx_phim": [vp, vp],
            "hx_ctx_set_stream": [vp, vp], "hx_ctx_sync": [vp],
            "hx_ctx_add_prime": [vp, u64, u64, vp], "hx_ctx_num_primes": [vp, vp],
            "hx_ctx_prime": [vp, ip, vp, vp],
            "hx_poly_create": [vp, ip, vp, ip, vp], "hx_poly_create_uninit": [vp, ip, vp, ip, vp],
            "hx_poly_wrap": [vp, ip, vp, ip, vp, vp],
            "hx_poly_destroy": [vp], "hx_poly_shape": [vp, vp, vp, vp], "hx_poly_primes": [vp, vp],
            "hx_poly_upload": [vp, vp], "hx_poly_download": [vp, vp], "hx_poly_copy": [vp, vp],
            "hx_poly_set_zero": [vp], "hx_poly_remove_primes": [vp, vp, ip],
            "hx_ntt_forward": [vp], "hx_ntt_inverse": [vp],
            "hx_add": [vp, vp], "hx_sub": [vp, vp], "hx_mul": [vp, vp], "hx_negate": [vp],
            "hx_add_scalar": [vp, vp], "hx_sub_scalar": [vp, vp], "hx_mul_scalar": [vp, vp],
            "hx_set_scalar": [vp, vp], "hx_exp": [vp, u64],
            "hx_automorph": [vp, u64], "hx_complex_conj": [vp],
            "hx_add_primes_and_scale": [vp, vp, ip], "hx_add_primes": [vp, vp, ip],
            "hx_poly_rem": [vp, C.c_uint64, vp],
            "hx_scale_down": [vp, vp, ip, u64],
            "hx_scale_down_multi": [vp, ip, vp, ip, u64],
            "hx_bring_to_set_multi": [vp, ip, vp, ip, vp, ip, u64],
            "hx_break_into_digits": [vp, vp, vp, ip, vp, ip, vp],
            "hx_ksk_create": [vp, ip, vp, ip, vp, vp, vp], "hx_ksk_destroy": [vp],
            "hx_ksk_shape": [vp, vp, vp, vp], "hx_ksk_download": [vp, vp, vp],
            "hx_tensor": [vp] * 7, "hx_key_switch_digits": [vp] * 4,
            "hx_mul_relin": [vp, vp, vp, vp, vp, vp, vp, ip, vp, vp],
            "hx_relinearize": [vp, vp, vp, vp, vp, vp, ip, vp, ip, vp, vp],
            "hx_ctx_defer_norms": [vp, ip], "hx_norms_flush": [vp],
            "hx_embedding_norm": [vp, vp, ip, vp],
            "hx_scale_down_multi_norms": [vp, ip, vp, ip, u64, vp, vp],
            "hx_bring_to_set_multi_norms": [vp, ip, vp, ip, vp, ip, u64, vp],
            "hx_break_into_digits_norms": [vp, vp, vp, ip, vp, ip, vp, vp],
            "hx_relinearize_norms": [vp, vp, vp, vp, vp, vp, ip, vp, ip, vp, vp, vp],
            "hx_intel_FFTFwd": [vp, vp, C.c_long, C.c_long],
            "hx_intel_FFTRev1": [vp, vp, C.c_long, C.c_long],
            "hx_intel_EltwiseAddMod": [vp, vp, vp, C.c_long, C.c_long],
            "hx_intel_EltwiseSubMod": [vp, vp, vp, C.c_long, C.c_long],
            "hx_intel_EltwiseMultMod": [vp, vp, vp, C.c_long, C.c_long],
            "hx_intel_EltwiseAddModScalar": [vp, vp, C.c_long, C.c_long, C.c_long],
            "hx_intel_EltwiseSubModScalar": [vp, vp, C.c_long, C.c_long, C.c_long],
            "hx_intel_EltwiseMultModScalar": [vp, vp, C.c_long, C.c_long, C.c_long],
            "hx_time_ntt": [vp, ip, ip, ip, vp],
            "hx_ctx_timer_begin": [vp], "hx_ctx_timer_end": [vp, vp],
            "hx_randomize": [vp, C.c_char_p, C.c_uint64],
            "hx_ctx_graph_begin": [vp], "hx_ctx_graph_end": [vp, vp], "hx_graph_launch": [vp],
            "hx_graph_destroy": [vp],
            "hx_profile_begin": [], "hx_profile_end": [vp, C.c_size_t, vp],
            "hx_ctx_arena_stats": [vp, vp], "hx_ctx_reserve": [vp, C.c_uint64],
            "hx_mul_relin_norms": [vp, vp, vp, vp, vp, vp, vp, ip, vp, vp, vp],
            "hx_tensor_bring_to_set": [vp] * 8 + [ip, vp, ip, u64],
            "hx_tensor_bring_to_set_norms": [vp] * 8 + [ip, vp, ip, u64, vp],
        }
        for name, args in sig.items():
            f = getattr(L, name)
            f.argtypes = args
            f.restype = C.c_int
        _lib = L
    return _lib


def _chk(rc):
    if rc != HX_OK:
        msg = lib().hx_last_error().decode()
        raise (InvalidArgument if rc == HX_ERR_INVALID else HxError)(rc, msg)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    rc = lib().hx_device_count(C.byref(n))
    return n.value if rc == HX_OK else 0


class Context:
    """Context::moduli + PAlgebra tables resident on one GPU."""

    def __init__(self, m, device=0):
        self.h = C.c_void_p()
        _chk(lib().hx_ctx_create(C.byref(self.h), device, m))
        self.m = m
        n = C.c_uint64()
        _chk(lib().hx_ctx_phim(self.h, C.byref(n)))
        self.phim = int(n.value)
        self.primes, self.roots = [], []

    def close(self):
        if self.h:
            lib().hx_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_prime(self, q, root=0):
        """Cmodulus(zms, q, root) -- returns the index in Context::moduli."""
        idx = C.c_int()
        _chk(lib().hx_ctx_add_prime(self.h, q, root, C.byref(idx)))
        qq, rr = C.c_uint64(), C.c_uint64()
        _chk(lib().hx_ctx_prime(self.h, idx.value, C.byref(qq), C.byref(rr)))
        self.primes.append(int(qq.value))
        self.roots.append(int(rr.value))
        return idx.value

    def ithPrime(self, i):
        return self.primes[i]

    def set_stream(self, stream_ptr):
        _chk(lib().hx_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def timerBegin(self):
        """HIP event on this context's stream (hx_ctx_timer_begin)."""
        _chk(lib().hx_ctx_timer_begin(self.h))

    def timerEnd(self):
        """Milliseconds of device time since timerBegin (waits for the closing event)."""
        ms = C.c_float()
        _chk(lib().hx_ctx_timer_end(self.h, C.byref(ms)))
        return ms.value

    def reserve(self, nbytes):
        """Reserve device memory for this context's slabs up front (hx_ctx_reserve)."""
        _chk(lib().hx_ctx_reserve(self.h, int(nbytes)))

    def arenaStats(self):
        """Device memory behind this context's DoubleCRT slabs (hx_ctx_arena_stats): bytes reserved from
        hipMalloc, bytes in use, number of hipMalloc calls so far, blocks parked for live HIP graphs."""
        v = (C.c_uint64 * 4)()
        _chk(lib().hx_ctx_arena_stats(self.h, v))
        return {"reserved": int(v[0]), "in_use": int(v[1]), "sys_calls": int(v[2]), "deferred": int(v[3])}

    def graphBegin(self):
        """Start recording everything enqueued on this context into a HIP graph (hx_ctx_graph_begin):
        the calls return as usual, nothing runs.  Run the sequence once eagerly first; no uploads,
        downloads or measured-noise read-backs inside."""
        _chk(lib().hx_ctx_graph_begin(self.h))

    def graphEnd(self):
        """Close the recording; returns a Graph whose launch() replays it with one launch."""
        g = C.c_void_p()
        _chk(lib().hx_ctx_graph_end(self.h, C.byref(g)))
        return Graph(self, g)

    def deferNorms(self, on):
        """Deferred read-back of the measured-noise norms (hx_ctx_defer_norms)."""
        if bool(on) != getattr(self, "_defer", False):
            _chk(lib().hx_ctx_defer_norms(self.h, 1 if on else 0))   # switching off flushes
            self._defer = bool(on)
            if not on:
                self._deferred = []

    def keepUntilFlush(self, out):
        """The library writes a deferred norm into the caller's array when the context is next
        flushed: the array must outlive that, whatever becomes of the ciphertext that asked for it
        (a result dropped without reading its noise estimate used to leave a dangling pointer)."""
        if getattr(self, "_defer", False):
            if not hasattr(self, "_deferred"):
                self._deferred = []
            self._deferred.append(out)
            if len(self._deferred) > 256:     # nobody is reading: complete them, keep the list short
                self.flushNorms()
        return out

    def flushNorms(self):
        _chk(lib().hx_norms_flush(self.h))
        self._deferred = []

    def sync(self):
        _chk(lib().hx_ctx_sync(self.h))


class Graph:
    """A captured sequence of engine calls (hx_graph): launch() re-executes the same kernels on the
    same buffers -- the inputs are whatever the input polys hold now, the results land in the polys
    the recorded calls produced."""

    def __init__(self, ctx, handle):
        self.context, self.h = ctx, handle

    def launch(self):
        _chk(lib().hx_graph_launch(self.h))
        return self

    def destroy(self):
        if self.h:
            lib().hx_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class DoubleCRT:
    """Batched DoubleCRT (include/helib/DoubleCRT.h:212-385 for the ops of this path).

    rows are numpy uint64 arrays of shape [nrows, batch, phim] on the host side."""

    def __init__(self, context, index_set, batch=1, data=None, zero=True):
        self.context = context
        idx = _i32(list(index_set))
        self.h = C.c_void_p()
        create = lib().hx_poly_create if (zero and data is None) else lib().hx_poly_create_uninit
        _chk(create(context.h, batch, _p(idx), len(idx), C.byref(self.h)))
        self.batch = batch
        if data is not None:
            self.upload(data)

    @classmethod
    def wrap(cls, context, index_set, batch, device_ptr):
        """hx_poly_wrap: a DoubleCRT over caller-owned device memory ([nrows][batch][phim] u64 at
        device_ptr, e.g. a torch tensor's data_ptr()); the rows never move out of that buffer and
        the caller keeps it alive.  Operations that need more rows than it holds fail."""
        self = cls.__new__(cls)
        self.context, self.batch = context, batch
        idx = _i32(list(index_set))
        self.h = C.c_void_p()
        _chk(lib().hx_poly_wrap(context.h, batch, _p(idx), len(idx), C.c_void_p(int(device_ptr)), C.byref(self.h)))
        return self

    def close(self):
        if self.h:
            lib().hx_poly_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- storage ---
    def getIndexSet(self):
        n = C.c_int()
        _chk(lib().hx_poly_shape(self.h, None, C.byref(n), None))
        out = np.zeros(max(n.value, 1), dtype=np.int32)
        _chk(lib().hx_poly_primes(self.h, _p(out)))
        return [int(x) for x in out[:n.value]]

    def upload(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        nrows = len(self.getIndexSet())
        assert rows.size == nrows * self.batch * self.context.phim, rows.shape
        _chk(lib().hx_poly_upload(self.h, _p(rows)))
        return self

    def download(self):
        nrows = len(self.getIndexSet())
        out = np.zeros((nrows, self.batch, self.context.phim), dtype=np.uint64)
        _chk(lib().hx_poly_download(self.h, _p(out)))
        return out

    def randomize(self, key32, stream):
        """DoubleCRT::randomize (src/DoubleCRT.cpp:1258-1378) on the device: uniform rows by the
        reference's rejection sampling from the ChaCha20 stream (key32, stream) -- hx_randomize."""
        key32 = bytes(key32)
        assert len(key32) == 32
        _chk(lib().hx_randomize(self.h, key32, int(stream)))
        return self

    def copy(self):
        # (a digit block lists its primes once per digit: hx_poly_copy resizes the destination and
        # takes over the source's row list, the destination only has to exist)
        o = DoubleCRT(self.context, list(dict.fromkeys(self.getIndexSet())), self.batch, zero=False)
        _chk(lib().hx_poly_copy(o.h, self.h))
        return o

    def device_ptr(self):
        return lib().hx_poly_device_ptr(self.h)

    # --- transforms (Cmodulus::FFT / iFFT over all rows) ---
    def FFT(self):
        _chk(lib().hx_ntt_forward(self.h))
        return self

    def iFFT(self):
        _chk(lib().hx_ntt_inverse(self.h))
        return self

    # --- ring ops ---
    def __iadd__(self, o):
        _chk(lib().hx_add(self.h, o.h))
        return self

    def __isub__(self, o):
        _chk(lib().hx_sub(self.h, o.h))
        return self

    def __imul__(self, o):
        _chk(lib().hx_mul(self.h, o.h))
        return self

    def Negate(self):
        _chk(lib().hx_negate(self.h))
        return self

    def _scalars(self, num):
        idx = self.getIndexSet()
        if np.isscalar(num) or isinstance(num, int):
            vals = [int(num) % self.context.primes[i] for i in idx]
        else:
            vals = [int(v) for v in num]
        return np.array(vals, dtype=np.uint64)

    def addConstant(self, num):
        s = self._scalars(num)
        _chk(lib().hx_add_scalar(self.h, _p(s)))
        return self

    def subConstant(self, num):
        s = self._scalars(num)
        _chk(lib().hx_sub_scalar(self.h, _p(s)))
        return self

    def mulConstant(self, num):
        s = self._scalars(num)
        _chk(lib().hx_mul_scalar(self.h, _p(s)))
        return self

    def setConstant(self, num):
        """DoubleCRT::operator=(ZZ): every entry becomes num mod q_i."""
        s = self._scalars(num)
        _chk(lib().hx_set_scalar(self.h, _p(s)))
        return self

    def Exp(self, e):
        """DoubleCRT::Exp: entry-wise PowerMod(x, e, q_i) for e >= 0."""
        if e < 0:
            raise ValueError("negative exponent")
        _chk(lib().hx_exp(self.h, int(e)))
        return self

    def automorph(self, k):
        _chk(lib().hx_automorph(self.h, k))
        return self

    def complexConj(self):
        _chk(lib().hx_complex_conj(self.h))
        return self

    # --- prime-set operations ---
    def removePrimes(self, s):
        s = _i32(list(s))
        _chk(lib().hx_poly_remove_primes(self.h, _p(s), len(s)))
        return self

    def addPrimesAndScale(self, s):
        s = _i32(list(s))
        _chk(lib().hx_add_primes_and_scale(self.h, _p(s), len(s)))
        return self

    def addPrimes(self, s):
        s = _i32(list(s))
        _chk(lib().hx_add_primes(self.h, _p(s), len(s)))
        return self

    def toPolyMod(self, t):
        """toPoly + PolyRed(t, abs=true) on the device: [batch, phim] residues in [0,t) of the
        centred coefficients (the tail of SecKey::Decrypt)."""
        out = np.zeros((self.batch, self.context.phim), dtype=np.uint64)
        _chk(lib().hx_poly_rem(self.h, int(t), _p(out)))
        return out

    def scaleDownToSet(self, keep_set, ptxtSpace, norms=False):
        """norms=True: returns embeddingLargestCoeff(delta/diffProd) per batch element instead of
        self (see scaleDownToSetMulti)."""
        if norms:
            return scaleDownToSetMulti([self], keep_set, ptxtSpace, norms=True)[0]
        drop = _i32([i for i in self.getIndexSet() if i not in set(keep_set)])
        _chk(lib().hx_scale_down(self.h, _p(drop), len(drop), ptxtSpace))
        return self

    def breakIntoDigits(self, digits, special, norms=False):
        dig_idx = _i32([p for d in digits for p in d])
        dig_off = _i32(np.concatenate([[0], np.cumsum([len(d) for d in digits])]))
        sp = _i32(list(special))
        out = DoubleCRT(self.context, self.getIndexSet(), self.batch, zero=False)
        if not norms:
            _chk(lib().hx_break_into_digits(self.h, _p(dig_idx), _p(dig_off), len(digits), _p(sp),
                                            len(sp), out.h))
            return out
        nrm = np.zeros((len(digits), self.batch), dtype=np.float64)
        self.context.deferNorms(False)
        _chk(lib().hx_break_into_digits_norms(self.h, _p(dig_idx), _p(dig_off), len(digits), _p(sp),
                                              len(sp), out.h, _p(nrm)))
        return out, nrm


class KeySwitch:
    """KeySwitch matrix W (include/helib/keySwitching.h:86-101) with explicit (b, a)."""

    def __init__(self, context, row_idx, b, a):
        b = np.ascontiguousarray(b, dtype=np.uint64)
        a = np.ascontiguousarray(a, dtype=np.uint64)
        idx = _i32(list(row_idx))
        self.ndig = b.shape[0]
        self.h = C.c_void_p()
        self.context = context
        self.row_idx = [int(i) for i in row_idx]
        _chk(lib().hx_ksk_create(context.h, self.ndig, _p(idx), len(idx), _p(b), _p(a),
                                 C.byref(self.h)))

    def download(self):
        """(b, a) back on the host, [ndig][nrows][phim] (hx_ksk_download)"""
        shape = (self.ndig, len(self.row_idx), self.context.phim)
        b, a = np.empty(shape, dtype=np.uint64), np.empty(shape, dtype=np.uint64)
        _chk(lib().hx_ksk_download(self.h, _p(b), _p(a)))
        return b, a

    def __del__(self):
        try:
            if self.h:
                lib().hx_ksk_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass


def tensorProduct(c0, c1, d0, d1):
    ctx = c0.context
    outs = [DoubleCRT(ctx, c0.getIndexSet(), c0.batch, zero=False) for _ in range(3)]
    _chk(lib().hx_tensor(c0.h, c1.h, d0.h, d1.h, outs[0].h, outs[1].h, outs[2].h))
    return outs


def keySwitchDigits(digits, W, out0, out1):
    """Ctxt::keySwitchDigits: out0 += sum_d digit_d*b_d, out1 += sum_d digit_d*a_d (out0/out1 on the
    ciphertext's primes followed by the special primes; digits as breakIntoDigits returns them)."""
    _chk(lib().hx_key_switch_digits(digits.h, W.h, out0.h, out1.h))


def breakIntoDigits(part, digits, special, norms=False):
    """DoubleCRT::breakIntoDigits as a backend entry point of helib_amd.ctxt (hoisting): the digit
    block (and, norms=True, embeddingLargestCoeff(digit)/P_d per digit and batch element)."""
    return part.breakIntoDigits(digits, special, norms)


def zerosLike(poly):
    return DoubleCRT(poly.context, poly.getIndexSet(), poly.batch)


def multiplyBy(c0, c1, d0, d1, W, digits, out0=None, out1=None):
    """Ctxt::multiplyBy data path at a fixed level (tensorProduct + reLinearize)."""
    ctx = c0.context
    dig_idx = _i32([p for d in digits for p in d])
    dig_off = _i32(np.concatenate([[0], np.cumsum([len(d) for d in digits])]))
    if out0 is None:
        out0 = DoubleCRT(ctx, c0.getIndexSet(), c0.batch, zero=False)
        out1 = DoubleCRT(ctx, c0.getIndexSet(), c0.batch, zero=False)
    _chk(lib().hx_mul_relin(c0.h, c1.h, d0.h, d1.h, W.h, _p(dig_idx), _p(dig_off), len(digits),
                            out0.h, out1.h))
    return out0, out1


def scaleDownToSetMulti(polys, keep_set, ptxtSpace, norms=False, fdelta=False, defer=False):
    """DoubleCRT::scaleDownToSet on several parts that share one prime set, batched into one
    pair of launches where possible.  norms=True additionally returns
    embeddingLargestCoeff(delta/diffProd) per (part, batch element) -- the measured mod-switch
    noise of Ctxt::modDownToSet (src/Ctxt.cpp:466-507) -- as an [nparts, batch] array
    (and the fdelta coefficients [nparts, batch, phi(m)] when fdelta=True)."""
    polys = list(polys)
    keep = set(keep_set)
    drop = _i32([i for i in polys[0].getIndexSet() if i not in keep])
    arr = (C.c_void_p * len(polys))(*[p.h for p in polys])
    if not norms:
        _chk(lib().hx_scale_down_multi(arr, len(polys), _p(drop), len(drop), ptxtSpace))
        return None
    out = np.zeros((len(polys), polys[0].batch), dtype=np.float64)
    fd = np.zeros((len(polys), polys[0].batch, polys[0].context.phim), dtype=np.float64) if fdelta else None
    polys[0].context.deferNorms(defer)   # defer=True: `out` is filled by normsFlush()
    polys[0].context.keepUntilFlush(out)
    _chk(lib().hx_scale_down_multi_norms(arr, len(polys), _p(drop), len(drop), ptxtSpace, _p(out),
                                         _p(fd) if fdelta else None))
    return (out, fd) if fdelta else out


def bringToSetMulti(polys, add_set, keep_set, ptxtSpace, norms=False, defer=False):
    """Ctxt::bringToSet on several parts sharing one prime set: mod-up by add_set, then mod-down
    to keep_set (fused into one pair of launches when a single prime is dropped).
    norms=True: also the measured mod-down noise, as in scaleDownToSetMulti."""
    polys = list(polys)
    add = _i32(list(add_set))
    keep = set(keep_set)
    cur = polys[0].getIndexSet() + [int(i) for i in add]
    drop = _i32([i for i in cur if i not in keep])
    arr = (C.c_void_p * len(polys))(*[p.h for p in polys])
    if not norms:
        _chk(lib().hx_bring_to_set_multi(arr, len(polys), _p(add), len(add), _p(drop), len(drop), ptxtSpace))
        return None
    out = np.zeros((len(polys), polys[0].batch), dtype=np.float64)
    polys[0].context.deferNorms(defer)
    polys[0].context.keepUntilFlush(out)
    _chk(lib().hx_bring_to_set_multi_norms(arr, len(polys), _p(add), len(add), _p(drop), len(drop),
                                           ptxtSpace, _p(out)))
    return out


def normsFlush(poly):
    """Complete every deferred norms read-back of poly's context (waits for the norm kernels
    only)."""
    poly.context.flushNorms()


def supportsNorms(m):
    """Measured (PGFFT-style) noise norms run on the device: power-of-two m (one N/2-point complex
    transform) and general m up to 131072 (complex-double Bluestein)."""
    return m >= 2 and ((m & (m - 1)) == 0 or m <= 131072)


def embeddingLargestCoeff(context, f):
    """embeddingLargestCoeff (src/norms.cpp:480-493) of real polynomials f[rows, phi(m)] on the
    device (m a power of two)."""
    f = np.ascontiguousarray(f, dtype=np.float64).reshape(-1, context.phim)
    out = np.zeros(f.shape[0], dtype=np.float64)
    _chk(lib().hx_embedding_norm(context.h, _p(f), f.shape[0], _p(out)))
    return out


def reLinearize(t0, t1, t2, W, digits, special, out0=None, out1=None, norms=False, defer=False):
    """Ctxt::reLinearize data path for a 3-part ciphertext (1, s, s^2), or with t1 = None for the
    (1, s(X^k)) ciphertext of Ctxt::smartAutomorph (t2 = the s(X^k) part).  norms=True also returns
    the [ndigits, batch] array embeddingLargestCoeff(digit)/P_digit (the pieces of
    breakIntoDigits' return value, src/DoubleCRT.cpp:538-545)."""
    ctx = t0.context
    dig_idx = _i32([p for d in digits for p in d])
    dig_off = _i32(np.concatenate([[0], np.cumsum([len(d) for d in digits])]))
    sp = _i32(list(special))
    if out0 is None:
        out0 = DoubleCRT(ctx, t0.getIndexSet(), t0.batch, zero=False)
        out1 = DoubleCRT(ctx, t0.getIndexSet(), t0.batch, zero=False)
    if not norms:
        _chk(lib().hx_relinearize(t0.h, t1.h if t1 is not None else None, t2.h, W.h, _p(dig_idx),
                                  _p(dig_off), len(digits), _p(sp), len(sp), out0.h, out1.h))
        return out0, out1
    nrm = np.zeros((len(digits), t0.batch), dtype=np.float64)
    ctx.deferNorms(defer)
    ctx.keepUntilFlush(nrm)
    _chk(lib().hx_relinearize_norms(t0.h, t1.h if t1 is not None else None, t2.h, W.h, _p(dig_idx),
                                    _p(dig_off), len(digits), _p(sp), len(sp), out0.h, out1.h, _p(nrm)))
    return out0, out1, nrm


def time_ntt(poly, inverse, iters, max_rows=0):
    ms = C.c_float()
    _chk(lib().hx_time_ntt(poly.h, 1 if inverse else 0, iters, max_rows, C.byref(ms)))
    return ms.value


def profileBegin():
    """Start timing every kernel the library launches with HIP events on its own stream (hx_profile_begin)."""
    _chk(lib().hx_profile_begin())


def profileEnd():
    """Wait for the launches recorded since profileBegin and return the per-kernel summary
    (hx_profile_end): {"launches", "dropped", "kernels": [{"kernel", "workgroups", "workgroup_size",
    "calls", "total_us", "avg_us", "min_us", "max_us"}, ...]} ordered by total time."""
    import json
    need = C.c_size_t()
    _chk(lib().hx_profile_end(None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    _chk(lib().hx_profile_end(buf, need.value, None))
    return json.loads(buf.value.decode())


def tensorBringToSet(c0, c1, d0, d1, add_set, keep_set, ptxtSpace, norms=False, defer=False):
    """Ctxt::tensorProduct followed by Ctxt::bringToSet of the three product parts (hx_tensor_bring_to_set): the
    parts (1), (s), (s^2) on `keep_set` (= the operands' primes + add_set - what is dropped).  norms: also the
    embeddingLargestCoeff of the mod-switch deltas, [3][batch] (deferred read-back as in bringToSetMulti)."""
    ctx = c0.context
    cur = c0.getIndexSet()
    add = [i for i in add_set if i not in cur]
    drop = [i for i in cur + add if i not in set(keep_set)]
    outs = [DoubleCRT(ctx, cur, c0.batch, zero=False) for _ in range(3)]
    a, d = _i32(add), _i32(drop)
    if not norms:
        _chk(lib().hx_tensor_bring_to_set(c0.h, c1.h, d0.h, d1.h, outs[0].h, outs[1].h, outs[2].h, _p(a), len(add),
                                          _p(d), len(drop), int(ptxtSpace)))
        return outs
    nrm = np.zeros((3, c0.batch), dtype=np.float64)
    ctx.deferNorms(defer)
    ctx.keepUntilFlush(nrm)
    _chk(lib().hx_tensor_bring_to_set_norms(c0.h, c1.h, d0.h, d1.h, outs[0].h, outs[1].h, outs[2].h, _p(a), len(add),
                                            _p(d), len(drop), int(ptxtSpace), _p(nrm)))
    return outs, nrm


def mulRelin(c0, c1, d0, d1, W, digits, norms=False, defer=False):
    """Ctxt::tensorProduct + Ctxt::reLinearize at the full level of the matrix W with the product parts formed inside
    the key-switch kernels (hx_mul_relin[_norms]); the operands' primes must be W's leading rows.  norms=True also
    returns the [ndigits, batch] array of reLinearize (embeddingLargestCoeff(digit) / P_digit)."""
    ctx = c0.context
    dig_idx = _i32([p for d in digits for p in d])
    dig_off = _i32(np.concatenate([[0], np.cumsum([len(d) for d in digits])]))
    out0 = DoubleCRT(ctx, c0.getIndexSet(), c0.batch, zero=False)
    out1 = DoubleCRT(ctx, c0.getIndexSet(), c0.batch, zero=False)
    if not norms:
        _chk(lib().hx_mul_relin(c0.h, c1.h, d0.h, d1.h, W.h, _p(dig_idx), _p(dig_off), len(digits), out0.h, out1.h))
        return out0, out1
    nrm = np.zeros((len(digits), c0.batch), dtype=np.float64)
    ctx.deferNorms(defer)
    ctx.keepUntilFlush(nrm)
    _chk(lib().hx_mul_relin_norms(c0.h, c1.h, d0.h, d1.h, W.h, _p(dig_idx), _p(dig_off), len(digits), out0.h, out1.h,
                                  _p(nrm)))
    return out0, out1, nrm
