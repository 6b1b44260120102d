"""Pseudo-random generator of the key material: ChaCha20 (RFC 8439) keyed from OS entropy.

The reference draws everything secret -- the secret key, RLWE errors, encryption randomness and
the `a` columns of key-switching matrices -- from NTL's RandomStream, a ChaCha20 stream seeded
from the operating system (src/keys.cpp GenKeySWmatrix: RandomBits(prgSeed, 256); NTL's own seed
expansion is not reproducible without NTL).  Here:

  * `ChaChaRng()`            -- key = os.urandom(32): the default of Sampler / SecKey / PubKey.
  * `ChaChaRng(seed=int)`    -- key = SHA-256 of the seed: DETERMINISTIC, for tests and benchmarks
                                only (anyone who knows the seed knows every key drawn from it).
  * host draws (sampleSmall / sampleHWt / sampleGaussian) come from stream 0 of the key;
  * uniform DoubleCRT rows (DoubleCRT::randomize, src/DoubleCRT.cpp:1258-1378) are filled ON THE
    DEVICE by hx_randomize from streams 1, 2, ... of the same key (`next_stream()`), one ChaCha20
    stream per row: nonce = (stream_lo, stream_hi, prime index | batch element << 16).

The block function is vectorised over blocks with numpy (uint32 arithmetic wraps).
"""
import hashlib
import os

import numpy as np

_SIGMA = np.array([0x61707865, 0x3320646e, 0x79622d32, 0x6b206574], dtype=np.uint32)


def _rotl(x, n):
    return (x << np.uint32(n)) | (x >> np.uint32(32 - n))


def _qr(x, a, b, c, d):
    x[a] += x[b]
    x[d] = _rotl(x[d] ^ x[a], 16)
    x[c] += x[d]
    x[b] = _rotl(x[b] ^ x[c], 12)
    x[a] += x[b]
    x[d] = _rotl(x[d] ^ x[a], 8)
    x[c] += x[d]
    x[b] = _rotl(x[b] ^ x[c], 7)


def chacha20_blocks(key, nonce, first_counter, nblocks):
    """`nblocks` consecutive 64-byte blocks (RFC 8439 section 2.3) as one bytes object.
    key: 32 bytes; nonce: 12 bytes or three uint32; counters first_counter, first_counter+1, ..."""
    k = np.frombuffer(bytes(key), dtype="<u4")
    nn = np.frombuffer(bytes(nonce), dtype="<u4") if isinstance(nonce, (bytes, bytearray)) else \
        np.asarray(nonce, dtype=np.uint32)
    s = np.empty((16, nblocks), dtype=np.uint32)
    s[0:4] = _SIGMA[:, None]
    s[4:12] = k[:, None]
    s[12] = (np.arange(nblocks, dtype=np.uint64) + np.uint64(first_counter)).astype(np.uint32)
    s[13:16] = nn[:, None]
    x = s.copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            _qr(x, 0, 4, 8, 12)
            _qr(x, 1, 5, 9, 13)
            _qr(x, 2, 6, 10, 14)
            _qr(x, 3, 7, 11, 15)
            _qr(x, 0, 5, 10, 15)
            _qr(x, 1, 6, 11, 12)
            _qr(x, 2, 7, 8, 13)
            _qr(x, 3, 4, 9, 14)
        x += s
    return np.ascontiguousarray(x.T).astype("<u4").tobytes()


class ChaChaRng:
    """The draws helib_amd.keys needs, over one ChaCha20 key."""

    def __init__(self, seed=None):
        if seed is None:
            self.key = os.urandom(32)
            self.deterministic = False
        elif isinstance(seed, (bytes, bytearray)) and len(seed) == 32:
            self.key = bytes(seed)
            self.deterministic = True
        else:
            self.key = hashlib.sha256(b"helib_amd deterministic test seed:" + str(int(seed)).encode()).digest()
            self.deterministic = True
        self._ctr = 0           # next block of the host stream (stream 0)
        self._buf = b""
        self._streams = 0       # device streams handed out so far

    def next_stream(self):
        """A fresh stream number for one hx_randomize call (never 0: that is the host stream)."""
        self._streams += 1
        return self._streams

    def bytes(self, n):
        while len(self._buf) < n:
            # (the block function is vectorised over blocks: large batches amortise the numpy calls --
            # the bytes of the stream are the same whatever the batch size)
            nb = max(4096, (n - len(self._buf) + 63) // 64)
            self._buf += chacha20_blocks(self.key, (0, 0, 0), self._ctr, nb)
            self._ctr += nb
        out, self._buf = self._buf[:n], self._buf[n:]
        return out

    def _u64(self, n):
        return np.frombuffer(self.bytes(8 * n), dtype="<u8")

    def random(self, size=None):
        """uniform doubles in [0, 1) with 53 random bits"""
        n = 1 if size is None else int(size)
        r = (self._u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return float(r[0]) if size is None else r

    def integers(self, low, high=None, size=None, dtype=np.int64):
        """uniform integers in [low, high): rejection sampling on the smallest covering bit mask"""
        if high is None:
            low, high = 0, low
        span = int(high) - int(low)
        assert span > 0
        n = 1 if size is None else int(np.prod(size))
        bits = max(1, (span - 1).bit_length())
        mask = np.uint64((1 << bits) - 1)
        out = np.empty(n, dtype=np.uint64)
        have = 0
        width = 1 if bits <= 8 else 2 if bits <= 16 else 4 if bits <= 32 else 8   # bytes per candidate
        while have < n:
            want = max(16, int((n - have) * 1.3))
            c = np.frombuffer(self.bytes(width * want), dtype=f"<u{width}").astype(np.uint64) & mask
            c = c[c < np.uint64(span)][:n - have]
            out[have:have + len(c)] = c
            have += len(c)
        if int(low) >= 0 and np.dtype(dtype) == np.uint64:
            res = out + np.uint64(low)
        else:
            res = (out.astype(np.int64) + np.int64(low)).astype(dtype)
        if size is None:
            return res[0].item()
        return res.reshape(size)

    def normal(self, loc=0.0, scale=1.0, size=None):
        """Box-Muller on pairs of uniforms (the method of src/sample.cpp:114-199)"""
        n = 1 if size is None else int(size)
        half = (n + 1) // 2
        u1 = 1.0 - self.random(half)        # (0, 1]
        u2 = self.random(half)
        r = np.sqrt(-2.0 * np.log(u1))
        z = np.concatenate([r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)])[:n]
        z = loc + scale * z
        return float(z[0]) if size is None else z

    def choice(self, n, size, replace=False):
        """`size` distinct positions below n in the order drawn (sampleHWt, src/sample.cpp:29-65:
        draw a position, skip it if already taken)"""
        assert not replace and size <= n
        seen, out = set(), []
        while len(out) < size:
            for v in self.integers(0, n, size=max(16, 2 * (size - len(out)))):
                v = int(v)
                if v not in seen:
                    seen.add(v)
                    out.append(v)
                    if len(out) == size:
                        break
        return np.array(out, dtype=np.int64)
