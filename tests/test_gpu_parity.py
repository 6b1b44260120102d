"""GPU parity tests: the HIP path (through the C ABI, include/helib_amd.h) against the
CPU oracle on the same seeded inputs.  Bit-exact: every word is compared.
Run with `pytest -m gpu` on an MI355X."""
import os

import math
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hx():
    # Import order matters for the tests that use torch (a caller-owned device buffer): PyTorch registers its GPU code
    # objects with the HIP runtime lazily only while that runtime has not been initialised.  Imported AFTER this
    # library has touched the device, every one of its fat binaries is unpacked on the spot -- on a fresh box, with a
    # cold comgr cache (~/.cache/comgr, 1.4 GB), that took more than five minutes inside one test (round 5, measured:
    # tools/torch_import_pytest_matrix.sh).  So: torch first, where there is a torch.  (bench.py and smoke() import
    # it first anyway; helib_amd itself never imports torch.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    from helib_amd import capi
    if capi.device_count() <= 0:    # a plain `pytest tests` on a CPU box: the device tests are skipped, not errors
        pytest.skip("no HIP device: the GPU parity tests run on an MI355X (pytest -m gpu)")
    return capi


class Pair:
    """A device context and an oracle context with the same primes/roots."""

    def __init__(self, hx, m, primes):
        self.hx = hx
        self.g = hx.Context(m)
        self.o = O.Ctx(m)
        for q in primes:
            i = self.o.add_prime(q)
            j = self.g.add_prime(q, self.o.roots[i])   # host supplies the root (fact 4)
            assert i == j
        self.N = self.o.N
        self.primes = list(primes)

    def rand(self, idx, seed, batch=1):
        """[nrows, batch, N] uniform residues"""
        out = np.zeros((len(idx), batch, self.N), dtype=np.uint64)
        for r, i in enumerate(idx):
            for b in range(batch):
                out[r, b] = O.fill_uniform(self.N, self.primes[i], seed * 100003 + i * 131 + b)
        return out


def primes_for(m, n, bits=60):
    g = O.PrimeGen(bits, m)
    return [g.next() for _ in range(n)]


# ---------------------------------------------------------------- config 1: fft_bench
@pytest.mark.parametrize("m", [16384, 32768, 65536])
def test_ntt_single_prime_like_fft_bench(hx, m):
    # benchmarks/fft_bench.cpp:24-73: one 49-bit prime, forward on random / monomial,
    # inverse on y[i] = i
    q = O.PrimeGen(49, m).next()
    P = Pair(hx, m, [q])
    N = P.N
    x = P.rand([0], 1)
    d = hx.DoubleCRT(P.g, [0], 1, x)
    got = d.FFT().download()
    assert np.array_equal(got[0, 0], P.o.fft([0], x[:, 0])[0])
    assert np.array_equal(d.iFFT().download(), x)
    mono = np.zeros((1, 1, N), dtype=np.uint64)
    mono[0, 0, N - 1] = 1
    d.upload(mono)
    assert np.array_equal(d.FFT().download()[0, 0], P.o.fft([0], mono[:, 0])[0])
    y = (np.arange(N, dtype=np.uint64) % np.uint64(q)).reshape(1, 1, N)
    d.upload(y)
    assert np.array_equal(d.iFFT().download()[0, 0], P.o.ifft([0], y[:, 0])[0])


@pytest.mark.parametrize("m,L,batch", [(32768, 16, 3), (16384, 5, 2), (65536, 4, 2),
                                       # N = 2^16 .. 2^19: radix-4 / -8 / -16 split into row-kernel sub-transforms
                                       (131072, 3, 2), (262144, 2, 2), (524288, 2, 1), (1048576, 2, 1)])
def test_ntt_doublecrt_batched(hx, m, L, batch):
    P = Pair(hx, m, primes_for(m, L))
    idx = list(range(L))
    x = P.rand(idx, 7, batch)
    d = hx.DoubleCRT(P.g, idx, batch, x)
    got = d.FFT().download()
    for b in range(batch):
        assert np.array_equal(got[:, b], P.o.fft(idx, x[:, b]))
    back = d.iFFT().download()
    assert np.array_equal(back, x)


def test_ntt_edge_values(hx):
    m = 32768
    P = Pair(hx, m, primes_for(m, 2))
    q0, q1 = P.primes
    x = np.zeros((2, 1, P.N), dtype=np.uint64)
    x[0, 0, :] = q0 - 1          # maximal residues
    x[1, 0, ::2] = q1 - 1
    d = hx.DoubleCRT(P.g, [0, 1], 1, x)
    assert np.array_equal(d.FFT().download()[:, 0], P.o.fft([0, 1], x[:, 0]))
    z = np.zeros_like(x)
    d.upload(z)
    assert not d.FFT().download().any()


# ---------------------------------------------------------------- round 5: Proth-form rows
@pytest.mark.parametrize("m", [16384, 32768, 65536])
def test_proth_and_shoup_rows_in_one_launch_and_across_contexts(hx, m, monkeypatch):
    """Rows of primes q = 1 (mod 2^32) run the Proth-form butterflies (word-wise Montgomery products on 8-byte table
    entries, helib_amd/csrc/ntt_core.h: ArProth), rows of any other prime the Shoup butterflies; the choice is per
    ROW.  One DoubleCRT over a 60-bit, a 31-bit (q < 2^32: never of the form), a 56-bit and a 45-bit prime is
    transformed by one launch that mixes both kinds, forward, inverse, and through the fused single-prime
    mod-switch (whose dropped row's last inverse stage carries the folded mod-up factor in the prime's own
    arithmetic) -- every word against the oracle; then the same on a context created under HX_NO_PROTH=1
    (all rows Shoup): identical words."""
    qs = [O.PrimeGen(60, m).next(), O.PrimeGen(31, m).next(), O.PrimeGen(56, m).next(), O.PrimeGen(45, m).next(),
          primes_for(m, 2, 60)[1]]
    assert [q & 0xffffffff == 1 for q in qs] == [True, False, True, True, True]
    outs = []
    for no_proth in (False, True):
        if no_proth:
            monkeypatch.setenv("HX_NO_PROTH", "1")
        else:
            monkeypatch.delenv("HX_NO_PROTH", raising=False)
        P = Pair(hx, m, qs)
        idx = list(range(len(qs)))
        x = P.rand(idx, 77, batch=3)
        x[0, 0, :] = qs[0] - 1
        x[2, 1, ::3] = 0
        d = hx.DoubleCRT(P.g, idx, 3, x)
        got = d.FFT().download()
        for b in range(3):
            assert np.array_equal(got[:, b], P.o.fft(idx, x[:, b]))
        assert np.array_equal(d.iFFT().download(), x)
        res = [got]
        # fused mod-switch: add one prime's worth of scale (mod-up folded into the dropped row's last stage), drop one
        for drop, keep_extra in ((0, True), (1, True), (2, False)):
            ev = hx.DoubleCRT(P.g, idx[:4], 3, got[:4].copy())
            keep = [i for i in idx[:4] if i != drop]
            if keep_extra:
                hx.bringToSetMulti([ev], [4], keep + [4], 65537)
                up = np.stack([np.vstack([P.o.scale_by_primes(idx[:4], got[:4, b], [4]),
                                          np.zeros((1, P.N), dtype=np.uint64)]) for b in range(3)], axis=1)
                src_idx, want_keep = idx[:4] + [4], keep + [4]
            else:
                ev.scaleDownToSet(keep, 65537)
                up, src_idx, want_keep = got[:4], idx[:4], keep
            gi = ev.getIndexSet()
            assert sorted(gi) == sorted(want_keep)
            g2 = ev.download()
            for b in range(3):
                want = P.o.scale_down(src_idx, up[:, b], [drop], 65537)
                for r, i in enumerate(gi):
                    assert np.array_equal(g2[r, b], want[want_keep.index(i)]), (no_proth, drop, i)
            res.append((gi, g2))
        outs.append(res)
    assert np.array_equal(outs[0][0], outs[1][0])
    for (ia, ga), (ib, gb) in zip(outs[0][1:], outs[1][1:]):
        assert ia == ib and np.array_equal(ga, gb)


# ---------------------------------------------------------------- config 2: add / mul
def test_doublecrt_add_mul_m32768_L16(hx):
    m, L = 32768, 16
    P = Pair(hx, m, primes_for(m, L))
    idx = list(range(L))
    a, b = P.rand(idx, 1), P.rand(idx, 2)
    da, db = hx.DoubleCRT(P.g, idx, 1, a), hx.DoubleCRT(P.g, idx, 1, b)
    da += db
    want = np.stack([O.row_op("add", a[r, 0], b[r, 0], P.primes[r]) for r in idx])
    assert np.array_equal(da.download()[:, 0], want)
    da *= db
    want = np.stack([O.row_op("mul", want[r], b[r, 0], P.primes[r]) for r in idx])
    assert np.array_equal(da.download()[:, 0], want)
    da -= db
    want = np.stack([O.row_op("sub", want[r], b[r, 0], P.primes[r]) for r in idx])
    assert np.array_equal(da.download()[:, 0], want)
    da.Negate()
    want = np.stack([O.row_op("neg", want[r], None, P.primes[r]) for r in idx])
    assert np.array_equal(da.download()[:, 0], want)


def test_elementwise_subset_broadcast_and_scalars(hx):
    m = 16384
    P = Pair(hx, m, primes_for(m, 5))
    a = P.rand([1, 3], 4, batch=3)
    b = P.rand([0, 1, 2, 3], 5, batch=1)          # superset, broadcast over the batch
    da = hx.DoubleCRT(P.g, [1, 3], 3, a)
    db = hx.DoubleCRT(P.g, [0, 1, 2, 3], 1, b)
    da *= db
    for r, (pi, br) in enumerate([(1, 1), (3, 3)]):
        for bb in range(3):
            assert np.array_equal(da.download()[r, bb],
                                  O.row_op("mul", a[r, bb], b[br, 0], P.primes[pi]))
    cur = da.download()
    big = (1 << 200) + 12345
    da.mulConstant(big)
    da.addConstant(7)
    da.subConstant(big)
    got = da.download()
    for r, pi in enumerate([1, 3]):
        q = P.primes[pi]
        for bb in range(3):
            w = O.row_op("mul_scalar", cur[r, bb], big % q, q)
            w = O.row_op("add_scalar", w, 7, q)
            w = O.row_op("sub_scalar", w, big % q, q)
            assert np.array_equal(got[r, bb], w)
    # DoubleCRT::Op throws when this.set is not a subset of other.set
    dc = hx.DoubleCRT(P.g, [1, 4], 3)
    with pytest.raises(hx.HxError) as ei:
        dc += db
    assert ei.value.code == hx.HX_ERR_PRIMESET


@pytest.mark.parametrize("m,k", [(32768, 3), (32768, 32767), (16384, 5 * 5 * 5)])
def test_automorph_pow2(hx, m, k):
    P = Pair(hx, m, primes_for(m, 3))
    idx = [0, 1, 2]
    a = P.rand(idx, 9, batch=2)
    d = hx.DoubleCRT(P.g, idx, 2, a)
    got = d.automorph(k).download()
    zms = O.zmstar(m)
    for r in idx:
        for b in range(2):
            assert np.array_equal(got[r, b], O.automorph(a[r, b], m, zms, k))
    with pytest.raises(hx.HxError) as ei:
        d.automorph(4)
    assert ei.value.code == hx.HX_ERR_NOT_IN_ZMSTAR
    # complexConj == row reversal (src/DoubleCRT.cpp:1240-1255)
    d.upload(a)
    assert np.array_equal(d.complexConj().download(), a[:, :, ::-1])


# ---------------------------------------------------------------- exact RNS pieces
def setup_rns(hx, m=16384, L=5, K=2, bits=60):
    P = Pair(hx, m, primes_for(m, L + K, bits))
    own, sp = list(range(L)), list(range(L, L + K))
    return P, own, sp


def test_add_primes_and_scale_and_add_primes(hx):
    P, own, sp = setup_rns(hx)
    a = P.rand(own, 3, batch=2)
    d = hx.DoubleCRT(P.g, own, 2, a)
    d.addPrimesAndScale(sp)
    got = d.download()
    assert d.getIndexSet() == own + sp
    for b in range(2):
        assert np.array_equal(got[:len(own), b], P.o.scale_by_primes(own, a[:, b], sp))
    assert not got[len(own):].any()
    d2 = hx.DoubleCRT(P.g, own[:3], 2, a[:3])
    d2.addPrimes([3, 5])
    got = d2.download()
    assert d2.getIndexSet() == [0, 1, 2, 3, 5]
    for b in range(2):
        assert np.array_equal(got[:3, b], a[:3, b])
        assert np.array_equal(got[3:, b], P.o.add_primes(own[:3], a[:3, b], [3, 5]))
    with pytest.raises(hx.HxError):
        d2.addPrimes([1])


@pytest.mark.parametrize("ptxt", [65537, 2, 1, 4])
def test_scale_down_to_set(hx, ptxt):
    P, own, sp = setup_rns(hx, L=5, K=2)
    allp = own + sp
    a = P.rand(allp, 5, batch=2)
    d = hx.DoubleCRT(P.g, allp, 2, a)
    d.scaleDownToSet(own, ptxt)             # drop the special primes
    got = d.download()
    assert d.getIndexSet() == own
    for b in range(2):
        assert np.array_equal(got[:, b], P.o.scale_down(allp, a[:, b], sp, ptxt))
    # dropping ctxt primes from the middle
    d = hx.DoubleCRT(P.g, own, 1, a[:5, :1])
    d.scaleDownToSet([0, 2, 4], ptxt)
    assert np.array_equal(d.download()[:, 0], P.o.scale_down(own, a[:5, 0], [1, 3], ptxt))


@pytest.mark.parametrize("digits", [[[0, 1], [2, 3], [4]], [[0, 1, 2, 3, 4]], [[0], [1], [2], [3], [4]]])
def test_break_into_digits(hx, digits):
    P, own, sp = setup_rns(hx)
    a = P.rand(own, 6, batch=2)
    d = hx.DoubleCRT(P.g, own, 2, a)
    dg = d.breakIntoDigits(digits, sp)
    got = dg.download()                      # [ndig*nall, batch, N]
    nall = len(own) + len(sp)
    for b in range(2):
        want = P.o.break_into_digits(own, a[:, b], digits, own + sp)
        assert np.array_equal(got[:, b].reshape(len(digits), nall, P.N), want)


def test_tensor_and_keyswitch(hx):
    P, own, sp = setup_rns(hx)
    allp = own + sp
    c0, c1, d0, d1 = (P.rand(own, s, batch=2) for s in (1, 2, 3, 4))
    G = [hx.DoubleCRT(P.g, own, 2, x) for x in (c0, c1, d0, d1)]
    t = hx.tensorProduct(*G)
    for b in range(2):
        w = P.o.tensor(own, c0[:, b], c1[:, b], d0[:, b], d1[:, b])
        for i in range(3):
            assert np.array_equal(t[i].download()[:, b], w[i])
    digits = [[0, 1], [2, 3], [4]]
    kb = np.stack([P.rand(allp, 20 + i)[:, 0] for i in range(3)])
    ka = np.stack([P.rand(allp, 30 + i)[:, 0] for i in range(3)])
    W = hx.KeySwitch(P.g, allp, kb, ka)
    dg = t[2].breakIntoDigits(digits, sp)
    o0 = hx.DoubleCRT(P.g, allp, 2, P.rand(allp, 40, 2))
    o1 = hx.DoubleCRT(P.g, allp, 2, P.rand(allp, 41, 2))
    i0, i1 = o0.download(), o1.download()
    hx.keySwitchDigits(dg, W, o0, o1)
    g0, g1 = o0.download(), o1.download()
    dgh = dg.download()
    for b in range(2):
        w0, w1 = P.o.key_switch_digits(allp, dgh[:, b].reshape(3, len(allp), P.N), kb, ka,
                                       i0[:, b], i1[:, b])
        assert np.array_equal(g0[:, b], w0) and np.array_equal(g1[:, b], w1)


@pytest.mark.parametrize("m,L,K,digits,batch", [
    (16384, 5, 2, [[0, 1], [2, 3], [4]], 2),
    # BASELINE config 3 shape: m=32768, L=16 x 60-bit, K=6 x 56-bit, D=3 (6/5/5)
    (32768, 16, 6, [list(range(0, 6)), list(range(6, 11)), list(range(11, 16))], 1),
    # a ring beyond one row kernel (m = 131072: transforms through pow2_big_rows)
    (131072, 4, 2, [[0, 1], [2, 3]], 2),
    # every digit count the key-switch kernel is instantiated for (keyswitch_kernel<ND>: 2, 3, 4 with the loads
    # hoisted, 0 = the run-time loop for anything else)
    (16384, 3, 2, [[0, 1, 2]], 2),
    (16384, 6, 2, [[0, 1], [2, 3], [4], [5]], 2),
    (16384, 6, 2, [[0, 1], [2], [3], [4], [5]], 3),
])
def test_multiply_relin_matches_oracle(hx, m, L, K, digits, batch):
    g = O.PrimeGen(60, m)
    primes = [g.next() for _ in range(L)]
    g2 = O.PrimeGen(56, m)
    primes += [g2.next() for _ in range(K)]
    P = Pair(hx, m, primes)
    own, sp = list(range(L)), list(range(L, L + K))
    allp = own + sp
    c0, c1, d0, d1 = (P.rand(own, s, batch) for s in (1, 2, 3, 4))
    D = len(digits)
    kb = np.stack([P.rand(allp, 20 + i)[:, 0] for i in range(D)])
    ka = np.stack([P.rand(allp, 30 + i)[:, 0] for i in range(D)])
    W = hx.KeySwitch(P.g, allp, kb, ka)
    G = [hx.DoubleCRT(P.g, own, batch, x) for x in (c0, c1, d0, d1)]
    o0, o1 = hx.multiplyBy(*G, W, digits)
    g0, g1 = o0.download(), o1.download()
    assert o0.getIndexSet() == allp
    for b in range(batch):
        w0, w1 = P.o.mul_relin(own, sp, digits, c0[:, b], c1[:, b], d0[:, b], d1[:, b], kb, ka)
        assert np.array_equal(g0[:, b], w0)
        assert np.array_equal(g1[:, b], w1)


def test_bgv_multiply_decrypts_on_gpu(hx):
    """decrypt(multiplyBy(enc a, enc b)) == a*b, with the multiply + mod-down on the GPU
    (tests/TestHEXL.cpp:158-187 style)."""
    from tests import bgv_ref as B
    m, p = 16384, 65537
    L, K = 5, 2
    digits = [[0, 1], [2, 3], [4]]
    P = Pair(hx, m, primes_for(m, L + K, 60))
    bp = B.Params(m, p, L, K, digits, ctx=P.o)
    s = B.keygen(bp)
    rng = np.random.default_rng(3)
    ma, mb = rng.integers(0, p, size=P.N), rng.integers(0, p, size=P.N)
    c0, c1 = B.encrypt(bp, s, ma, 1)
    d0, d1 = B.encrypt(bp, s, mb, 2)
    kb, ka = B.gen_ksk(bp, s)
    W = hx.KeySwitch(P.g, bp.all, kb, ka)
    G = [hx.DoubleCRT(P.g, bp.own, 1, x[:, None, :]) for x in (c0, c1, d0, d1)]
    o0, o1 = hx.multiplyBy(*G, W, digits)
    o0.scaleDownToSet(bp.own, p)
    o1.scaleDownToSet(bp.own, p)
    got, _ = B.decrypt(bp, s, o0.download()[:, 0], o1.download()[:, 0], bp.own)
    # negacyclic product of the plaintexts mod p
    want = np.zeros(P.N, dtype=object)
    fa = np.array([int(v) for v in ma], dtype=object)
    # use the oracle transform mod a big prime to multiply exactly, then reduce mod p
    q = P.primes[0]
    ea = P.o.fft([0], np.array([ma % q], dtype=np.uint64))
    eb = P.o.fft([0], np.array([mb % q], dtype=np.uint64))
    prod = P.o.ifft([0], np.array([O.row_op("mul", ea[0], eb[0], q)]))[0]
    want = [int(v) - q if int(v) > q // 2 else int(v) for v in prod]
    assert got == [w % p for w in want]


# ---------------------------------------------------------------- full-size properties
def test_full_size_properties_m32768(hx):
    """size-independent properties at BASELINE size with a batch: round trip, linearity,
    convolution theorem."""
    m, L, batch = 32768, 16, 8
    P = Pair(hx, m, primes_for(m, L))
    idx = list(range(L))
    a, b = P.rand(idx, 1, batch), P.rand(idx, 2, batch)
    da, db = hx.DoubleCRT(P.g, idx, batch, a), hx.DoubleCRT(P.g, idx, batch, b)
    s = da.copy()
    s += db
    s.FFT()
    da.FFT()
    db.FFT()
    t = da.copy()
    t += db
    assert np.array_equal(s.download(), t.download())          # NTT(a+b) = NTT(a)+NTT(b)
    assert np.array_equal(da.copy().iFFT().download(), a)       # round trip
    # X * a(X): multiply by the monomial in the eval domain == negacyclic shift
    mono = np.zeros((L, 1, P.N), dtype=np.uint64)
    mono[:, 0, 1] = 1
    dm = hx.DoubleCRT(P.g, idx, 1, mono).FFT()
    da *= dm
    sh = da.iFFT().download()
    want = np.roll(a, 1, axis=2)
    for r in idx:
        q = np.uint64(P.primes[r])
        w0 = want[r, :, 0]
        want[r, :, 0] = np.where(w0 == 0, np.uint64(0), q - w0)
    assert np.array_equal(sh, want)


# ---------------------------------------------------------------- small rings / HEXL shim
@pytest.mark.parametrize("logn", list(range(1, 13)))
def test_ntt_small_rings_match_oracle(hx, logn):
    N = 1 << logn
    m = 2 * N
    primes = primes_for(m, 2, 50)
    P = Pair(hx, m, primes)
    x = P.rand([0, 1], 3, batch=3)
    d = hx.DoubleCRT(P.g, [0, 1], 3, x)
    got = d.FFT().download()
    for b in range(3):
        assert np.array_equal(got[:, b], P.o.fft([0, 1], x[:, b]))
    assert np.array_equal(d.iFFT().download(), x)


@pytest.mark.parametrize("phim,m", [(8, 16), (64, 128), (256, 512)])
def test_CModulusFFT_like_TestHEXL(hx, phim, m):
    # tests/TestHEXL.cpp:189-218: q = PrimeGenerator(HELIB_SP_NBITS, m).next(); FFT then iFFT of 5X
    q = O.PrimeGen(60, m).next()
    P = Pair(hx, m, [q])
    assert P.N == phim
    x = np.zeros((1, 1, phim), dtype=np.uint64)
    x[0, 0, 1] = 5
    d = hx.DoubleCRT(P.g, [0], 1, x)
    y = d.FFT().download()
    assert np.array_equal(y[0, 0], P.o.fft([0], x[:, 0])[0])
    assert np.array_equal(d.iFFT().download(), x)


def test_intel_shim_like_TestHEXL_hexlInUse(hx):
    # tests/TestHEXL.cpp:139-156: intel::FFTFwd then FFTRev1 on N=64, q=769 is the identity -- and, beyond the
    # round trip, the ORDER the reference's call sites need: HEXL's ComputeForward delivers bit-reversed
    # evaluations (Cmodulus::FFT_aux applies BitReverseCopy afterwards, src/CModulus.cpp:385, :421-426) and
    # ComputeInverse consumes them (iFFT bit-reverses first, :510-514), under the NTT object's own root
    # MinimalPrimitiveRoot(2n, q).  Checker: the oracle's restatement of HEXL's published reference transform.
    import ctypes as C
    L = hx.lib()
    N, q = 64, 769
    a = (np.arange(N, dtype=np.int64) * 7 + 1) % q
    out = np.zeros(N, dtype=np.int64)
    assert L.hx_intel_FFTFwd(out.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), N, q) == 0
    assert not np.array_equal(out, a)
    assert np.array_equal(out.astype(np.uint64), O.hexl_forward(a.astype(np.uint64), q))
    psi = O.hexl_minimal_primitive_root(q, 2 * N)
    cm = O.Cmod(2 * N, q, psi)
    # FFT_aux's shape: shim forward, then BitReverseCopy = the natural row y[j] = f(psi^(2j+1))
    nat = O.bit_reverse_copy(out.astype(np.uint64))
    assert np.array_equal(nat, cm.fft(a.astype(np.uint64)))
    back = np.zeros(N, dtype=np.int64)
    assert L.hx_intel_FFTRev1(back.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), N, q) == 0
    assert np.array_equal(back, a)
    # iFFT's shape: BitReverseCopy of the natural row, then shim inverse
    rev = np.ascontiguousarray(O.bit_reverse_copy(nat)).astype(np.int64)
    back[:] = 0
    assert L.hx_intel_FFTRev1(back.ctypes.data_as(C.c_void_p), rev.ctypes.data_as(C.c_void_p), N, q) == 0
    assert np.array_equal(back, a)
    b = (np.arange(N, dtype=np.int64) * 13 + 5) % q
    r = np.zeros(N, dtype=np.int64)
    p = lambda v: v.ctypes.data_as(C.c_void_p)
    assert L.hx_intel_EltwiseAddMod(p(r), p(a), p(b), N, q) == 0
    assert np.array_equal(r, (a + b) % q)
    assert L.hx_intel_EltwiseSubMod(p(r), p(a), p(b), N, q) == 0
    assert np.array_equal(r, (a - b) % q)
    assert L.hx_intel_EltwiseMultMod(p(r), p(a), p(b), N, q) == 0
    assert np.array_equal(r, (a * b) % q)
    assert L.hx_intel_EltwiseMultModScalar(p(r), p(a), 5, N, q) == 0
    assert np.array_equal(r, (a * 5) % q)
    assert L.hx_intel_EltwiseAddModScalar(p(r), p(a), 768, N, q) == 0
    assert np.array_equal(r, (a + 768) % q)
    assert L.hx_intel_EltwiseSubModScalar(p(r), p(a), 768, N, q) == 0
    assert np.array_equal(r, (a - 768) % q)


@pytest.mark.parametrize("n,bits", [(64, 0), (16384, 60), (16384, 56), (32768, 60)])
def test_intel_shim_at_the_reference_call_sites(hx, n, bits):
    """The HEXL seam as src/CModulus.cpp:375-426 (FFT_aux) and :493-553 (iFFT) use it, on chain primes at the
    benchmark ring sizes: (1) intel::FFTFwd equals the oracle's restatement of hexl::NTT::ComputeForward, every
    word; (2) shim forward + BitReverseCopy equals hx_ntt_forward's natural row on a context that registers the
    prime with the shim's root; (3) BitReverseCopy + intel::FFTRev1 recovers x; (4) DoubleCRT::automorph
    (src/DoubleCRT.cpp:1160-1202, index j <-> 2j+1) applied to rows that came through the shim path gives the
    shim-path rows of a(X^k) -- what breaks if the seam hands back natural order."""
    import ctypes as C
    L = hx.lib()
    m = 2 * n
    q = 769 if bits == 0 else primes_for(m, 1, bits)[0]
    rng = np.random.default_rng(n + bits)
    a = rng.integers(0, q, n, dtype=np.uint64)
    a[:3] = [q - 1, 0, 1]
    p = lambda v: v.ctypes.data_as(C.c_void_p)

    def shim_fwd(x):
        o = np.zeros(n, dtype=np.int64)
        assert L.hx_intel_FFTFwd(p(o), p(np.ascontiguousarray(x).astype(np.int64)), n, q) == 0
        return o.astype(np.uint64)

    out = shim_fwd(a)
    assert np.array_equal(out, O.hexl_forward(a, q))                       # (1)
    psi = O.hexl_minimal_primitive_root(q, m)
    g = hx.Context(m, 0)
    g.add_prime(q, psi)
    d = hx.DoubleCRT(g, [0], 1, a.reshape(1, 1, n))
    nat = d.FFT().download().reshape(n)
    assert np.array_equal(O.bit_reverse_copy(out), nat)                    # (2)
    back = np.zeros(n, dtype=np.int64)
    rev = np.ascontiguousarray(O.bit_reverse_copy(nat)).astype(np.int64)
    assert L.hx_intel_FFTRev1(p(back), p(rev), n, q) == 0
    assert np.array_equal(back.astype(np.uint64), a)                       # (3)
    for k in (3, 5, m - 1, 2 * 7 + 1):
        ak = np.zeros(n, dtype=np.uint64)                                  # a(X^k) mod X^n + 1
        e = (np.arange(n, dtype=np.int64) * k) % m
        ak[e % n] = np.where(e < n, a, (q - a) % q)
        want = O.bit_reverse_copy(shim_fwd(ak))
        dk = hx.DoubleCRT(g, [0], 1, O.bit_reverse_copy(out).reshape(1, 1, n).copy())   # rows via the shim path
        dk.automorph(k)
        assert np.array_equal(dk.download().reshape(n), want), k            # (4)


# ---------------------------------------------------------------- general m (Bluestein)
@pytest.mark.parametrize("m", [105, 1705, 4369, 12, 20, 1000, 28])
def test_bluestein_small_m_matches_oracle(hx, m):
    primes = primes_for(m, 3, 60)
    P = Pair(hx, m, primes)
    idx = [0, 1, 2]
    x = P.rand(idx, 5, batch=2)
    d = hx.DoubleCRT(P.g, idx, 2, x)
    got = d.FFT().download()
    for b in range(2):
        assert np.array_equal(got[:, b], P.o.fft(idx, x[:, b]))
    y = P.rand(idx, 6, batch=2)
    d.upload(y)
    back = d.iFFT().download()
    for b in range(2):
        assert np.array_equal(back[:, b], P.o.ifft(idx, y[:, b]))
    assert np.array_equal(d.FFT().download(), y)


def test_add_prime_rejects_a_root_of_smaller_order_for_general_m(hx):
    """FindPrimRootT's independent check (src/NumbTh.cpp): a caller-supplied root must have order
    exactly e -- root^(e/p) != 1 for EVERY prime factor p of e, not only p = 2 (m = 105: e = 105,
    a root of order 35 or 21 satisfies root^105 = 1 and root^52 != 1)."""
    m = 105
    q = O.PrimeGen(50, m).next()
    g = O.lib().ho_find_prim_root(q, m)
    ctx = hx.Context(m)
    for bad in (pow(g, 3, q), pow(g, 5, q), pow(g, 7, q), pow(g, 15, q)):
        with pytest.raises(hx.HxError) as ei:
            ctx.add_prime(q, bad)
        assert "primitive" in str(ei.value)
    assert ctx.add_prime(q, g) == 0


def test_bluestein_m21845_config5(hx):
    """BASELINE configs[4]: bootstrapping-style m = 21845 = 5*17*257, phi = 16384, conv 2^16
    (tests/GTestBootstrapping.cpp:113 parameters); forward + inverse DoubleCRT NTT bit-exact vs
    the restatement of src/bluestein.cpp."""
    m = 21845
    primes = primes_for(m, 4, 60)
    P = Pair(hx, m, primes)
    assert P.N == 16384
    idx = [0, 1, 2, 3]
    x = P.rand(idx, 9, batch=2)
    d = hx.DoubleCRT(P.g, idx, 2, x)
    got = d.FFT().download()
    for b in range(2):
        assert np.array_equal(got[:, b], P.o.fft(idx, x[:, b]))
    assert np.array_equal(d.iFFT().download(), x)          # round trip
    y = P.rand(idx, 10, batch=1)
    d1 = hx.DoubleCRT(P.g, idx, 1, y)
    assert np.array_equal(d1.iFFT().download()[:, 0], P.o.ifft(idx, y[:, 0]))
    # ring identity at full size: FFT(a)*FFT(b) = FFT(a*b mod Phi_m): X * a(X) with deg a < phi-1
    a = x.copy()
    a[:, :, -1] = 0
    da = hx.DoubleCRT(P.g, idx, 2, a).FFT()
    mono = np.zeros((4, 1, P.N), dtype=np.uint64)
    mono[:, 0, 1] = 1
    da *= hx.DoubleCRT(P.g, idx, 1, mono).FFT()
    assert np.array_equal(da.iFFT().download(), np.roll(a, 1, axis=2))


@pytest.mark.parametrize("path", ["pfa", "pfa_conv_rem", "fused", "old"])
def test_bluestein_m21845_config5_at_L16(hx, path, monkeypatch):
    """BASELINE configs[4] at its own shape: m = 21845, L = 16 primes of PrimeGenerator(60, 21845), a batch of 2
    DoubleCRT objects -- forward and inverse transforms of all 32 rows against the restatement of
    src/bluestein.cpp / src/CModulus.cpp:431-443, 555-577, every word.
      pfa   (round 6, the default): Good-Thomas x Rader, 21845 = 5 * 17 * 257 (pfa_kernels.hip): ONE launch per
            direction, the 4- / 16- / 256-point cyclic convolutions of Rader's form of DFT_5 (x) DFT_17 (x) DFT_257;
            the inverse has rem Phi_m behind it in the same launch as binomial passes (Phi_m is a quotient of products
            of x^d - 1: no multiplication at all);
      pfa_conv_rem (HX_PFA_NO_REM): the same transform, rem Phi_m as two fused convolution launches;
      fused (HX_NO_PFA): Bluestein on the convolution row kernel (conv_kernels.hip);
      old   (HX_BLUE_OLD): the round-2 chain of separate passes, the fallback for exotic primes and sizes.
    All three must give the oracle's words; the kernel summary pins which one ran (the switches are read per
    context: switches.h)."""
    if path == "fused":
        monkeypatch.setenv("HX_NO_PFA", "1")
    elif path == "pfa_conv_rem":
        monkeypatch.setenv("HX_PFA_NO_REM", "1")
    elif path == "old":
        monkeypatch.setenv("HX_BLUE_OLD", "1")
    m, L = 21845, 16
    P = Pair(hx, m, primes_for(m, L, 60))
    idx = list(range(L))
    x = P.rand(idx, 9, batch=2)
    d = hx.DoubleCRT(P.g, idx, 2, x)
    hx.profileBegin()
    got = d.FFT().download()
    names = " ".join(k["kernel"] for k in hx.profileEnd()["kernels"])
    assert ("pfa_row_kernel<0," in names) == path.startswith("pfa"), names
    assert ("ntt_conv_kernel" in names) == (path == "fused"), names
    for b in range(2):
        assert np.array_equal(got[:, b], P.o.fft(idx, x[:, b]))
    hx.profileBegin()
    assert np.array_equal(d.iFFT().download(), x)
    names = " ".join(k["kernel"] for k in hx.profileEnd()["kernels"])
    assert ("pfa_row_kernel<2," in names) == (path == "pfa"), names
    assert ("pfa_row_kernel<1," in names) == (path == "pfa_conv_rem"), names
    assert ("ntt_conv_kernel" in names) == (path in ("pfa_conv_rem", "fused")), names
    y = P.rand(idx, 10, batch=2)
    # extreme words: all q - 1, all zero, a single one
    y[:, 0, :3] = np.array([[q - 1, 0, 1] for q in P.primes[:L]], dtype=np.uint64)
    d1 = hx.DoubleCRT(P.g, idx, 2, y)
    back = d1.iFFT().download()
    for b in range(2):
        assert np.array_equal(back[:, b], P.o.ifft(idx, y[:, b]))
    # a lazily copied (shared) poly is transformed out of place into its own slab
    c = d1.copy()
    assert np.array_equal(c.FFT().download(), y) and np.array_equal(d1.download(), back)
    # the rows of all-(q-1) and of zeros (every lazy bound of the kernels at its extreme)
    e = np.stack([np.stack([np.full(P.N, q - 1, dtype=np.uint64), np.zeros(P.N, dtype=np.uint64)]) for q in P.primes[:L]])
    de = hx.DoubleCRT(P.g, idx, 2, e)
    ge = de.FFT().download()
    for b in range(2):
        assert np.array_equal(ge[:, b], P.o.fft(idx, e[:, b]))
    assert np.array_equal(de.iFFT().download(), e)


@pytest.mark.parametrize("no_proth", [False, True])
def test_good_thomas_rader_on_the_small_primes_of_a_chain(hx, no_proth, monkeypatch):
    """m = 21845 on primes that are NOT of the Proth form -- the 36- / 40- / 48-bit small primes of a chain are
    t m 2^k + 1 with k < 32 -- next to 60-bit ones in ONE row list: every row runs the Good-Thomas x Rader kernel
    (the generic Montgomery product of pfa_core.h QCG on the small primes; no Bluestein launch at all), forward,
    inverse and the extreme rows, against the oracle.  Under HX_NO_PROTH every row takes the generic product."""
    if no_proth:
        monkeypatch.setenv("HX_NO_PROTH", "1")
    m = 21845
    primes = primes_for(m, 2, 60) + primes_for(m, 2, 40) + primes_for(m, 1, 48) + primes_for(m, 1, 36)
    assert sum(1 for q in primes if (q & 0xffffffff) == 1) == 2
    P = Pair(hx, m, primes)
    idx = list(range(len(primes)))
    x = P.rand(idx, 19, batch=3)
    x[:, 2, :] = np.array([q - 1 for q in primes], dtype=np.uint64)[:, None]
    d = hx.DoubleCRT(P.g, idx, 3, x)
    hx.profileBegin()
    got = d.FFT().download()
    back = d.iFFT().download()
    names = " ".join(k["kernel"] for k in hx.profileEnd()["kernels"])
    assert "pfa_row_kernel<0," in names and "pfa_row_kernel<2," in names and "ntt_conv_kernel" not in names, names
    for b in range(3):
        assert np.array_equal(got[:, b], P.o.fft(idx, x[:, b]))
    assert np.array_equal(back, x)
    y = P.rand(idx, 20, batch=1)
    assert np.array_equal(hx.DoubleCRT(P.g, idx, 1, y).iFFT().download()[:, 0], P.o.ifft(idx, y[:, 0]))


def test_general_m_multiply_relin_and_automorph(hx):
    m = 1705                                    # 5*11*31, phi = 1200
    L, K = 4, 2
    digits = [[0, 1], [2, 3]]
    P = Pair(hx, m, primes_for(m, L + K, 60))
    own, sp = list(range(L)), list(range(L, L + K))
    allp = own + sp
    c0, c1, d0, d1 = (P.rand(own, s, 2) for s in (1, 2, 3, 4))
    kb = np.stack([P.rand(allp, 20 + i)[:, 0] for i in range(2)])
    ka = np.stack([P.rand(allp, 30 + i)[:, 0] for i in range(2)])
    W = hx.KeySwitch(P.g, allp, kb, ka)
    G = [hx.DoubleCRT(P.g, own, 2, x) for x in (c0, c1, d0, d1)]
    o0, o1 = hx.multiplyBy(*G, W, digits)
    for b in range(2):
        w0, w1 = P.o.mul_relin(own, sp, digits, c0[:, b], c1[:, b], d0[:, b], d1[:, b], kb, ka)
        assert np.array_equal(o0.download()[:, b], w0) and np.array_equal(o1.download()[:, b], w1)
    o0.scaleDownToSet(own, 7)
    assert np.array_equal(o0.download()[:, 0],
                          P.o.scale_down(allp, P.o.mul_relin(own, sp, digits, c0[:, 0], c1[:, 0], d0[:, 0],
                                                             d1[:, 0], kb, ka)[0], sp, 7))
    zms = O.zmstar(m)
    a = P.rand(own, 8, 1)
    got = hx.DoubleCRT(P.g, own, 1, a).automorph(3).download()
    for r in own:
        assert np.array_equal(got[r, 0], O.automorph(a[r, 0], m, zms, 3))


# ---------------------------------------------------------------- full Ctxt::multiplyBy sequence
@pytest.mark.parametrize("m,p,bits", [(16384, 65537, 250), (1705, 7, 200), (21845, 2, 950)])
def test_ctxt_multiplyBy_full_sequence_gpu_vs_oracle(hx, m, p, bits, monkeypatch):
    """The reference's own order of operations for fresh ciphertexts (src/Ctxt.cpp:1681-1774):
    bringToSet (mod-up by a small prime, mod-down by a ctxt prime) -> tensorProduct ->
    dropSmallAndSpecialPrimes -> reLinearize, driven by the same host logic (helib_amd.ctxt) once
    over the GPU and once over the oracle; every part must agree bit-for-bit and decrypt to a*b."""
    from helib_amd import ctxt as hc
    from tests import test_ctxt_host as T
    from oracle.backend import OKeySwitch, OPoly, OracleOps
    # measured noise (device norms) where the device has them; the oracle side is made
    # to take the same branch so that the two runs are comparable
    monkeypatch.setattr(hc.Ctxt, "measure", hx.supportsNorms(m))
    ctx = hc.ChainContext(m, p, 1, bits=bits, c=3)
    P = Pair(hx, m, ctx.primes)
    s, allp, kb, ka, rows = T.make_keys(ctx, P.o)
    rng = np.random.default_rng(4)
    ma, mb = rng.integers(0, p, size=P.N), rng.integers(0, p, size=P.N)
    ea, eb = T.encrypt(ctx, P.o, s, ma, 1, rows), T.encrypt(ctx, P.o, s, mb, 2, rows)
    # oracle-driven
    oops = OracleOps(P.o)
    oW = OKeySwitch(allp, kb, ka)
    oa = hc.Ctxt.fresh(ctx, oops, *(OPoly(P.o, ctx.ctxtPrimes, x) for x in ea), ksw=oW)
    ob = hc.Ctxt.fresh(ctx, oops, *(OPoly(P.o, ctx.ctxtPrimes, x) for x in eb), ksw=oW)
    oa.multiplyBy(ob)
    # GPU-driven
    gW = hx.KeySwitch(P.g, allp, kb, ka)
    ga = hc.Ctxt.fresh(ctx, hx, *(hx.DoubleCRT(P.g, ctx.ctxtPrimes, 1, x[:, None, :]) for x in ea), ksw=gW)
    gb = hc.Ctxt.fresh(ctx, hx, *(hx.DoubleCRT(P.g, ctx.ctxtPrimes, 1, x[:, None, :]) for x in eb), ksw=gW)
    ga.multiplyBy(gb)
    assert ga.primeSet == oa.primeSet and ga.intFactor == oa.intFactor
    assert abs(ga.lnNoise - oa.lnNoise) < 1e-8      # measured norms agree to ~1e-12 relative
    assert ga._meas == hx.supportsNorms(m) == oa._meas
    for h in ("1", "s"):
        gi, oi = ga.parts[h].getIndexSet(), oa.parts[h].getIndexSet()
        assert sorted(gi) == sorted(oi)
        gd, od = ga.parts[h].download()[:, 0], oa.parts[h].download()[:, 0]
        for r, i in enumerate(gi):
            assert np.array_equal(gd[r], od[oi.index(i)]), (h, i)
    from tests import bgv_ref as B
    want = [int(v) for v in B.polymul_mod_phi(ma, mb, m, p)] if m < 4096 else None
    got = T.decrypt(ctx, P.o, s, ga, rows)
    if want is not None:
        assert got == want
    else:
        assert got == T.decrypt(ctx, P.o, s, oa, rows)


class FusedCallSpy:
    """Wraps the backend module's fused entry points for one test: records every hx.tensorBringToSet / hx.mulRelin
    call of the host logic with its shape (operand primes, added, dropped, kept, batch) and fails the test if the
    unfused hx.tensorProduct runs -- i.e. it pins WHICH device path a ciphertext-level parity test went through."""

    def __init__(self, hx, monkeypatch, allow_tensor=False):
        self.calls = []
        tb, mr, tp = hx.tensorBringToSet, hx.mulRelin, hx.tensorProduct

        def tensorBringToSet(c0, c1, d0, d1, add, keep, ptxt, **kw):
            own = c0.getIndexSet()
            a = [i for i in add if i not in own]
            self.calls.append(("tensorBringToSet", dict(own=own, add=a, drop=[i for i in own + a if i not in set(keep)],
                                                        nkeep=len(set(keep)), batch=c0.batch)))
            return tb(c0, c1, d0, d1, add, keep, ptxt, **kw)

        def mulRelin(c0, c1, d0, d1, W, digits, **kw):
            self.calls.append(("mulRelin", dict(own=c0.getIndexSet(), batch=c0.batch)))
            return mr(c0, c1, d0, d1, W, digits, **kw)

        def tensorProduct(*a):
            assert allow_tensor, "the unfused tensor product ran: the host did not take the fused path"
            return tp(*a)

        monkeypatch.setattr(hx, "tensorBringToSet", tensorBringToSet)
        monkeypatch.setattr(hx, "mulRelin", mulRelin)
        monkeypatch.setattr(hx, "tensorProduct", tensorProduct)

    def names(self):
        return [c[0] for c in self.calls]


@pytest.mark.parametrize("bits,B", [(950, 4)])
def test_fresh_multiplyBy_at_the_benchmarked_shape_batched(hx, bits, B, monkeypatch):
    """bench.py's own step through the kernels bench.py times, bit-exact: BASELINE configs[2] (m=32768, p=65537,
    bits=950 -> L=16, K=6, D=3), the FRESH multiplyBy sequence of src/Ctxt.cpp:1681-1774 over a batch of B DISTINCT
    ciphertext pairs in one set of launches, driven by helib_amd.ctxt -- which, like the C++ host of the timed loop,
    leaves the tensor product to its consumer: hx_bring_to_set_multi of the four operand parts (add 1 small prime,
    drop 1 ctxt prime), then hx_tensor_bring_to_set_norms (ntt_moddown_prep_tensor_kernel +
    ntt_moddown_apply_tensor_kernel<14,false> at nkeep = 16, the md_tile g=2 launch of the bench), then
    hx_relinearize.  Then LEVEL 2, the product times itself (operands that carry the special primes):
    the several-primes mod-switch of the operands and hx_tensor_bring_to_set_norms in its several-primes form
    (ntt_moddown_prep_multi_tensor_kernel + ntt_moddown_apply_tensor_kernel<14,true>).  The call spy pins that
    these entry points -- and never hx_tensor -- ran, with the shapes named above.  The oracle runs the same host
    logic once per batch element over the reference's unfused sequence (oracle/backend.py); EVERY part, row and
    batch element is compared at both levels, and every element decrypts to its plaintext product."""
    from helib_amd import ctxt as hc
    from tests import test_ctxt_host as T
    from oracle.backend import OKeySwitch, OPoly, OracleOps
    m, p = 32768, 65537
    monkeypatch.setattr(hc.Ctxt, "measure", True)
    spy = FusedCallSpy(hx, monkeypatch)
    ctx = hc.ChainContext(m, p, 1, bits=bits, c=3)
    assert len(ctx.ctxtPrimes) == 16 and len(ctx.specialPrimes) == 6 and len(ctx.digits) == 3
    P = Pair(hx, m, ctx.primes)
    s, allp, kb, ka, rows = T.make_keys(ctx, P.o)
    rng = np.random.default_rng(40)
    msgs = rng.integers(0, p, size=(2, B, P.N))
    enc = [[T.encrypt(ctx, P.o, s, msgs[j, b], 10 * j + b + 1, rows) for b in range(B)] for j in range(2)]
    oops, oW = OracleOps(P.o), OKeySwitch(allp, kb, ka)
    outs = []
    for b in range(B):
        oa = hc.Ctxt.fresh(ctx, oops, *(OPoly(P.o, ctx.ctxtPrimes, x) for x in enc[0][b]), ksw=oW)
        ob = hc.Ctxt.fresh(ctx, oops, *(OPoly(P.o, ctx.ctxtPrimes, x) for x in enc[1][b]), ksw=oW)
        oa.multiplyBy(ob)
        outs.append(oa)
    gW = hx.KeySwitch(P.g, allp, kb, ka)

    def batched(j):
        return [hx.DoubleCRT(P.g, ctx.ctxtPrimes, B, np.stack([enc[j][b][k] for b in range(B)], axis=1))
                for k in range(2)]

    def same(gc, ocs):
        for h in ("1", "s"):
            gi = gc.parts[h].getIndexSet()
            gd = gc.parts[h].download()
            assert gd.shape[1] == B
            for b in range(B):
                oi, od = ocs[b].parts[h].getIndexSet(), ocs[b].parts[h].download()[:, 0]
                assert sorted(gi) == sorted(oi)
                for r, i in enumerate(gi):
                    assert np.array_equal(gd[r, b], od[oi.index(i)]), (h, i, b)

    ga = hc.Ctxt.fresh(ctx, hx, *batched(0), ksw=gW)
    gb = hc.Ctxt.fresh(ctx, hx, *batched(1), ksw=gW)
    ga.multiplyBy(gb)
    # the device path taken: the product formed inside the single-prime mod-switch at nkeep = 16
    assert spy.names() == ["tensorBringToSet"]
    shape = spy.calls[0][1]
    assert (len(shape["own"]), len(shape["add"]), len(shape["drop"]), shape["nkeep"], shape["batch"]) == (16, 1, 1, 16, B)
    # the batch shares one prime-set decision: the estimate is the maximum over its elements
    assert all(ga.primeSet == o.primeSet and ga.intFactor == o.intFactor for o in outs)
    on = [o.lnNoise for o in outs]
    assert max(on) - 1e-6 <= ga.lnNoise <= max(on) + 0.05
    same(ga, outs)
    for b in range(B):
        full = np.convolve(msgs[0, b].astype(np.int64), msgs[1, b].astype(np.int64))
        want = (full[:P.N] - np.append(full[P.N:], 0)) % p
        assert T.decrypt(ctx, P.o, s, outs[b], rows) == [int(v) for v in want]
    # level 2: (a*b)^2, every element with the batch's estimate (its prime-set decisions are the batch's)
    worst = ga.lnNoise
    for o in outs:
        o.lnNoise = worst
    ga.multiplyBy(ga.clone())
    for o in outs:
        o.multiplyBy(o.clone())
    assert spy.names() == ["tensorBringToSet", "tensorBringToSet"]
    shape = spy.calls[1][1]
    assert len(shape["drop"]) >= 2 and shape["nkeep"] == 16 and shape["batch"] == B, shape   # the several-primes form
    assert all(ga.primeSet == o.primeSet and ga.intFactor == o.intFactor for o in outs)
    same(ga, outs)


BENCH_SHAPES = {
    # the tensorBringToSet calls of the benchmark loops, as helib_amd.ctxt / the C++ host issue them (pinned by the
    # call spies of test_fresh_multiplyBy_at_the_benchmarked_shape_batched and the CKKS chain tests):
    # name: (scheme, m, p or precision, bits, own rows, add, drop, batch); c = ctxt primes, s = small primes by position
    "bgv950_level1": ("bgv", 32768, 65537, 950, "c[:15] + [s[4]]", "[c[15]]", "[s[4]]", 4),
    "bgv950_level2": ("bgv", 32768, 65537, 950, "c[:14] + s[0:2]", "c[14:16]", "s[0:2]", 4),
    "ckks1400_level2": ("ckks", 65536, 20, 1400, "c[:21] + [s[0], s[2], s[3]]", "c[21:23]", "[s[0], s[2], s[3]]", 2),
    "ckks1400_precision1_level2": ("ckks", 65536, 1, 1400, "c[:22] + s[2:4]", "c[22:24]", "s[2:4]", 2),
    "ckks440_level2": ("ckks", 65536, 1, 440, "c[:6] + s[2:4]", "c[6:8]", "s[2:4]", 2),
}


@pytest.mark.parametrize("name", list(BENCH_SHAPES))
def test_tensor_folded_into_the_mod_switch_at_the_benchmarked_shapes(hx, name):
    """hx_tensor_bring_to_set_norms at the launch shapes of the timed loops (bench.py; benchmarks/bgv_basic.cpp:144-165,
    benchmarks/ckks_basic.cpp:161-180 on the chains of bits = 950 / 1400 / 440): the real chain's primes (60-bit
    ctxt primes next to 40..57-bit small primes), operand rows in the order the host leaves them, nkeep = 16 (BGV:
    the md_tile g=2 tile of ntt_moddown_apply_tensor_kernel<14,*>) / 23 and 8 (CKKS, <15,true>), every word of every
    batch element of the three product parts against (a) hx_tensor + hx_bring_to_set_multi on the device and (b) the
    oracle's tensor product, addPrimesAndScale and scaleDownToSet; the measured norms against each other."""
    from helib_amd import ctxt as hc
    scheme, m, pr, bits, own_e, add_e, drop_e, B = BENCH_SHAPES[name]
    ctx = hc.ChainContext(m, 65537, 1, bits=bits, c=3) if scheme == "bgv" else \
        hc.ChainContext(m, -1, pr, bits=bits, c=3, ckks=True)
    env = {"c": list(ctx.ctxtPrimes), "s": list(ctx.smallPrimes)}
    own, add, drop = (list(eval(e, env)) for e in (own_e, add_e, drop_e))
    ptxt = 65537 if scheme == "bgv" else 1
    keep = [i for i in own + add if i not in drop]
    P = Pair(hx, m, ctx.primes)
    ops = [P.rand(own, 900 + i, batch=B) for i in range(4)]
    c0, c1, d0, d1 = (hx.DoubleCRT(P.g, own, B, x) for x in ops)
    fused, nf = hx.tensorBringToSet(c0, c1, d0, d1, add, keep, ptxt, norms=True)
    t = list(hx.tensorProduct(c0, c1, d0, d1))
    ns = hx.bringToSetMulti(t, add, keep, ptxt, norms=True)
    assert np.allclose(nf, ns, rtol=1e-12, atol=0)
    for part in range(3):
        gi, ti = fused[part].getIndexSet(), t[part].getIndexSet()
        assert sorted(gi) == sorted(keep) == sorted(ti)
        gd, td = fused[part].download(), t[part].download()
        for r, i in enumerate(gi):
            assert np.array_equal(gd[r], td[ti.index(i)]), (part, i)
    fd = [f.download() for f in fused]
    for b in range(B):
        w = P.o.tensor(own, ops[0][:, b], ops[1][:, b], ops[2][:, b], ops[3][:, b])
        for part in range(3):
            up = np.vstack([P.o.scale_by_primes(own, w[part], add)] + [np.zeros((1, P.N), dtype=np.uint64)] * len(add))
            want = P.o.scale_down(own + add, up, drop, ptxt)
            gi = fused[part].getIndexSet()
            for r, i in enumerate(gi):
                assert np.array_equal(fd[part][r, b], want[keep.index(i)]), (part, i, b)


@pytest.mark.parametrize("ptxt", [65537, 2, 1, 4, 4294967311, 1 << 40])   # (the last two: beyond a 32-bit word)
@pytest.mark.parametrize("m", [16384, 32768])
def test_scale_down_single_prime_fused_path(hx, m, ptxt):
    """One dropped prime takes the fused path (delta prepared inside the inverse transform of
    the dropped row, subtract/divide inside the forward transform's store); the surviving rows
    may be re-ordered (last row moves into the freed slot), so rows are matched by prime."""
    P, own, sp = setup_rns(hx, m=m, L=5, K=2)
    allp = own + sp
    a = P.rand(allp, 12, batch=2)
    for drop in (allp[-1], allp[2], allp[0]):
        d = hx.DoubleCRT(P.g, allp, 2, a)
        keep = [i for i in allp if i != drop]
        d.scaleDownToSet(keep, ptxt)
        got_idx = d.getIndexSet()
        assert sorted(got_idx) == sorted(keep)
        got = d.download()
        for b in range(2):
            want = P.o.scale_down(allp, a[:, b], [drop], ptxt)
            for r, i in enumerate(got_idx):
                assert np.array_equal(got[r, b], want[keep.index(i)]), (drop, i)


@pytest.mark.parametrize("ndrop", [1, 2])
def test_bring_to_set_multi_matches_separate_steps(hx, ndrop):
    """hx_bring_to_set_multi == addPrimesAndScale then scaleDownToSet (oracle), for the fused
    single-drop path and for the generic fallback, on several parts at once."""
    P, own, sp = setup_rns(hx, m=16384, L=5, K=2)
    add = [sp[0]]
    drop = own[-ndrop:]
    parts = [P.rand(own, 50 + i, batch=2) for i in range(3)]
    polys = [hx.DoubleCRT(P.g, own, 2, x) for x in parts]
    keep = [i for i in own + add if i not in drop]
    hx.bringToSetMulti(polys, add, keep, 65537)
    for x, d in zip(parts, polys):
        idx = d.getIndexSet()
        assert sorted(idx) == sorted(keep)
        got = d.download()
        for b in range(2):
            up = np.vstack([P.o.scale_by_primes(own, x[:, b], add), np.zeros((1, P.N), dtype=np.uint64)])
            want = P.o.scale_down(own + add, up, drop, 65537)
            for r, i in enumerate(idx):
                assert np.array_equal(got[r, b], want[keep.index(i)])


@pytest.mark.parametrize("m,ptxt", [(16384, 65537), (32768, 65537), (16384, 2), (65536, 1)])
@pytest.mark.parametrize("shape", ["add1_drop1", "drop1", "drop_last", "add1_drop2", "drop2", "drop3"])
def test_tensor_folded_into_the_mod_switch(hx, m, ptxt, shape):
    """hx_tensor_bring_to_set (Ctxt::tensorProduct followed by Ctxt::bringToSet of the product, what Ctxt::multiplyBy
    does between multLowLvl and the key switch): the product parts formed inside the single-prime mod-down kernels
    from the operands' rows must equal hx_tensor + hx_bring_to_set_multi word for word -- and those equal the
    oracle's tensor product, addPrimesAndScale and scaleDownToSet.  Shapes: mod-up by one prime + one dropped prime
    (the fresh multiply's), a pure single-prime mod-down (dropped row in the middle and last), and two / three dropped
    primes with and without a mod-up (the several-primes kernels with the product parts formed on load and in the
    store: every multiply after the first).  A batch of 3, operands that are lazy copies of each other, measured
    norms."""
    P, own, sp = setup_rns(hx, m=m, L=5, K=2)
    B = 3
    add = [sp[0]] if shape.startswith("add1") else []
    drop = {"add1_drop1": [own[2]], "drop1": [own[1]], "drop_last": [own[-1]], "add1_drop2": [own[1], own[3]],
            "drop2": [own[0], own[4]], "drop3": [own[1], own[2], own[4]]}[shape]
    keep = [i for i in own + add if i not in drop]
    ops = [P.rand(own, 700 + i, batch=B) for i in range(3)]
    c0 = hx.DoubleCRT(P.g, own, B, ops[0])
    c1 = hx.DoubleCRT(P.g, own, B, ops[1])
    d0 = c0.copy()                                           # lazily shared with c0
    d1 = hx.DoubleCRT(P.g, own, B, ops[2])
    fused, nf = hx.tensorBringToSet(c0, c1, d0, d1, add, keep, ptxt, norms=True)
    for d, x in zip((c0, c1, d0, d1), (ops[0], ops[1], ops[0], ops[2])):
        assert np.array_equal(d.download(), x)               # operands untouched
    t = list(hx.tensorProduct(c0, c1, d0, d1))
    ns = hx.bringToSetMulti(t, add, keep, ptxt, norms=True)
    for part in range(3):
        assert sorted(fused[part].getIndexSet()) == sorted(keep) == sorted(t[part].getIndexSet())
        gi, ti = fused[part].getIndexSet(), t[part].getIndexSet()
        gd, td = fused[part].download(), t[part].download()
        for r, i in enumerate(gi):
            assert np.array_equal(gd[r], td[ti.index(i)]), (part, i)
    assert np.allclose(nf, ns, rtol=1e-12, atol=0)
    # and against the oracle, element 0 of the batch
    w = P.o.tensor(own, ops[0][:, 0], ops[1][:, 0], ops[0][:, 0], ops[2][:, 0])
    for part in range(3):
        up = np.vstack([P.o.scale_by_primes(own, w[part], add)] + [np.zeros((1, P.N), dtype=np.uint64)] * len(add)) if add else w[part]
        want = P.o.scale_down(own + add, up, drop, ptxt)
        gi, gd = fused[part].getIndexSet(), fused[part].download()
        for r, i in enumerate(gi):
            assert np.array_equal(gd[r, 0], want[keep.index(i)]), (part, i)


@pytest.mark.parametrize("m", [16384, 65536])
@pytest.mark.parametrize("ptxt", [65537, 1])
def test_several_primes_mod_switch_batched_over_parts_mixed_prime_sizes(hx, m, ptxt):
    """The batched several-primes mod-switch (engine.hip scale_down_multi_fused): three parts, two of
    them lazy copies of one object, a batch of 2, dropped primes of DIFFERENT sizes (56-bit special
    primes next to a 60-bit ctxt prime, as every multiply after the first drops them: the dropped
    set is Garner-ed in ascending order), with and without a mod-up by a small prime folded in.
    Every word against the oracle's addPrimesAndScale + scaleDownToSet."""
    g60, g56, g45 = O.PrimeGen(60, m), O.PrimeGen(56, m), O.PrimeGen(45, m)
    primes = [g60.next() for _ in range(5)] + [g56.next() for _ in range(3)] + [g45.next()]
    P = Pair(hx, m, primes)
    ct, sp, small = [0, 1, 2, 3, 4], [5, 6, 7], 8
    cur = ct + sp
    B = 2
    base = [P.rand(cur, 300 + i, batch=B) for i in range(2)]
    for add in ([], [small]):
        drop = sp + [ct[-1]]
        keep = [i for i in cur if i not in drop] + add
        x = hx.DoubleCRT(P.g, cur, B, base[0])
        parts = [x.copy(), x.copy(), hx.DoubleCRT(P.g, cur, B, base[1])]
        if add:
            hx.bringToSetMulti(parts, add, keep, ptxt)
        else:
            hx.scaleDownToSetMulti(parts, keep, ptxt)
        assert np.array_equal(x.download(), base[0])          # the shared source is untouched
        for part, src in zip(parts, (base[0], base[0], base[1])):
            idx = part.getIndexSet()
            assert sorted(idx) == sorted(keep)
            got = part.download()
            for b in range(B):
                rows, have = src[:, b], list(cur)
                if add:
                    rows = np.vstack([P.o.scale_by_primes(cur, rows, add), np.zeros((len(add), P.N), dtype=np.uint64)])
                    have = cur + add
                want = P.o.scale_down(have, rows, drop, ptxt)
                wkeep = [i for i in have if i not in drop]
                for r, i in enumerate(idx):
                    assert np.array_equal(got[r, b], want[wkeep.index(i)]), (add, i, b)


# ---------------------------------------------------------------- C++ host facade
def test_cpp_facade_matches_oracle(hx, tmp_path):
    """include/helib_amd.hpp (the reference-named C++ classes over the C ABI) driven by a C++
    program and compared with the C oracle: FFT/iFFT, ring ops, automorph + its RuntimeError,
    IndexSet mismatch, addPrimes, scaleDownToSet, breakIntoDigits, multiplyBy; m = 4096 and the
    Bluestein case m = 1705."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle")])
    exe = str(tmp_path / "facade_test")
    libdir = os.path.join(root, "helib_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"),
                           "-I" + os.path.join(root, "oracle"),
                           os.path.join(root, "tests", "cpp", "facade_test.cpp"),
                           "-L" + libdir, "-lhelib_amd", os.path.join(root, "oracle", "liboracle.so"),
                           "-Wl,-rpath," + libdir, "-Wl,-rpath," + os.path.join(root, "oracle"),
                           "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout + r.stderr
    assert "FAIL" not in r.stdout


# ---------------------------------------------------------------- N1: measured noise norms
NORM_RTOL = 1e-9   # double-precision FFT vs the oracle's long-double evaluation


@pytest.mark.parametrize("m", [4, 16, 1024, 16384, 32768, 65536, 131072])
def test_embedding_norm_device_matches_oracle(hx, m):
    """embeddingLargestCoeff (src/norms.cpp:480-493) on the device: N <= 8192 in one workgroup,
    larger N split by DIF levels over 2..8 workgroups per polynomial."""
    g = hx.Context(m)
    N = g.phim
    rng = np.random.default_rng(m)
    f = np.stack([rng.normal(size=N), rng.integers(-32768, 32769, size=N).astype(float),
                  np.ones(N), np.eye(1, N, N - 1)[0], np.zeros(N),
                  np.cos(2 * np.pi * 5 * np.arange(N) / m)])
    got = hx.embeddingLargestCoeff(g, f)
    for r in range(f.shape[0]):
        want = O.embedding_largest_coeff(m, f[r])
        assert got[r] == pytest.approx(want, rel=NORM_RTOL, abs=1e-300), (m, r)


@pytest.mark.parametrize("m", [12, 105, 1705, 21845, 65539])
def test_embedding_norm_general_m_matches_oracle(hx, m):
    """General m: complex-double Bluestein on the device (norm_kernels.h bnorm_*; 2^bk-point
    transforms in sub-transforms of <= 8192 points) against the oracle's long-double evaluation of
    max_j |f(omega^j)|, j in Z_m^* (src/norms.cpp:480-493 via PGFFT in the reference)."""
    g = hx.Context(m)
    rng = np.random.default_rng(m)
    n = g.phim
    rows = 3 if m > 20000 else 6
    f = rng.normal(0, 1000.0, size=(rows, n))
    f[0] = np.rint(rng.normal(0, 3.2, size=n))                 # an error polynomial
    f[1] = 0.0
    f[1, n // 3] = 1.0                                         # a monomial: every |f(omega^j)| = 1
    got = hx.embeddingLargestCoeff(g, f)
    want = np.array([O.embedding_largest_coeff(m, r) for r in f])
    assert abs(got[1] - 1.0) < 1e-12
    assert np.allclose(got, want, rtol=NORM_RTOL, atol=0), (got, want)


def test_embedding_norm_unsupported_beyond_2_18(hx):
    g = hx.Context(131073)
    with pytest.raises(hx.HxError) as ei:
        hx.embeddingLargestCoeff(g, np.ones(g.phim))
    assert ei.value.code == hx.HX_ERR_UNSUPPORTED


@pytest.mark.parametrize("ndrop,ptxt", [(1, 65537), (2, 65537), (1, 2), (2, 1), (1, 1)])
def test_scale_down_norms_and_fdelta(hx, ndrop, ptxt):
    """Rows bit-exact as without norms; fdelta and embeddingLargestCoeff(fdelta) against the
    oracle for the fused single-prime path and the generic path."""
    P, own, sp = setup_rns(hx, m=16384, L=5, K=2)
    allp = own + sp
    drop = allp[-ndrop:]
    keep = [i for i in allp if i not in drop]
    parts = [P.rand(allp, 60 + i, batch=2) for i in range(3)]
    polys = [hx.DoubleCRT(P.g, allp, 2, x) for x in parts]
    norms, fd = hx.scaleDownToSetMulti(polys, keep, ptxt, norms=True, fdelta=True)
    assert norms.shape == (3, 2) and fd.shape == (3, 2, P.N)
    for k, (x, d) in enumerate(zip(parts, polys)):
        idx = d.getIndexSet()
        got = d.download()
        for b in range(2):
            want, wfd = P.o.scale_down(allp, x[:, b], drop, ptxt, want_fdelta=True)
            for r, i in enumerate(idx):
                assert np.array_equal(got[r, b], want[keep.index(i)])
            assert np.abs(fd[k, b] - wfd).max() <= 1e-9 * (ptxt / 2 + 1)
            assert norms[k, b] == pytest.approx(O.embedding_largest_coeff(P.o.m, wfd), rel=NORM_RTOL)


def test_bring_to_set_norms(hx):
    P, own, sp = setup_rns(hx, m=16384, L=5, K=2)
    add, drop = [sp[0]], [own[-1]]
    keep = [i for i in own + add if i not in drop]
    parts = [P.rand(own, 70 + i, batch=2) for i in range(2)]
    polys = [hx.DoubleCRT(P.g, own, 2, x) for x in parts]
    plain = [hx.DoubleCRT(P.g, own, 2, x) for x in parts]
    norms = hx.bringToSetMulti(polys, add, keep, 65537, norms=True)
    hx.bringToSetMulti(plain, add, keep, 65537)
    for k, (x, d, e) in enumerate(zip(parts, polys, plain)):
        assert d.getIndexSet() == e.getIndexSet() and np.array_equal(d.download(), e.download())
        for b in range(2):
            up = np.vstack([P.o.scale_by_primes(own, x[:, b], add), np.zeros((1, P.N), dtype=np.uint64)])
            _, wfd = P.o.scale_down(own + add, up, drop, 65537, want_fdelta=True)
            assert norms[k, b] == pytest.approx(O.embedding_largest_coeff(P.o.m, wfd), rel=NORM_RTOL)


@pytest.mark.parametrize("digits", [[[0, 1], [2, 3], [4]], [[0, 1, 2, 3, 4]]])
def test_break_into_digits_and_relinearize_norms(hx, digits):
    P, own, sp = setup_rns(hx, m=16384, L=5, K=2)
    allp = own + sp
    a = P.rand(own, 80, batch=2)
    d = hx.DoubleCRT(P.g, own, 2, a)
    dg, nrm = d.breakIntoDigits(digits, sp, norms=True)
    assert np.array_equal(dg.download(), d.breakIntoDigits(digits, sp).download())
    want = [P.o.break_into_digits(own, a[:, b], digits, allp, want_norms=True)[1] for b in range(2)]
    for k in range(len(digits)):
        for b in range(2):
            assert nrm[k, b] == pytest.approx(want[b][k], rel=NORM_RTOL)
    # the same numbers out of the fused reLinearize path
    D = len(digits)
    kb = np.stack([P.rand(allp, 20 + i)[:, 0] for i in range(D)])
    ka = np.stack([P.rand(allp, 30 + i)[:, 0] for i in range(D)])
    W = hx.KeySwitch(P.g, allp, kb, ka)
    t0, t1 = (hx.DoubleCRT(P.g, own, 2, P.rand(own, s, 2)) for s in (81, 82))
    o0, o1, nrm2 = hx.reLinearize(t0, t1, d, W, digits, sp, norms=True)
    p0, p1 = hx.reLinearize(t0, t1, d, W, digits, sp)
    assert np.array_equal(o0.download(), p0.download()) and np.array_equal(o1.download(), p1.download())
    assert np.allclose(nrm2, nrm, rtol=1e-12, atol=0)


# ---------------------------------------------------------------- N4: rotation path
@pytest.mark.parametrize("m,p,bits,k", [(16384, 65537, 250, 3), (16384, 65537, 250, 16383), (1705, 7, 200, 2)])
def test_smartAutomorph_gpu_vs_oracle(hx, m, p, bits, k, monkeypatch):
    """Ctxt::smartAutomorph (automorph + reLinearize with the s(X^k) -> s matrix,
    src/Ctxt.cpp:2437-2515): hx_automorph on both parts, hx_relinearize with no s part, driven by
    the same host logic on the GPU and on the oracle; parts bit-identical, decrypts to m(X^k)."""
    from helib_amd import ctxt as hc
    from tests import test_ctxt_host as T
    from oracle.backend import OKeySwitch, OPoly, OracleOps
    monkeypatch.setattr(hc.Ctxt, "measure", hx.supportsNorms(m))
    ctx = hc.ChainContext(m, p, 1, bits=bits, c=3)
    P = Pair(hx, m, ctx.primes)
    s, allp, kb, ka, rows = T.make_keys(ctx, P.o)
    _, _, kbk, kak, _ = T.make_keys(ctx, P.o, auto_k=k)
    rng = np.random.default_rng(5)
    ma = rng.integers(0, p, size=P.N)
    ea = T.encrypt(ctx, P.o, s, ma, 1, rows)
    oa = hc.Ctxt.fresh(ctx, OracleOps(P.o), *(OPoly(P.o, ctx.ctxtPrimes, x) for x in ea), ksw=OKeySwitch(allp, kb, ka))
    oa.ksw_auto = {k: OKeySwitch(allp, kbk, kak)}
    oa.smartAutomorph(k)
    ga = hc.Ctxt.fresh(ctx, hx, *(hx.DoubleCRT(P.g, ctx.ctxtPrimes, 1, x[:, None, :]) for x in ea),
                       ksw=hx.KeySwitch(P.g, allp, kb, ka))
    ga.ksw_auto = {k: hx.KeySwitch(P.g, allp, kbk, kak)}
    ga.smartAutomorph(k)
    assert ga.primeSet == oa.primeSet and abs(ga.lnNoise - oa.lnNoise) < 1e-8
    for h in ("1", "s"):
        gi, oi = ga.parts[h].getIndexSet(), oa.parts[h].getIndexSet()
        assert sorted(gi) == sorted(oi)
        gd, od = ga.parts[h].download()[:, 0], oa.parts[h].download()[:, 0]
        for r, i in enumerate(gi):
            assert np.array_equal(gd[r], od[oi.index(i)]), (h, i)
    got = T.decrypt(ctx, P.o, s, ga, rows)
    if m < 4096:
        from tests import bgv_ref as B
        assert got == B.automorph_mod_phi(ma, m, k, p)
    else:
        assert got == T.decrypt(ctx, P.o, s, oa, rows)


# ---------------------------------------------------------------- N3: wire format
def test_wire_format_round_trip_through_device(hx):
    """DoubleCRT::writeTo -> read (helib_amd/wire.py) through device-resident objects."""
    from helib_amd import wire
    P = Pair(hx, 4096, primes_for(4096, 4))
    idx = [2, 0, 3]                                    # device row order need not be ascending
    x = P.rand(idx, 3, batch=2)
    d = hx.DoubleCRT(P.g, idx, 2, x)
    for b in range(2):
        raw = wire.writeTo(d, b)
        i2, rows, off = wire.read_rows(raw)
        assert off == len(raw) and i2 == [0, 2, 3]
        assert np.array_equal(rows, x[[1, 0, 2], b])
    e, off = wire.readFrom(hx, P.g, wire.writeTo(d, 1), batch=3)
    assert e.getIndexSet() == [0, 2, 3] and e.batch == 3
    got = e.download()
    for b in range(3):
        assert np.array_equal(got[:, b], x[[1, 0, 2], 1])
    e.FFT()                                            # a loaded object is a normal DoubleCRT
    assert np.array_equal(e.download()[:, 0], P.o.fft([0, 2, 3], x[[1, 0, 2], 1]))


def test_set_constant_and_exp(hx):
    """DoubleCRT::operator=(ZZ) (src/DoubleCRT.cpp:866-884) and DoubleCRT::Exp (:1142-1156)."""
    P = Pair(hx, 8192, primes_for(8192, 3))
    idx = [0, 1, 2]
    x = P.rand(idx, 11, batch=2)
    d = hx.DoubleCRT(P.g, idx, 2, x)
    for e in (0, 1, 2, 5, 65537):
        d.upload(x)
        got = d.Exp(e).download()
        for r, i in enumerate(idx):
            q = P.primes[i]
            want = np.array([pow(int(v), e, q) for v in x[r, 0, :64]], dtype=np.uint64)
            assert np.array_equal(got[r, 0, :64], want)
            assert got[r, 1, 100] == pow(int(x[r, 1, 100]), e, q)
    big = (1 << 190) + 77
    got = d.setConstant(big).download()
    for r, i in enumerate(idx):
        assert (got[r] == big % P.primes[i]).all()
    # a constant polynomial evaluates to the constant everywhere: iFFT gives (c, 0, 0, ...)
    back = d.iFFT().download()
    for r, i in enumerate(idx):
        assert back[r, 0, 0] == big % P.primes[i] and not back[r, 0, 1:].any()


def test_scale_down_norms_when_parts_are_switched_one_by_one(hx):
    """Parts whose rows sit in different orders cannot share the fused launches: every part then
    runs the single-prime path on its own and reuses the same (x, S) scratch -- the norms must
    still be those of each part's own delta."""
    P, own, sp = setup_rns(hx, m=16384, L=4, K=1)
    allp = own + sp
    orders = [allp, [allp[1], allp[0]] + allp[2:], allp[::-1][1:] + [allp[-1]]]
    parts = [P.rand(o, 90 + i, batch=2) for i, o in enumerate(orders)]
    polys = [hx.DoubleCRT(P.g, o, 2, x) for o, x in zip(orders, parts)]
    keep = own
    norms = hx.scaleDownToSetMulti(polys, keep, 65537, norms=True)
    for k, (o, x, d) in enumerate(zip(orders, parts, polys)):
        idx = d.getIndexSet()
        got = d.download()
        for b in range(2):
            want, wfd = P.o.scale_down(o, x[:, b], sp, 65537, want_fdelta=True)
            kept = [i for i in o if i in keep]
            for r, i in enumerate(idx):
                assert np.array_equal(got[r, b], want[kept.index(i)])
            assert norms[k, b] == pytest.approx(O.embedding_largest_coeff(P.o.m, wfd), rel=NORM_RTOL)


# ---------------------------------------------------------------- edge cases and maximum sizes
def test_empty_index_set_and_single_element_batch(hx):
    """A DoubleCRT on the empty prime set (the reference constructs these, e.g.
    src/DoubleCRT.cpp:506) and the degenerate calls on it."""
    P = Pair(hx, 4096, primes_for(4096, 3))
    e = hx.DoubleCRT(P.g, [], 1)
    assert e.getIndexSet() == [] and e.download().shape[0] == 0
    e.FFT().iFFT().Negate()
    e += e
    e *= e
    e.automorph(3)
    full = hx.DoubleCRT(P.g, [0, 1, 2], 1, P.rand([0, 1, 2], 1))
    before = full.download()
    e += full                              # empty set is a subset of every set
    with pytest.raises(hx.HxError) as ei:
        full += e                          # ... but not the other way round
    assert ei.value.code == hx.HX_ERR_PRIMESET
    assert np.array_equal(full.download(), before)
    # nothing to drop / nothing to add
    full.scaleDownToSet([0, 1, 2], 65537)
    full.addPrimes([])
    full.addPrimesAndScale([])
    assert np.array_equal(full.download(), before) and full.getIndexSet() == [0, 1, 2]
    with pytest.raises(hx.HxError):
        full.scaleDownToSet([], 65537)     # "s and the index set must have some intersection"
    with pytest.raises(hx.HxError):
        hx.DoubleCRT(P.g, [0, 0], 1)       # repeated prime
    with pytest.raises(hx.HxError):
        hx.DoubleCRT(P.g, [7], 1)          # prime index out of range


def test_multiply_relin_at_the_reference_benchmark_chain_size(hx):
    """The reference's own bgv_basic parameter (bits=6400 at m=32768, benchmarks/bgv_basic.cpp:247):
    L=107 ctxt primes, K=36 special primes, D=3 digits of 36/36/35 -- 143 rows, digits wider than
    the 16-prime fast path (rns_extend_kernel<40>), batch 1.  Against the oracle, bit-exact."""
    m, L, K = 32768, 107, 36
    digits = [list(range(0, 36)), list(range(36, 72)), list(range(72, 107))]
    P = Pair(hx, m, primes_for(m, L + K, 60))
    own, sp = list(range(L)), list(range(L, L + K))
    allp = own + sp
    c0, c1, d0, d1 = (P.rand(own, s, 1) for s in (1, 2, 3, 4))
    kb = np.stack([P.rand(allp, 20 + i)[:, 0] for i in range(3)])
    ka = np.stack([P.rand(allp, 30 + i)[:, 0] for i in range(3)])
    W = hx.KeySwitch(P.g, allp, kb, ka)
    G = [hx.DoubleCRT(P.g, own, 1, x) for x in (c0, c1, d0, d1)]
    o0, o1 = hx.multiplyBy(*G, W, digits)
    w0, w1 = P.o.mul_relin(own, sp, digits, c0[:, 0], c1[:, 0], d0[:, 0], d1[:, 0], kb, ka)
    assert o0.getIndexSet() == allp
    assert np.array_equal(o0.download()[:, 0], w0)
    assert np.array_equal(o1.download()[:, 0], w1)
    # and the mod-down of the result by all 36 special primes at once (generic path, 36 sources)
    o0.scaleDownToSet(own, 65537)
    assert np.array_equal(o0.download()[:, 0], P.o.scale_down(allp, w0, sp, 65537))


def test_fresh_multiplyBy_at_the_reference_benchmark_chain_size(hx, monkeypatch):
    """benchmarks/bgv_basic.cpp:247 (m=32768, p=65537, bits=6400: L=107, K=36, three digits of 36/36/35
    primes) -- the FRESH Ctxt::multiplyBy sequence (both bringToSet mod-switches, tensorProduct,
    dropSmallAndSpecialPrimes, reLinearize), not only hx_mul_relin at a fixed level: device vs
    oracle under the same host logic, every row, and the product decrypts."""
    from helib_amd import ctxt as hc
    from tests import test_ctxt_host as T
    from oracle.backend import OKeySwitch, OPoly, OracleOps
    m, p = 32768, 65537
    monkeypatch.setattr(hc.Ctxt, "measure", True)
    spy = FusedCallSpy(hx, monkeypatch)
    ctx = hc.ChainContext(m, p, 1, bits=6400, c=3)
    assert len(ctx.ctxtPrimes) == 107 and len(ctx.specialPrimes) == 36
    P = Pair(hx, m, ctx.primes)
    s, allp, kb, ka, rows = T.make_keys(ctx, P.o)
    rng = np.random.default_rng(41)
    ma, mb = rng.integers(0, p, size=P.N), rng.integers(0, p, size=P.N)
    ea, eb = T.encrypt(ctx, P.o, s, ma, 1, rows), T.encrypt(ctx, P.o, s, mb, 2, rows)
    oops, oW = OracleOps(P.o), OKeySwitch(allp, kb, ka)
    oa = hc.Ctxt.fresh(ctx, oops, *(OPoly(P.o, ctx.ctxtPrimes, x) for x in ea), ksw=oW)
    ob = hc.Ctxt.fresh(ctx, oops, *(OPoly(P.o, ctx.ctxtPrimes, x) for x in eb), ksw=oW)
    oa.multiplyBy(ob)
    gW = hx.KeySwitch(P.g, allp, kb, ka)
    ga = hc.Ctxt.fresh(ctx, hx, *(hx.DoubleCRT(P.g, ctx.ctxtPrimes, 1, x[:, None, :]) for x in ea), ksw=gW)
    gb = hc.Ctxt.fresh(ctx, hx, *(hx.DoubleCRT(P.g, ctx.ctxtPrimes, 1, x[:, None, :]) for x in eb), ksw=gW)
    ga.multiplyBy(gb)
    assert ga.primeSet == oa.primeSet and ga.intFactor == oa.intFactor
    assert abs(ga.lnNoise - oa.lnNoise) < 1e-8
    for h in ("1", "s"):
        gi, oi = ga.parts[h].getIndexSet(), oa.parts[h].getIndexSet()
        assert sorted(gi) == sorted(oi)
        gd, od = ga.parts[h].download()[:, 0], oa.parts[h].download()[:, 0]
        for r, i in enumerate(gi):
            assert np.array_equal(gd[r], od[oi.index(i)]), (h, i)
    full = np.convolve(ma.astype(np.int64), mb.astype(np.int64))
    want = (full[:P.N] - np.append(full[P.N:], 0)) % p
    assert T.decrypt(ctx, P.o, s, oa, rows) == [int(v) for v in want]
    assert spy.names() == ["tensorBringToSet"]
    # level 2: both operands carry the 36 special primes -- the several-primes mod-switch drops them through
    # rns_extend_wide_kernel<40> on a plan that carries P^-1 and the ptxtSpace correction
    ga.multiplyBy(ga.clone())
    oa.multiplyBy(oa.clone())
    assert ga.primeSet == oa.primeSet and ga.intFactor == oa.intFactor
    assert abs(ga.lnNoise - oa.lnNoise) < 1e-8
    for h in ("1", "s"):
        gi, oi = ga.parts[h].getIndexSet(), oa.parts[h].getIndexSet()
        assert sorted(gi) == sorted(oi)
        gd, od = ga.parts[h].download()[:, 0], oa.parts[h].download()[:, 0]
        for r, i in enumerate(gi):
            assert np.array_equal(gd[r], od[oi.index(i)]), (h, i)
    assert spy.names() == ["tensorBringToSet", "tensorBringToSet"]


# ---------------------------------------------------------------- copy-on-write DoubleCRT copies
def test_lazy_copies_are_copy_on_write(hx):
    """hx_poly_copy shares the source's slab until somebody writes (engine.hip, hx_poly::Share).
    Every mutating entry point must leave all other holders untouched, and must give the result
    it gives on an independent object -- whichever side (copy or source) is written, with the
    fused mod-switch / transforms / automorphism taking their out-of-place routes."""
    P, own, sp = setup_rns(hx, m=16384, L=5, K=2)
    allp = own + sp
    B = 2
    base = P.rand(allp, 77, batch=B)
    other = P.rand(allp, 78, batch=B)
    key = bytes(range(32))

    def fresh():        # an independent object with the same rows (uploaded, shares nothing)
        return hx.DoubleCRT(P.g, allp, B, base)

    o = hx.DoubleCRT(P.g, allp, B, other)
    ops = {
        "iFFT": lambda d: d.iFFT(),
        "FFT": lambda d: d.FFT(),
        "iadd": lambda d: d.__iadd__(o),
        "imul": lambda d: d.__imul__(o),
        "Negate": lambda d: d.Negate(),
        "mulConstant": lambda d: d.mulConstant(12345),
        "addConstant": lambda d: d.addConstant(7),
        "setConstant": lambda d: d.setConstant(3),
        "Exp": lambda d: d.Exp(3),
        "automorph": lambda d: d.automorph(5),
        "removePrimes": lambda d: d.removePrimes([allp[1]]),
        "scaleDown1": lambda d: d.scaleDownToSet([i for i in allp if i != allp[2]], 65537),   # fused path
        "scaleDown2": lambda d: d.scaleDownToSet(own, 65537),                                # generic path
        "randomize": lambda d: d.randomize(key, 9),
        "upload": lambda d: d.upload(other),
        "selfmul": lambda d: d.__imul__(d),
    }

    def rows_by_prime(d):
        return dict(zip(d.getIndexSet(), d.download()))

    def same(x, y):
        a, b = rows_by_prime(x), rows_by_prime(y)
        return sorted(a) == sorted(b) and all(np.array_equal(a[i], b[i]) for i in a)

    for name, op in ops.items():
        want = fresh()
        op(want)
        # write the copy: the source keeps its rows
        src = fresh()
        cp = src.copy()
        cp2 = cp.copy()                 # a chain of copies
        op(cp)
        assert np.array_equal(src.download(), base), name
        assert np.array_equal(cp2.download(), base), name
        assert same(cp, want), name
        # write the source: the copies keep theirs
        op(src)
        assert same(src, want), name
        assert np.array_equal(cp2.download(), base), name
        # the last holder works in place (no other holder left to disturb)
        del src, cp
        op(cp2)
        assert same(cp2, want), name
    # subset operands
    low = hx.DoubleCRT(P.g, own, B, base[:len(own)])
    for add_then in ("addPrimesAndScale", "addPrimes"):
        want = hx.DoubleCRT(P.g, own, B, base[:len(own)])
        getattr(want, add_then)(sp)
        cp = low.copy()
        getattr(cp, add_then)(sp)
        assert np.array_equal(low.download(), base[:len(own)]) and same(cp, want)
    # the fused bringToSet on several parts, two of which share ONE slab, one a lazy copy of a
    # third object, one exclusive
    x, y = hx.DoubleCRT(P.g, own, B, base[:len(own)]), hx.DoubleCRT(P.g, own, B, other[:len(own)])
    parts = [x.copy(), x.copy(), y.copy(), hx.DoubleCRT(P.g, own, B, other[:len(own)])]
    keep = [i for i in own + [sp[0]] if i != own[-1]]
    hx.bringToSetMulti(parts, [sp[0]], keep, 65537)
    wx, wy = hx.DoubleCRT(P.g, own, B, base[:len(own)]), hx.DoubleCRT(P.g, own, B, other[:len(own)])
    hx.bringToSetMulti([wx, wy], [sp[0]], keep, 65537)
    assert np.array_equal(x.download(), base[:len(own)]) and np.array_equal(y.download(), other[:len(own)])
    assert same(parts[0], wx) and same(parts[1], wx) and same(parts[2], wy) and same(parts[3], wy)
    # outputs that are lazy copies
    t = [hx.DoubleCRT(P.g, own, B, P.rand(own, 80 + i, batch=B)) for i in range(4)]
    w0, w1, w2 = hx.tensorProduct(*t)
    spare = hx.DoubleCRT(P.g, own, B, base[:len(own)])
    held = spare.copy()
    assert np.array_equal(held.download(), base[:len(own)])


# ---------------------------------------------------------------- DoubleCRT::randomize
@pytest.mark.parametrize("m,batch", [(16384, 3), (32768, 2), (128, 2), (1705, 1)])
def test_randomize_on_the_device_matches_oracle(hx, m, batch):
    """hx_randomize = DoubleCRT::randomize (src/DoubleCRT.cpp:1258-1378) on the device: the
    reference's rejection sampling (2048-byte buffers, ceil(k/8) bytes per candidate, k-bit mask,
    keep when < q) from one ChaCha20 stream per (row, batch element).  Against the oracle's
    restatement on every word -- HElib-style primes just below 2^60 / 2^56 / 2^40 (a rejection is a
    2^-30 event), and primes just ABOVE a power of two, where every second candidate is rejected
    and the order-preserving compaction and the extra buffers are exercised."""
    if m & (m - 1) == 0:
        near = [O.PrimeGen(60, m).next(), O.PrimeGen(56, m).next(), O.PrimeGen(40, m).next()]
        step = 2 * m
        above = []
        for lo in (1 << 59, 1 << 47, (1 << 33) + (1 << 31)):
            q = lo - (lo % step) + step + 1
            while not O.lib().ho_is_prime(q):
                q += step
            above.append(q)
        primes = near + above
    else:
        g = O.PrimeGen(60, m)
        primes = [g.next(), g.next()]
    P = Pair(hx, m, primes)
    idx = list(range(len(primes)))
    key = bytes((7 * i + 3) & 0xff for i in range(32))
    for stream in (1, (5 << 32) | 9):
        d = hx.DoubleCRT(P.g, idx, batch, zero=False).randomize(key, stream)
        got = d.download()
        for r, i in enumerate(idx):
            for b in range(batch):
                want, nbuf = O.randomize_row(P.N, primes[i], key, stream, i, b)
                assert np.array_equal(got[r, b], want), (primes[i], b, stream)
                assert got[r, b].max() < primes[i]
    # distinct streams / rows / batch elements are distinct
    a = hx.DoubleCRT(P.g, idx, batch, zero=False).randomize(key, 1).download()
    assert not np.array_equal(a, got)
    assert not np.array_equal(a[0, 0], a[0, batch - 1]) or batch == 1


# ---------------------------------------------------------------- keys, encrypt, decrypt
@pytest.mark.parametrize("m,p,bits", [(16384, 65537, 250), (128, 257, 150), (1705, 7, 200)])
def test_keys_encrypt_decrypt_gpu_vs_oracle(hx, m, p, bits, monkeypatch):
    """helib_amd.keys (GenSecKey, GenKeySWmatrix, PubKey::Encrypt, multiplyBy, smartAutomorph,
    SecKey::Decrypt; src/keys.cpp:39-85, 358-488, 1099-1255, 1327-1420) run twice from the same
    seed -- DoubleCRT operations on the GPU vs the oracle backend: every key row, ciphertext
    part and decrypted coefficient must be identical."""
    from helib_amd import ctxt as hc, keys as hk
    from oracle.backend import OracleBackend
    monkeypatch.setattr(hc.Ctxt, "measure", hx.supportsNorms(m))
    cc = hc.ChainContext(m, p, 1, bits=bits, c=3)
    P = Pair(hx, m, cc.primes)
    gsk = hk.SecKey(cc, hk.HxBackend(P.g, cc), seed=11)
    osk = hk.SecKey(cc, OracleBackend(P.o, cc), seed=11)
    for sk in (gsk, osk):
        sk.GenSecKey(maxDegKswitch=2)
        sk.GenKeySWmatrix(1, 3)
    assert np.array_equal(gsk.sKeys[0], osk.sKeys[0])

    def same(gp, op):
        gi, oi = gp.getIndexSet(), op.getIndexSet()
        assert sorted(gi) == sorted(oi)
        gd, od = gp.download()[:, 0], op.download()[:, 0]
        for r, i in enumerate(gi):
            assert np.array_equal(gd[r], od[oi.index(i)]), i

    for k in range(2):
        same(gsk.pubEncrKey[k], osk.pubEncrKey[k])
    for key in ((2, 1), (1, 3)):
        g, o = gsk.getKeySWmatrix(*key), osk.getKeySWmatrix(*key)
        assert np.array_equal(g.b, o.b) and np.array_equal(g.a, o.a) and g.noiseBound == o.noiseBound
    rng = np.random.default_rng(8)
    ma, mb = rng.integers(0, p, size=cc.phim), rng.integers(0, p, size=cc.phim)
    ga, gb, oa, ob = gsk.Encrypt(ma), gsk.Encrypt(mb), osk.Encrypt(ma), osk.Encrypt(mb)
    for h in ("1", "s"):
        same(ga.parts[h], oa.parts[h])
    assert gsk.Decrypt(ga) == [int(v) for v in ma] == osk.Decrypt(oa)
    ga.multiplyBy(gb)
    oa.multiplyBy(ob)
    assert ga.primeSet == oa.primeSet and ga.intFactor == oa.intFactor
    for h in ("1", "s"):
        same(ga.parts[h], oa.parts[h])
    prod = gsk.Decrypt(ga)
    assert prod == osk.Decrypt(oa)
    ga.smartAutomorph(3)
    oa.smartAutomorph(3)
    for h in ("1", "s"):
        same(ga.parts[h], oa.parts[h])
    rot = gsk.Decrypt(ga)
    assert rot == osk.Decrypt(oa)
    # plaintext constants (Ctxt::multByConstant / addConstant, src/Ctxt.cpp:896-935, 1832-1856)
    allp = list(cc.ctxtPrimes) + list(cc.specialPrimes)
    bal = np.array([int(x) - p if int(x) > p // 2 else int(x) for x in mb], dtype=np.int64)
    for sk, ct in ((gsk, ga), (osk, oa)):
        ct.multByConstant(sk.be.fromCoeffs(allp, bal))
        ct.addConstant(sk.be.fromCoeffs(allp, bal))
    for h in ("1", "s"):
        same(ga.parts[h], oa.parts[h])
    assert abs(ga.lnNoise - oa.lnNoise) < 1e-8
    withc = gsk.Decrypt(ga)
    assert withc == osk.Decrypt(oa)
    if m < 4096:
        from tests import bgv_ref as B
        want = [int(v) for v in B.polymul_mod_phi(ma, mb, m, p)]
        assert prod == want and rot == [int(v) for v in B.automorph_mod_phi(want, m, 3, p)]
        assert withc == [(int(x) + int(y)) % p for x, y in zip(B.polymul_mod_phi(rot, mb, m, p), mb)]


@pytest.mark.parametrize("m,precision,bits,c", [(16384, 20, 300, 2), (128, 20, 200, 2),
                                                (65536, 20, 1400, 3)])   # BASELINE configs[3]: L=24, K=8, D=3
def test_ckks_encrypt_multiply_decrypt_gpu_vs_oracle(hx, m, precision, bits, c, monkeypatch):
    """The CKKS chain of the same path (BASELINE configs[3]; PubKey::CKKSencrypt src/keys.cpp:501-581,
    the CKKS branches of computeIntervalForMul / tensorProduct / relin_CKKS_adjust in src/Ctxt.cpp):
    device vs oracle backend from one seed -- identical parts, prime sets and raw decryptions, and
    the decoded product is the real product to the promised precision."""
    from helib_amd import ctxt as hc, keys as hk
    from oracle.backend import OracleBackend
    monkeypatch.setattr(hc.Ctxt, "measure", True)
    cc = hc.ChainContext(m, -1, precision, bits=bits, c=c, ckks=True)
    if m == 65536:   # the benchmark chain of benchmarks/ckks_basic.cpp at the "~24 primes" of BASELINE
        assert (len(cc.ctxtPrimes), len(cc.specialPrimes), len(cc.digits)) == (24, 8, 3)
    P = Pair(hx, m, cc.primes)
    gsk = hk.SecKey(cc, hk.HxBackend(P.g, cc), seed=13)
    osk = hk.SecKey(cc, OracleBackend(P.o, cc), seed=13)
    for sk in (gsk, osk):
        sk.GenSecKey(maxDegKswitch=2)
    n = cc.phim
    rng = np.random.default_rng(4)
    a, b = rng.uniform(-1, 1, n) / n, rng.uniform(-1, 1, n) / n
    f = float(1 << precision)
    pa, pb = np.rint(a * f).astype(np.int64), np.rint(b * f).astype(np.int64)
    ga, gb = gsk.CKKSencrypt(pa, 1.0, f), gsk.CKKSencrypt(pb, 1.0, f)
    oa, ob = osk.CKKSencrypt(pa, 1.0, f), osk.CKKSencrypt(pb, 1.0, f)

    def same(gc, oc):
        assert gc.primeSet == oc.primeSet and set(gc.parts) == set(oc.parts)
        assert abs(gc.lnRatFactor - oc.lnRatFactor) < 1e-9 and abs(gc.lnNoise - oc.lnNoise) < 1e-7
        for h in gc.parts:
            gi, oi = gc.parts[h].getIndexSet(), oc.parts[h].getIndexSet()
            gd, od = gc.parts[h].download()[:, 0], oc.parts[h].download()[:, 0]
            for r, i in enumerate(gi):
                assert np.array_equal(gd[r], od[oi.index(i)]), (h, i)

    same(ga, oa)
    for g, o in ((ga, gb), (oa, ob)):
        g.multiplyBy(o)                   # level 1: fresh x fresh (no mod-switch; tensor + key switch)
    same(ga, oa)
    for g in (ga, oa):
        g.multiplyBy(g.clone())           # level 2, (a*b)^2: the several-primes mod-down first
    same(ga, oa)
    raw = gsk.Decrypt(ga)
    assert raw == osk.Decrypt(oa)
    if m <= 1024:
        got = np.array([float(v) for v in raw]) / math.exp(ga.lnRatFactor)
        ab = np.convolve(a, b)
        ab = ab[:n] - np.append(ab[n:], 0.0)
        w = np.convolve(ab, ab)
        want = w[:n] - np.append(w[n:], 0.0)
        assert np.max(np.abs(got - want)) < 2.0 ** (-precision + 6) / n


def test_half_row_forward_transform_at_2p15_probe(hx, monkeypatch):
    """HX_HALF15=1: the forward transform of the digit rows of hx_mul_relin at N = 2^15 as two 2^14-point workgroups per
    row (ntt_kernels.hip ntt_row_half15_kernel: first Cooley-Tukey stage in the load, the sub-transform's tables,
    interleaved outputs; Cmodulus::FFT, src/CModulus.cpp:389-426) -- measured slower than the one-workgroup kernel and off
    by default (profiles/r06_ab_half_row_forward_2p15.json); the probe must stay bit-exact: the reference's own CKKS
    parameters through it, and the kernel that ran checked by name."""
    monkeypatch.setenv("HX_HALF15", "1")
    hx.profileBegin()
    test_ckks_m65536_chain_batched_bit_exact(hx, monkeypatch, 1, 440, (8, 3, 3))
    names = " ".join(k["kernel"] for k in hx.profileEnd()["kernels"])
    assert "ntt_row_half15_kernel<8>" in names, names


@pytest.mark.parametrize("precision,bits,shape", [(20, 1400, (24, 8, 3)), (1, 1400, (24, 8, 3)), (1, 440, (8, 3, 3))])
def test_ckks_m65536_chain_batched_bit_exact(hx, monkeypatch, precision, bits, shape):
    """BASELINE configs[3] as bench.py --workload ckks65536 runs it: m = 65536, bits = 1400 (L = 24, K = 8, D = 3) --
    and the reference's own benchmark parameters, ContextBuilder<CKKS>().m(65536).precision(1).bits(440).scale(10)
    (benchmarks/ckks_basic.cpp:263, benchmarks/ckks_common.h:45-50: L = 8 primes of 56 bits, K = 3 of 55) --
    a BATCH of two distinct CKKSencrypt-ed pairs packed along the batch axis, multiplied at level 1 (fresh x fresh:
    hx_mul_relin_norms, the product formed inside ntt_inv_mul_kernel<15> and the key-switch kernel) and level 2
    (product x product: the several-primes mod-switch of the operands at N = 32768, then hx_tensor_bring_to_set_norms
    = ntt_moddown_prep_multi_tensor_kernel + ntt_moddown_apply_tensor_kernel<15,true>) in one set of launches -- the
    call spy pins these entry points; the oracle backend runs the same host logic once per pair over the
    reference's unfused sequence.  Every part, row and batch element must agree word for word (the prime-set
    decisions are the batch's: its noise estimate is the largest element's), and the level-2 result decodes to the
    real product of the encoded values within the bound the ciphertext reports."""
    from helib_amd import ctxt as hc, keys as hk
    from oracle.backend import OracleBackend
    monkeypatch.setattr(hc.Ctxt, "measure", True)
    spy = FusedCallSpy(hx, monkeypatch)
    m, B = 65536, 2
    cc = hc.ChainContext(m, -1, precision, bits=bits, c=3, ckks=True)
    assert (len(cc.ctxtPrimes), len(cc.specialPrimes), len(cc.digits)) == shape
    P = Pair(hx, m, cc.primes)
    gsk = hk.SecKey(cc, hk.HxBackend(P.g, cc), seed=21)
    osk = hk.SecKey(cc, OracleBackend(P.o, cc), seed=21)
    for sk in (gsk, osk):
        sk.GenSecKey(maxDegKswitch=2)
    n, L = cc.phim, len(cc.ctxtPrimes)
    rng = np.random.default_rng(9)
    # the factor PubKey::Encrypt(Ptxt<CKKS>) encodes with (EncryptedArrayCx::encodeScalingFactor: 2^11 at
    # precision(1), 2^30 at precision(20)) and real coefficients whose canonical embedding stays below the declared
    # size 1 -- the plaintexts of helib_amd/csrc/host_session.cpp
    f = float(cc.encodeScalingFactor())
    assert f == {1: 2.0 ** 11, 20: 2.0 ** 30}[precision]
    vals = rng.uniform(-1, 1, size=(2, B, n)) / (8.0 * math.sqrt(n / 3.0))
    genc = [[gsk.CKKSencrypt(np.rint(vals[j, b] * f).astype(np.int64), 1.0, f) for b in range(B)] for j in range(2)]
    oenc = [[osk.CKKSencrypt(np.rint(vals[j, b] * f).astype(np.int64), 1.0, f) for b in range(B)] for j in range(2)]
    ops = []
    for j in range(2):
        c = genc[j][0].clone()
        c.lnNoise = max(x.lnNoise for x in genc[j])
        c.parts = {h: hx.DoubleCRT(P.g, list(cc.ctxtPrimes), B,
                                   np.stack([genc[j][b].parts[h].download()[:, 0] for b in range(B)], axis=1))
                   for h in ("1", "s")}
        ops.append(c)
    ga, gb = ops

    def same(gc, ocs):
        for b, oc in enumerate(ocs):
            assert gc.primeSet == oc.primeSet and set(gc.parts) == set(oc.parts)
            for h in gc.parts:
                gi, oi = gc.parts[h].getIndexSet(), oc.parts[h].getIndexSet()
                gd, od = gc.parts[h].download()[:, b], oc.parts[h].download()[:, 0]
                for r, i in enumerate(gi):
                    assert np.array_equal(gd[r], od[oi.index(i)]), (b, h, i)

    ga.multiplyBy(gb)
    for b in range(B):
        oenc[0][b].multiplyBy(oenc[1][b])
    same(ga, oenc[0])
    # the batch takes its prime-set decision from its largest noise estimate: give each oracle element that estimate
    worst = max(o.lnNoise for o in oenc[0])
    assert abs(ga.lnNoise - worst) < 1e-7
    for o in oenc[0]:
        o.lnNoise = worst
    ga.multiplyBy(ga.clone())
    for o in oenc[0]:
        o.multiplyBy(o.clone())
    same(ga, oenc[0])
    assert spy.names() == ["mulRelin", "tensorBringToSet"], spy.names()
    assert len(spy.calls[0][1]["own"]) == shape[0] and spy.calls[0][1]["batch"] == B
    assert len(spy.calls[1][1]["drop"]) >= 2 and spy.calls[1][1]["batch"] == B
    # decode (raw / ratFactor) against the real product of the encoded values, within the reported bound
    for b in range(B):
        raw = osk.Decrypt(oenc[0][b])
        got = np.array([float(v) for v in raw]) / math.exp(oenc[0][b].lnRatFactor)
        x, y = (np.rint(vals[j, b] * f) / f for j in range(2))
        xy = np.convolve(x, y)
        xy = xy[:n] - np.append(xy[n:], 0.0)
        w = np.convolve(xy, xy)
        want = w[:n] - np.append(w[n:], 0.0)
        assert np.max(np.abs(got - want)) <= math.exp(oenc[0][b].lnNoise - oenc[0][b].lnRatFactor)


@pytest.mark.parametrize("m,L,t", [(16384, 5, 65537), (16384, 3, 2), (128, 4, (1 << 59) + 1), (1705, 3, 49),
                                   # whole chains of the reference's own benchmark parameter (bits=6400: 143
                                   # primes): beyond 64 source primes the digits live in private memory
                                   (128, 70, 257), (256, 150, 65537)])
def test_poly_rem_is_toPoly_then_PolyRed(hx, m, L, t):
    """hx_poly_rem = DoubleCRT::toPoly (centred CRT) + PolyRed(t, abs=true), the tail of
    SecKey::Decrypt (src/keys.cpp:1383-1405), against the oracle's big-integer toPoly."""
    P = Pair(hx, m, primes_for(m, L))
    idx = list(range(L))
    x = P.rand(idx, 31, batch=2)
    d = hx.DoubleCRT(P.g, idx, 2, x)
    got = d.toPolyMod(t)
    assert np.array_equal(d.download(), x)                       # operand unchanged
    for b in range(2):
        want = [int(v) % t for v in P.o.to_poly(idx, x[:, b])]
        assert [int(v) for v in got[b]] == want
    # small centred values survive exactly: a polynomial with coefficients in (-t/2, t/2)
    rng = np.random.default_rng(5)
    small = rng.integers(-min(t // 2, 1000), min(t // 2, 1000) + 1, size=P.N)
    rows = np.array([[int(c) % P.primes[i] for c in small] for i in idx], dtype=np.uint64)
    e = hx.DoubleCRT(P.g, idx, 1, P.o.fft(idx, rows)[:, None, :])
    assert [int(v) for v in e.toPolyMod(t)[0]] == [int(c) % t for c in small]
    with pytest.raises(hx.HxError):
        d.toPolyMod(1)


@pytest.mark.parametrize("m,p,bits", [(16384, 65537, 250), (128, 257, 150)])
def test_hoisted_automorphisms_gpu_vs_oracle(hx, m, p, bits, monkeypatch):
    """BasicAutomorphPrecon (src/matmul.cpp:48-184): digits broken once on the device,
    each rotation = hx_automorph on the digit block + hx_key_switch_digits; GPU vs the oracle
    backend from the same seed, at full level and after a multiplication (fewer primes, leading
    digits only); power-of-two m: equal to smartAutomorph bit for bit."""
    from helib_amd import ctxt as hc, keys as hk
    from oracle.backend import OracleBackend
    monkeypatch.setattr(hc.Ctxt, "measure", hx.supportsNorms(m))
    cc = hc.ChainContext(m, p, 1, bits=bits, c=3)
    P = Pair(hx, m, cc.primes)
    gsk = hk.SecKey(cc, hk.HxBackend(P.g, cc), seed=21)
    osk = hk.SecKey(cc, OracleBackend(P.o, cc), seed=21)
    ks = [3, m - 1]
    for sk in (gsk, osk):
        sk.GenSecKey(maxDegKswitch=2)
        for k in ks:
            sk.GenKeySWmatrix(1, k)
    rng = np.random.default_rng(3)
    ma, mb = rng.integers(0, p, size=cc.phim), rng.integers(0, p, size=cc.phim)

    def same(g, o):
        assert g.primeSet == o.primeSet and abs(g.lnNoise - o.lnNoise) < 1e-8
        for h in ("1", "s"):
            gi, oi = g.parts[h].getIndexSet(), o.parts[h].getIndexSet()
            gd, od = g.parts[h].download()[:, 0], o.parts[h].download()[:, 0]
            for r, i in enumerate(gi):
                assert np.array_equal(gd[r], od[oi.index(i)]), (h, i)

    ga, oa = gsk.Encrypt(ma), osk.Encrypt(ma)
    gpre, opre = hc.BasicAutomorphPrecon(ga), hc.BasicAutomorphPrecon(oa)
    for k in ks:
        gr, orr = gpre.automorph(k), opre.automorph(k)
        same(gr, orr)
        assert gsk.Decrypt(gr) == osk.Decrypt(orr)
        ref = ga.clone()
        ref.smartAutomorph(k)
        for h in ("1", "s"):
            assert np.array_equal(gr.parts[h].download(), ref.parts[h].download())
    # a rotation that needs two steps of the key-switch map (3 then 3): first step hoisted, rest plain
    for sk in (gsk, osk):
        sk.setKeySwitchMap()
    g2, o2 = gsk.Encrypt(ma), osk.Encrypt(ma)
    gr, orr = hc.BasicAutomorphPrecon(g2).automorph(9), hc.BasicAutomorphPrecon(o2).automorph(9)
    same(gr, orr)
    assert gsk.Decrypt(gr) == osk.Decrypt(orr)
    ga.multiplyBy(gsk.Encrypt(mb))
    oa.multiplyBy(osk.Encrypt(mb))
    gpre, opre = hc.BasicAutomorphPrecon(ga), hc.BasicAutomorphPrecon(oa)
    assert gpre.ctxt.primeSet == opre.ctxt.primeSet <= frozenset(cc.ctxtPrimes)
    gr, orr = gpre.automorph(3), opre.automorph(3)
    same(gr, orr)
    assert gsk.Decrypt(gr) == osk.Decrypt(orr)


def test_dropped_results_do_not_leave_dangling_norm_buffers(hx):
    """Deferred norms are written into host arrays at the next flush; a ciphertext dropped without
    reading its noise estimate must not take its array with it (regression: heap corruption at
    interpreter exit after tools/bench_keys.py dropped 40 rotated ciphertexts)."""
    import gc
    from helib_amd import ctxt as hc, keys as hk
    m, p = 16384, 65537
    cc = hc.ChainContext(m, p, 1, bits=250, c=3)
    P = Pair(hx, m, cc.primes)
    sk = hk.SecKey(cc, hk.HxBackend(P.g, cc), seed=2)
    sk.GenSecKey(maxDegKswitch=2)
    sk.GenKeySWmatrix(1, 3)
    rng = np.random.default_rng(0)
    ma = rng.integers(0, p, size=cc.phim)
    ca = sk.Encrypt(ma)
    want = sk.Decrypt(ca.clone().smartAutomorph(3))
    for _ in range(30):
        c = ca.clone()
        c.smartAutomorph(3)          # leaves deferred norms behind
        del c
    gc.collect()
    junk = [np.ones(64) for _ in range(2000)]      # reuse the freed heap blocks
    assert len(getattr(P.g, "_deferred", [])) > 0
    P.g.flushNorms()
    assert P.g._deferred == [] and all(float(j.sum()) == 64.0 for j in junk)
    assert sk.Decrypt(ca.clone().smartAutomorph(3)) == want


@pytest.mark.parametrize("m,p,bits,k,measure", [(16384, 65537, 250, 3, 0), (16384, 65537, 250, 5, 1), (128, 257, 150, 3, 0)])
def test_cpp_host_ctxt_matches_python_mirror(hx, m, p, bits, k, measure, tmp_path, monkeypatch):
    """include/helib_amd_ctxt.hpp -- the C++ host side (ChainContext, ModuliSizes, Ctxt::multiplyBy /
    addCtxt / smartAutomorph with the reference's bookkeeping) over the C ABI -- against the python
    mirror on the same keys and ciphertexts: prime sets, intFactor, noise estimate and every word
    of every part, for the product, product + product and the rotated product; and decrypt."""
    import struct
    import subprocess
    from helib_amd import ctxt as hc
    from tests import test_ctxt_host as T
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ctxt_test")
    libdir = os.path.join(root, "helib_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "ctxt_test.cpp"), "-L" + libdir, "-lhelib_amd",
                           "-Wl,-rpath," + libdir, "-o", exe])
    monkeypatch.setattr(hc.Ctxt, "measure", bool(measure))
    cc = hc.ChainContext(m, p, 1, bits=bits, c=3)
    P = Pair(hx, m, cc.primes)
    s, allp, kb, ka, rows = T.make_keys(cc, P.o)
    _, _, kbk, kak, _ = T.make_keys(cc, P.o, auto_k=k)
    rng = np.random.default_rng(4)
    ma, mb = rng.integers(0, p, size=P.N), rng.integers(0, p, size=P.N)
    ea, eb = T.encrypt(cc, P.o, s, ma, 1, rows), T.encrypt(cc, P.o, s, mb, 2, rows)
    L, D, N = len(cc.ctxtPrimes), len(cc.digits), P.N
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<10q", m, p, bits, k, measure, len(cc.primes), D, len(allp), L, N))
        f.write(np.array(P.o.roots, dtype="<u8").tobytes())
        for arr in (kb, ka, kbk, kak, ea[0], ea[1], eb[0], eb[1]):
            f.write(np.ascontiguousarray(arr, dtype="<u8").tobytes())
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # the python mirror over the same device backend
    gW, gWk = hx.KeySwitch(P.g, allp, kb, ka), hx.KeySwitch(P.g, allp, kbk, kak)
    mk = lambda e: hc.Ctxt.fresh(cc, hx, *(hx.DoubleCRT(P.g, cc.ctxtPrimes, 1, x[:, None, :]) for x in e), ksw=gW)  # noqa: E731
    ga, gb = mk(ea), mk(eb)
    ga.ksw_auto = {k: gWk}
    ga.multiplyBy(gb)
    prod = ga.clone()
    gsum = ga.clone()
    gsum.addCtxt(ga)
    ga.smartAutomorph(k)
    rot1 = ga.clone()
    kmap = [0] * m                   # PubKey::setKeySwitchMap with the single edge k
    cur = k
    while cur != 1 and kmap[cur] == 0:
        kmap[cur] = k
        cur = cur * k % m
    ga.ksw_map = kmap
    ga.smartAutomorph(k * k % m)
    buf = open(fout, "rb").read()
    off = 0
    hname = {(0, 1): "1", (1, 1): "s"}
    for want in (prod, gsum, rot1, ga):
        nset, intFactor, nparts = struct.unpack_from("<3q", buf, off)
        (ln,) = struct.unpack_from("<d", buf, off + 24)
        off += 32
        pset = list(struct.unpack_from(f"<{nset}q", buf, off))
        off += 8 * nset
        assert pset == sorted(want.primeSet) and intFactor == want.intFactor and nparts == len(want.parts)
        assert abs(ln - want.lnNoise) < 1e-8, (ln, want.lnNoise)
        for _ in range(nparts):
            sp, xp, nr = struct.unpack_from("<3q", buf, off)
            idx = list(struct.unpack_from(f"<{nr}q", buf, off + 24))
            off += 24 + 8 * nr
            got = np.frombuffer(buf, dtype="<u8", count=nr * N, offset=off).reshape(nr, N)
            off += 8 * nr * N
            part = want.parts[hname[(sp, xp)]]
            wi, wd = part.getIndexSet(), part.download()[:, 0]
            assert sorted(idx) == sorted(wi)
            for rr, i in enumerate(idx):
                assert np.array_equal(got[rr], wd[wi.index(i)]), (sp, xp, i)
    (errs,) = struct.unpack_from("<q", buf, off)
    assert errs == 3 and off + 8 == len(buf)          # LogicError and InvalidArgument both raised
    if m < 4096:
        from tests import bgv_ref as B
        ab = [int(v) for v in B.polymul_mod_phi(ma, mb, m, p)]
        assert T.decrypt(cc, P.o, s, prod, rows) == ab
        assert T.decrypt(cc, P.o, s, rot1, rows) == [int(v) for v in B.automorph_mod_phi(ab, m, k, p)]
        assert T.decrypt(cc, P.o, s, ga, rows) == [int(v) for v in B.automorph_mod_phi(ab, m, pow(k, 3, m), p)]


def test_threads_on_distinct_polys_of_one_context(hx):
    """HElib's NTL thread pool calls DoubleCRT operations concurrently on distinct objects of one
    Context (re-entrancy, src/CModulus.cpp:580-610).  Eight threads drive their own polys through
    transforms, ring operations and the exact-RNS operations (shared plan cache, scratch slots and
    slab pool inside the context); every result must equal the single-threaded oracle."""
    from concurrent.futures import ThreadPoolExecutor
    m, L = 16384, 5
    P = Pair(hx, m, primes_for(m, L + 2))
    own, extra = list(range(L)), [L, L + 1]

    def work(t):
        x, y = P.rand(own, 100 + t), P.rand(own, 200 + t)
        a, b = hx.DoubleCRT(P.g, own, 1, x), hx.DoubleCRT(P.g, own, 1, y)
        out = {}
        for _ in range(3):                                  # repeat: more interleavings
            a2 = a.copy()
            a2 *= b
            a2.automorph(3 + 2 * t)
            out["mul"] = a2.download()[:, 0]
            c = a.copy()
            c.addPrimes(extra)
            out["addPrimes"] = c.download()[:, 0]
            d = a.copy()
            d.scaleDownToSet(own[:-2], 65537)                # two dropped primes: the generic path
            out["scaleDown"] = d.download()[:, 0]
            e = a.copy()
            e.scaleDownToSet(own[:-1], 65537)                # one dropped prime: the fused path
            out["scaleDown1"] = e.download()[:, 0]
        return t, x[:, 0], y[:, 0], out

    with ThreadPoolExecutor(max_workers=8) as ex:
        results = list(ex.map(work, range(8)))
    zms = O.zmstar(m)
    for t, x, y, out in results:
        prod = np.stack([O.row_op("mul", x[r], y[r], P.primes[i]) for r, i in enumerate(own)])
        want = np.stack([O.automorph(r, m, zms, 3 + 2 * t) for r in prod])
        assert np.array_equal(out["mul"], want), t
        assert np.array_equal(out["addPrimes"], np.vstack([x, P.o.add_primes(own, x, extra)])), t
        assert np.array_equal(out["scaleDown"], P.o.scale_down(own, x, own[-2:], 65537)), t
        assert np.array_equal(out["scaleDown1"], P.o.scale_down(own, x, own[-1:], 65537)), t


# ---------------------------------------------------------------- the reference's own fixture, on the device
def test_reference_fixture_on_the_device(hx):
    """The only value-level vectors the reference ships (tests/test_resources/iotest_asciiLE.txt, m = 12,
    five primes written by an older HElib; tests/golden/iotest_m12.json) through the HIP path itself:
    the secret-key rows inverse-transform to s = 1 - X + X^2 + X^3 under every prime and transform
    back to the fixture rows bit for bit; the public key satisfies b + a*s = 7*e with the short e of
    the fixture; and hx_poly_rem recovers e's sign pattern.  These primes have only 2^3..2^15 | q-1,
    so the Bluestein convolutions run through the auxiliary-prime path (crt3_kernel)."""
    import json
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "iotest_m12.json")))
    m, p, primes = gold["m"], gold["p"], gold["primes"]
    g = hx.Context(m)
    for q in primes:
        g.add_prime(q)                                    # FindPrimRootT(q, 2m), the fixture's convention
    idx5, idx3 = list(range(5)), list(range(3))
    srows = np.array(gold["seckey"]["rows"], dtype=np.uint64)
    s = hx.DoubleCRT(g, idx5, 1, srows[:, None, :])
    coef = s.copy().iFFT().download()[:, 0]
    for q, row in zip(primes, coef):
        assert [int(x) - q if int(x) > q // 2 else int(x) for x in row] == gold["expect_s_coeffs"]
    assert np.array_equal(s.copy().iFFT().FFT().download()[:, 0], srows)
    b = hx.DoubleCRT(g, idx3, 1, np.array(gold["pubkey_b"]["rows"], dtype=np.uint64)[:, None, :])
    a = hx.DoubleCRT(g, idx3, 1, np.array(gold["pubkey_a"]["rows"], dtype=np.uint64)[:, None, :])
    a *= s                                                # s lives on more primes: Mul(matchIndexSets=false)
    b += a
    pe = b.copy().iFFT().download()[:, 0]
    for q, row in zip(primes, pe):
        assert [int(x) - q if int(x) > q // 2 else int(x) for x in row] == [p * c for c in gold["expect_e_coeffs"]]
    # the decryption tail on the device: (b + a s) centred mod Q, reduced mod p = 7 -> 0; mod 1000003 -> 7 e
    assert [int(v) for v in b.toPolyMod(p)[0]] == [0, 0, 0, 0]
    t = 1000003
    assert [int(v) for v in b.toPolyMod(t)[0]] == [(p * c) % t for c in gold["expect_e_coeffs"]]
    # the oracle agrees row for row on a random polynomial under these primes (forward and inverse)
    o = O.Ctx(m)
    for q in primes:
        o.add_prime(q)
    x = np.stack([O.fill_uniform(4, q, 3 + i) for i, q in enumerate(primes)])
    d = hx.DoubleCRT(g, idx5, 1, x[:, None, :])
    assert np.array_equal(d.FFT().download()[:, 0], o.fft(idx5, x))
    assert np.array_equal(d.iFFT().download()[:, 0], x)


@pytest.mark.parametrize("m", [105, 1705, 12])
def test_bluestein_auxiliary_prime_path_matches_oracle(hx, m):
    """General m with primes that lack the 2-power roots (q = 2 k e + 1 with k odd-ish): the chirp and
    rem-Phi_m convolutions go through three auxiliary NTT primes; mixed with a PrimeGenerator prime
    in the same DoubleCRT."""
    e = 2 * m if m % 2 == 0 else m
    primes, k = [], (1 << 40) // e
    while len(primes) < 2:
        k += 1
        q = k * e + 1
        v2 = (q - 1) & -(q - 1)
        if v2 <= 8 and all(q % s for s in (3, 5, 7, 11, 13)) and pow(2, q - 1, q) == 1 and pow(3, q - 1, q) == 1:
            from helib_amd import hostnt
            if hostnt.is_prime(q):
                primes.append(q)
    primes.append(O.PrimeGen(50, m).next())               # a normal one beside them
    P = Pair(hx, m, primes)
    idx = [0, 1, 2]
    x = P.rand(idx, 5, batch=2)
    d = hx.DoubleCRT(P.g, idx, 2, x)
    y = d.FFT().download()
    for b in range(2):
        assert np.array_equal(y[:, b], P.o.fft(idx, x[:, b]))
    assert np.array_equal(d.iFFT().download(), x)


@pytest.mark.parametrize("m", [65539, 131071, 262139])
def test_bluestein_conv_2_18_radix8_split(hx, m):
    """General m with 2m-1 > 2^17: the chirp convolution has 2^18 points = eight 2^15-point
    sub-transforms behind a radix-8 split (conv_core.h split_fwdN<3>); m = 131071 is the largest
    prime m with 2m-1 <= 2^18.  Beyond it (m = 262139, prime) the convolution
    has 2^19 points = sixteen sub-transforms behind the radix-16 split (split_fwdN<4>).  Forward and
    inverse vs the oracle, batch 2."""
    g = O.PrimeGen(56, m)
    P = Pair(hx, m, [g.next(), g.next()])
    idx = [0, 1]
    x = P.rand(idx, 9, batch=2)
    d = hx.DoubleCRT(P.g, idx, 2, x)
    y = d.FFT().download()
    for b in range(2):
        assert np.array_equal(y[:, b], P.o.fft(idx, x[:, b]))
    assert np.array_equal(d.iFFT().download(), x)
    if m == 131071:
        with pytest.raises(hx.HxError):                  # beyond the radix-16 split: conv size 2^20
            hx.Context(262145).add_prime(O.PrimeGen(56, 262145).next())


def test_wrapped_poly_multi_prime_scale_down_stays_in_caller_memory(hx):
    """hx_poly_wrap + the generic (several dropped primes) scaleDownToSet: a library-owned poly
    swaps to a fresh compact slab, a wrapped one must get the same rows written back into the
    caller's buffer."""
    import torch
    m = 16384
    primes = primes_for(m, 7)
    P = Pair(hx, m, primes)
    idx = list(range(7))
    x = P.rand(idx, 71, batch=3)
    own = hx.DoubleCRT(P.g, idx, 3, x)
    buf = torch.from_numpy(x.view(np.int64).copy()).to("cuda:0")
    w = hx.DoubleCRT.wrap(P.g, idx, 3, buf.data_ptr())
    assert np.array_equal(w.download(), x)
    keep = [0, 2, 3, 6]                                    # drops 1, 4, 5: compaction would move rows 2, 3, 6
    own.scaleDownToSet(keep, 65537)
    w.scaleDownToSet(keep, 65537)
    assert w.getIndexSet() == own.getIndexSet() == keep
    want = own.download()
    for b in range(3):
        assert np.array_equal(want[:, b], P.o.scale_down(idx, x[:, b], [1, 4, 5], 65537))
    assert np.array_equal(w.download(), want)
    torch.cuda.synchronize()
    back = buf.cpu().numpy().view(np.uint64).reshape(7, 3, P.N)
    assert np.array_equal(back[:4], want)                 # the result lives in the caller's tensor
    w.close()


@pytest.mark.parametrize("m,p,bits,measure", [(128, 257, 150, 0), (16384, 65537, 250, 1), (16384, 2, 250, 0),
                                              (128, -1, 250, 0), (8192, -1, 300, 1)])
def test_cpp_host_keys_encrypt_multiply_rotate_decrypt(hx, m, p, bits, measure, tmp_path):
    """include/helib_amd_keys.hpp -- SecKey::GenSecKey / GenKeySWmatrix / PubKey::Encrypt /
    SecKey::Decrypt in C++ over the C ABI (src/keys.cpp:39-85, 358-488, 1099-1255, 1327-1420) --
    with the C++ Ctxt: decrypt(encrypt(m)) = m, decrypt(a*b) = a*b mod (X^N+1, p) also before
    relinearisation, a*b + a*b, and rotations in one and two key-switch-map steps
    (tests/cpp/keys_test.cpp checks against schoolbook arithmetic).  p = -1: the CKKS chain --
    CKKSencrypt, products over two levels, sums/differences across scaling factors
    (equalizeRationalFactors), DecryptCKKS; every decoded error below the ciphertext's own bound."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "keys_test")
    libdir = os.path.join(root, "helib_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "keys_test.cpp"), "-L" + libdir, "-lhelib_amd",
                           "-Wl,-rpath," + libdir, "-o", exe])
    r = subprocess.run([exe, str(m), str(p), str(bits), str(measure)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "keys_test OK" in r.stdout, r.stdout + r.stderr


_CTXT_OPS_EXE = {}


@pytest.mark.parametrize("m,p,bits,measure", [(128, 257, 300, 0), (4096, 65537, 500, 1), (2048, 3, 400, 1),
                                              (128, -1, 400, 0), (2048, -1, 500, 1), (105, 257, 300, 0),
                                              (1705, 2, 300, 1)])
def test_cpp_host_ctxt_operations(hx, m, p, bits, measure, tmp_path_factory):
    """include/helib_amd_ctxt.hpp beyond multiplyBy, from C++ over the C ABI (tests/cpp/ctxt_ops_test.cpp
    checks every result against schoolbook arithmetic after decryption): multiplyBy2 / cube / power through
    the 4-part ciphertext and keySwitchPart (hx_break_into_digits[_norms] + hx_key_switch_digits per part,
    src/Ctxt.cpp:1776-1828, 720-842), totalProduct / incrementalProduct / innerProduct (:2803-2904),
    BasicAutomorphPrecon (hoisting: one digit block, hx_automorph on it, one key switch per rotation,
    src/matmul.cpp:48-184), frobeniusAutomorph, multByConstant / addConstant, capacity / isCorrect; p = -1:
    the CKKS forms incl. complex conjugation; m = 105 and 1705: the Bluestein rows, keys from the general-m
    samplers (reduction modulo Phi_m).  The same program runs over the CPU mock of the C ABI in
    tests/test_cpp_host_cpu.py."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if "exe" not in _CTXT_OPS_EXE:       # one build for all parameter sets
        exe = str(tmp_path_factory.mktemp("ctxt_ops") / "ctxt_ops_test")
        libdir = os.path.join(root, "helib_amd", "lib")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"),
                               os.path.join(root, "tests", "cpp", "ctxt_ops_test.cpp"), "-L" + libdir, "-lhelib_amd",
                               "-Wl,-rpath," + libdir, "-o", exe])
        _CTXT_OPS_EXE["exe"] = exe
    r = subprocess.run([_CTXT_OPS_EXE["exe"], str(m), str(p), str(bits), str(measure)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "ctxt_ops_test OK" in r.stdout, r.stdout + r.stderr


def test_captured_graph_replays_a_multiply(hx):
    """hx_ctx_graph_begin / _end / hx_graph_launch: tensorProduct + reLinearize (hx_mul_relin) recorded
    once and replayed with one launch.  A replay re-executes the same kernels on the same buffers, so
    (a) nothing runs while recording, (b) new data uploaded into the SAME input polys gives the new
    product in the SAME output polys, bit-exact against the oracle, (c) slabs released while a graph
    is alive are not handed out again (the graph may point at them), and come back to the pool when it
    is destroyed."""
    m, L, K = 16384, 3, 1
    digits = [[0, 1], [2]]
    g60 = O.PrimeGen(60, m)
    primes = [g60.next() for _ in range(L + K)]
    P = Pair(hx, m, primes)
    own, sp = list(range(L)), list(range(L, L + K))
    allp = own + sp
    batch = 2
    ins = [P.rand(own, s, batch) for s in (1, 2, 3, 4)]
    kb = np.stack([P.rand(allp, 20 + i)[:, 0] for i in range(len(digits))])
    ka = np.stack([P.rand(allp, 30 + i)[:, 0] for i in range(len(digits))])
    W = hx.KeySwitch(P.g, allp, kb, ka)
    G = [hx.DoubleCRT(P.g, own, batch, x) for x in ins]

    def want(data):
        return [P.o.mul_relin(own, sp, digits, *(x[:, b] for x in data), kb, ka) for b in range(batch)]

    def check(o0, o1, data):
        g0, g1 = o0.download(), o1.download()
        for b, (w0, w1) in enumerate(want(data)):
            assert np.array_equal(g0[:, b], w0) and np.array_equal(g1[:, b], w1)

    e0, e1 = hx.multiplyBy(*G, W, digits)              # eagerly once: plans, tables, kernel attributes
    check(e0, e1, ins)
    P.g.graphBegin()
    o0, o1 = hx.multiplyBy(*G, W, digits)
    graph = P.g.graphEnd()
    zeros = np.zeros_like(o0.download())
    o0.upload(zeros)                                    # (whatever the slabs held: nothing ran yet)
    o1.upload(zeros)
    graph.launch()
    check(o0, o1, ins)
    # new operands in the same polys; garbage allocated and dropped in between (would reuse a slab the
    # graph's temporaries live in if the context handed those out)
    new = [P.rand(own, 50 + s, batch) for s in range(4)]
    for d, x in zip(G, new):
        d.upload(x)
    junk = [hx.DoubleCRT(P.g, allp, batch, P.rand(allp, 70 + i, batch)) for i in range(6)]
    del junk
    o0.upload(zeros)
    graph.launch()
    check(o0, o1, new)
    graph.launch().launch()                             # idempotent on unchanged inputs
    check(o0, o1, new)
    # not capturable: a call that has to wait for the device fails (and leaves the recording open)
    P.g.graphBegin()
    with pytest.raises(hx.HxError) as ei:
        o0.download()
    assert "cannot be captured" in str(ei.value)
    with pytest.raises(hx.HxError):
        P.g.graphBegin()                                # one recording at a time
    P.g.graphEnd().destroy()
    graph.destroy()
    f0, f1 = hx.multiplyBy(*G, W, digits)              # the context works as before
    check(f0, f1, new)


def test_captured_graph_inputs_that_had_lazy_copies(hx):
    """hx_poly_copy is copy-on-write, a graph replays on recorded addresses.  An input
    poly that shares its slab with a clone must not move off that slab when it is written after the
    capture (upload / set_zero / element-wise): hx_ctx_graph_begin gives every shared poly its own slab,
    and copies made while a recording is open or a graph is alive are eager.  Also: eager work next to
    a live graph recycles its slabs (no growth), and a raw device pointer taken from a poly keeps later
    copies of it independent."""
    m, L, K = 16384, 3, 1
    digits = [[0, 1], [2]]
    g60 = O.PrimeGen(60, m)
    primes = [g60.next() for _ in range(L + K)]
    P = Pair(hx, m, primes)
    own, sp = list(range(L)), list(range(L, L + K))
    allp = own + sp
    ins = [P.rand(own, s, 1) for s in (1, 2, 3, 4)]
    kb = np.stack([P.rand(allp, 20 + i)[:, 0] for i in range(len(digits))])
    ka = np.stack([P.rand(allp, 30 + i)[:, 0] for i in range(len(digits))])
    W = hx.KeySwitch(P.g, allp, kb, ka)
    G = [hx.DoubleCRT(P.g, own, 1, x) for x in ins]
    clones = [d.copy() for d in G]                       # lazy: every input now shares its slab with a clone
    hx.multiplyBy(*G, W, digits)                         # eagerly once
    P.g.graphBegin()
    during = G[0].copy()                                 # recorded as a real copy
    o0, o1 = hx.multiplyBy(*G, W, digits)
    graph = P.g.graphEnd()
    new = [P.rand(own, 50 + s, 1) for s in range(4)]
    for d, x in zip(G, new):
        d.upload(x)                                      # must write the slab the graph reads
    graph.launch()
    w0, w1 = P.o.mul_relin(own, sp, digits, *(x[:, 0] for x in new), kb, ka)
    assert np.array_equal(o0.download()[:, 0], w0) and np.array_equal(o1.download()[:, 0], w1)
    assert np.array_equal(during.download(), new[0])     # the recorded copy replays too
    for cl, x in zip(clones, ins):
        assert np.array_equal(cl.download(), x)          # the clones kept the old data
    # eager work while the graph is alive: its temporaries recycle instead of piling up
    reserved = []
    for rep in range(12):
        tmp = [hx.DoubleCRT(P.g, allp, 1, P.rand(allp, 90 + i, 1)) for i in range(4)]
        e0, e1 = hx.multiplyBy(*G, W, digits)
        del tmp, e0, e1
        reserved.append(P.g.arenaStats()["reserved"])
    assert reserved[-1] == reserved[3], reserved
    graph.launch()
    assert np.array_equal(o0.download()[:, 0], w0)
    graph.destroy()
    assert P.g.arenaStats()["deferred"] == 0
    # a raw pointer makes later copies eager
    a = hx.DoubleCRT(P.g, own, 1, ins[0])
    ptr = a.device_ptr()
    b = a.copy()
    assert b.device_ptr() != ptr
    a.upload(new[1])
    assert np.array_equal(b.download(), ins[0])


def test_captured_graph_of_the_fresh_multiplyBy_sequence(hx, monkeypatch):
    """The whole fresh Ctxt::multiplyBy (copy, both bringToSet mod-switches, tensor product,
    dropSmallAndSpecialPrimes, key switch; the reference's noise bounds -- measured noise needs
    read-backs) recorded through the python mirror and replayed: the replayed result equals the
    eager one word for word -- the loop body of benchmarks/bgv_basic.cpp:158-164 as one launch."""
    from helib_amd import ctxt as hc
    from tests import test_ctxt_host as T
    m, p = 16384, 65537
    monkeypatch.setattr(hc.Ctxt, "measure", False)
    ctx = hc.ChainContext(m, p, 1, bits=250, c=3)
    P = Pair(hx, m, ctx.primes)
    s, allp, kb, ka, rows = T.make_keys(ctx, P.o)
    rng = np.random.default_rng(8)
    ma, mb = rng.integers(0, p, size=P.N), rng.integers(0, p, size=P.N)
    ea, eb = T.encrypt(ctx, P.o, s, ma, 1, rows), T.encrypt(ctx, P.o, s, mb, 2, rows)
    gW = hx.KeySwitch(P.g, allp, kb, ka)
    fa = hc.Ctxt.fresh(ctx, hx, *(hx.DoubleCRT(P.g, ctx.ctxtPrimes, 1, x[:, None, :]) for x in ea), ksw=gW)
    fb = hc.Ctxt.fresh(ctx, hx, *(hx.DoubleCRT(P.g, ctx.ctxtPrimes, 1, x[:, None, :]) for x in eb), ksw=gW)
    eager = fa.clone()
    eager.multiplyBy(fb)
    want = {h: (q.getIndexSet(), q.download()) for h, q in eager.parts.items()}
    P.g.graphBegin()
    rec = fa.clone()
    rec.multiplyBy(fb)
    graph = P.g.graphEnd()
    for _ in range(3):
        graph.launch()
    assert rec.primeSet == eager.primeSet and set(rec.parts) == set(want)
    for h, (idx, data) in want.items():
        assert rec.parts[h].getIndexSet() == idx and np.array_equal(rec.parts[h].download(), data)
    assert T.decrypt(ctx, P.o, s, rec, rows) == T.decrypt(ctx, P.o, s, eager, rows)
    graph.destroy()


_FACADE2_EXE = {}


@pytest.mark.parametrize("m", [128, 32768, 1705, 21845])
def test_cpp_cmodulus_bignum_toPoly_and_namespace_intel(hx, m, tmp_path_factory):
    """The facade pieces a HElib caller of this path also uses, from C++ on the device (tests/cpp/facade2_test.cpp,
    self-checking): Cmodulus{FFT, iFFT} (include/helib/CModulus.h:137-145) against the O(N^2) definition,
    DoubleCRT::toPoly with big-integer coefficients (src/DoubleCRT.cpp:925-1113), and `namespace intel`
    (include/helib_amd_intel.hpp: the eight signatures of src/intelExt.h:20-59) through call sites shaped like
    the reference's USE_INTEL_HEXL ones.  m = 32768: the benchmark ring; 1705, 21845: Bluestein rows."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if "exe" not in _FACADE2_EXE:
        exe = str(tmp_path_factory.mktemp("facade2") / "facade2_test")
        libdir = os.path.join(root, "helib_amd", "lib")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(root, "include"),
                               os.path.join(root, "tests", "cpp", "facade2_test.cpp"), "-L" + libdir, "-lhelib_amd",
                               "-Wl,-rpath," + libdir, "-o", exe])
        _FACADE2_EXE["exe"] = exe
    r = subprocess.run([_FACADE2_EXE["exe"], str(m)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "facade2_test OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("scheme,m,p,r,bits,batch", [("bgv", 16384, 65537, 1, 300, 3), ("ckks", 16384, -1, 20, 500, 2)])
def test_cpp_host_session_library_on_the_device(hx, scheme, m, p, r, bits, batch):
    """helib_amd/lib/libhelib_amd_host.so (the C++17 host bench.py times, include/helib_amd_host.h) on the device:
    keys and batched encryptions in C++, the benchmark loop at level 1 and level 2 with measured noise read back
    lazily, every batch element of both products decrypted in C++ and compared with the plaintext product."""
    from helib_amd import host
    s = host.Session(scheme, m, p, r, bits, batch, seed=5)
    assert s.verify(0) == batch
    s.multiply(1, 3, True)
    assert s.verify(1) == batch
    s.multiply(2, 2, True)
    assert s.verify(2) == batch
    s.multiply(1, 2, False)                       # noise bounds instead of measured noise
    assert s.verify(1) == batch
    s.multiply_single(True)
    s.close()


@pytest.mark.parametrize("scheme,m,p,r,bits,batch,shape", [
    ("bgv", 32768, 65537, 1, 950, 4, (16, 6, 3)),          # bench.py's default line (BASELINE configs[2])
    ("ckks", 65536, -1, 1, 1400, 2, (24, 8, 3)),           # bench.py --workload ckks65536 (BASELINE configs[3])
    ("ckks", 65536, -1, 1, 440, 2, (8, 3, 3)),             # the reference's own benchmarks/ckks_basic.cpp:263 parameters
    ("bgv", 16384, 65537, 1, 300, 3, None),
    # the reference's general-m benchmark parameters (benchmarks/bgv_basic.cpp:236 big_params: m = 32003 prime, p = 2,
    # bits = 5800): Bluestein transforms at N = 32002 under 97 + 33 primes, digits of 33 / 32 / 32 primes through
    # rns_extend_wide_kernel; level 1 only (the oracle needs over a minute per multiply here)
    ("bgv", 32003, 2, 1, 5800, 1, (97, 33, 3))])
def test_cpp_session_products_equal_the_oracle_replay(hx, scheme, m, p, r, bits, batch, shape):
    """The TIMED path of bench.py against the oracle, word for word: libhelib_amd_host.so's session (C++ Ctxt /
    SecKey; multiplyBy leaving the tensor product to hx_tensor_bring_to_set_norms / hx_mul_relin_norms, i.e.
    ntt_moddown_prep_tensor_kernel, ntt_moddown_apply_tensor_kernel<14|15, false|true>, ntt_inv_mul_kernel and the
    key-switch kernel with a TensorSrc) at the benchmark parameters.  tests/session_replay.py downloads the session's
    own operands, relinearisation matrix and bookkeeping, drives the python mirror over the oracle backend (the
    reference's unfused sequence, src/Ctxt.cpp:1563-1608, :346-562, :720-786) per batch element, and compares every
    word of the kept products of level 1 and level 2.  Then a second session built from the first one's exported key
    material (one key pair replicated) multiplies its own encryptions and decrypts under that key."""
    from helib_amd import host
    from tests.session_replay import replay_and_compare
    s = host.Session(scheme, m, p, r, bits, batch, seed=31)
    if shape:
        assert (s.L_ctxt, s.K, s.D) == shape
    keys = s.export_keys()
    levels = (1,) if m == 32003 else (1, 2)
    words = replay_and_compare(s, scheme, m, p, r, bits, measure=True, levels=levels)
    assert words == 2 * batch * s.phim * sum(len(s.result_primes(lv)) for lv in levels)   # both parts, every level run
    assert s.verify(levels[-1]) == batch
    if m == 32003:
        s.close()
        return
    t = host.Session(scheme, m, p, r, bits, batch, seed=32, keys=keys)
    t.multiply(1, 1, True)
    assert t.verify(1) == batch
    s.close()
    t.close()


@pytest.mark.parametrize("scheme,m,p,r,bits,batch,elements,shape", [
    ("bgv", 32768, 65537, 1, 950, 128, (0, 1, 63, 64, 127), (16, 6, 3)),   # bench.py's default line: batch 128
    ("ckks", 65536, -1, 1, 1400, 64, (0, 31, 32, 63), (24, 8, 3))])        # bench.py --workload ckks65536: batch 64
def test_cpp_session_products_at_the_timed_launch_shapes(hx, scheme, m, p, r, bits, batch, elements, shape):
    """Word-for-word parity AT THE LAUNCH SHAPES THE BENCHMARK LINE TIMES.  The work-to-workgroup
    maps of the hot kernels are functions of the launch size -- md_tile(nkeep, npoly * batch) picks the row group, the
    chunk and the per-XCD split (ntt_kernels.hip), xcd_remap(blockIdx.x, gridDim.x) the XCD slot (ntt_kernel_util.h) --
    so the 8192- / 6144- / 6400- / 16384-workgroup launches of the line are different index arithmetic from the
    256-workgroup launches of the batch-4 / batch-2 cases above; decryption (Session.verify) would catch a dropped row,
    not a wrong word that still decrypts.  Here the session runs at bench.py's own batch (128 for BGV bits = 950, 64
    for CKKS bits = 1400), and a stratified sample of its elements -- first, second, the two around the middle, last --
    is replayed on the oracle through the python mirror (src/Ctxt.cpp:1681-1774): every word of both parts of the
    kept products of level 1 AND level 2.  Every element is also decrypted and compared with the plaintext product."""
    from helib_amd import host
    from tests.session_replay import replay_and_compare
    s = host.Session(scheme, m, p, r, bits, batch, seed=41)
    assert (s.L_ctxt, s.K, s.D) == shape
    words = replay_and_compare(s, scheme, m, p, r, bits, measure=True, levels=(1, 2), elements=elements)
    assert words == 2 * len(elements) * s.phim * sum(len(s.result_primes(lv)) for lv in (1, 2))
    assert s.verify(2) == batch
    s.close()


@pytest.mark.parametrize("eps", ["default", "0.05", "1.0"])
def test_hps_form_of_the_rns_kernels_and_its_redo_list(hx, monkeypatch, eps):
    """The fast basis-extension / digit kernels in their HPS form (rns_kernels.h: ExtRep, engine.hip: hps_min_n) --
    y_k = a_k (P/p_k)^-1, quotient from a double-precision sum, coefficients whose sum is too close to a boundary
    appended to a redo list that a Garner launch then recomputes.  HX_HPS_MIN_N = 2 sends every plan through it
    (default: nine source primes and more); eps = 2^-30 (default: the list is almost always empty), 0.05 (about
    a third of the coefficients redone next to trusted ones) and 1.0 (every coefficient redone).  Same checks as
    the plain tests: every output word against the oracle, fdelta and norms within their tolerances."""
    monkeypatch.setenv("HX_HPS_MIN_N", "2")
    if eps != "default":
        monkeypatch.setenv("HX_HPS_EPS", eps)
    test_scale_down_to_set(hx, 65537)
    test_scale_down_to_set(hx, 1)
    test_break_into_digits(hx, [[0, 1], [2, 3], [4]])
    test_break_into_digits(hx, [[0, 1, 2, 3, 4]])
    test_scale_down_norms_and_fdelta(hx, 2, 65537)
    test_scale_down_norms_and_fdelta(hx, 2, 1)
    test_break_into_digits_and_relinearize_norms(hx, [[0, 1], [2, 3], [4]])
    test_several_primes_mod_switch_batched_over_parts_mixed_prime_sizes(hx, 16384, 65537)
    test_tensor_folded_into_the_mod_switch(hx, 16384, 65537, "drop3")


@pytest.mark.parametrize("no_proth_rns", [False, True])
def test_proth_form_of_the_fast_rns_kernels(hx, monkeypatch, no_proth_rns):
    """The fast digit / basis-extension kernels on Proth-form primes (rns_kernels.h: ExtPlanDev::src_mont,
    TgtRec::mont; engine.hip: rec_to_mont): Garner steps, the targets' limb sums and the later rows' fix-ups as
    Montgomery products / mont_redc128 -- the choice is per source set and per TARGET.  First the plain cases
    (every prime of the form), then a chain that mixes a 38-bit prime that is NOT of the form (q != 1 mod 2^32) in:
    as a source next to 60-bit ones (that digit keeps the Shoup Garner steps, its targets are of both kinds) and as
    a target of all-Proth digits (red128_any next to mont_redc128 in one launch); also in the HPS form
    (HX_HPS_MIN_N = 2) and through addPrimes / scaleDownToSet.  Every word against the oracle, and the same under
    HX_NO_PROTH_RNS = 1 (the Barrett / Shoup forms everywhere)."""
    if no_proth_rns:
        monkeypatch.setenv("HX_NO_PROTH_RNS", "1")
    else:
        monkeypatch.delenv("HX_NO_PROTH_RNS", raising=False)
    test_break_into_digits(hx, [[0, 1], [2, 3], [4]])
    test_break_into_digits(hx, [[0, 1, 2, 3, 4]])
    test_scale_down_to_set(hx, 65537)
    test_break_into_digits_and_relinearize_norms(hx, [[0, 1], [2, 3], [4]])
    m = 16384
    q38 = O.PrimeGen(38, m).next()
    assert q38 & 0xffffffff != 1 and q38 >> 32
    g60, g56 = O.PrimeGen(60, m), O.PrimeGen(56, m)
    primes = [q38] + [g60.next() for _ in range(5)] + [g56.next() for _ in range(2)]
    for hps in (False, True):
        if hps:
            monkeypatch.setenv("HX_HPS_MIN_N", "2")
        P = Pair(hx, m, primes)
        own, sp = list(range(6)), [6, 7]
        a = P.rand(own, 9, batch=2)
        a[1, 0, :] = primes[1] - 1
        a[0, 1, ::2] = 0
        for digits in ([[0, 1], [2, 3], [4, 5]], [[0], [1, 2, 3], [4, 5]], [[0, 1, 2, 3, 4, 5]]):
            dg = hx.DoubleCRT(P.g, own, 2, a).breakIntoDigits(digits, sp)
            got = dg.download()
            for b in range(2):
                want = P.o.break_into_digits(own, a[:, b], digits, own + sp)
                assert np.array_equal(got[:, b].reshape(len(digits), 8, P.N), want), (hps, digits, b)
        d2 = hx.DoubleCRT(P.g, [1, 2, 3], 2, a[1:4])
        d2.addPrimes([0, 4, 6])
        got = d2.download()
        assert d2.getIndexSet() == [1, 2, 3, 0, 4, 6]
        for b in range(2):
            ext = P.o.add_primes([1, 2, 3], a[1:4, b], [0, 4, 6])
            assert np.array_equal(got[3:, b], ext) and np.array_equal(got[:3, b], a[1:4, b])
        for drop in ([4, 5], [0, 5], [1, 2, 3]):
            ev = hx.DoubleCRT(P.g, own, 2, a)
            keep = [i for i in own if i not in drop]
            ev.scaleDownToSet(keep, 65537)
            g2 = ev.download()
            for b in range(2):
                want = P.o.scale_down(own, a[:, b], drop, 65537)
                assert np.array_equal(g2[:, b], want), (hps, drop, b)


def _many_source_extension_cases(hx, n, form, valu_kernel, digits_too):
    """addPrimes, scaleDownToSet (ptxtSpace 65537 / 2 / 1, fdelta, norms) and -- digits_too -- breakIntoDigits with an
    n-prime digit, from n 60-bit source primes onto 60- / 56- / 45-bit ones on a small ring; every word against the
    oracle; the extension kernel that ran is checked by name."""
    m, B = 256, 3
    g60, g56, g45 = O.PrimeGen(60, m), O.PrimeGen(56, m), O.PrimeGen(45, m)
    primes = [g60.next() for _ in range(n + 6)] + [g56.next() for _ in range(3)] + [g45.next() for _ in range(2)]
    P = Pair(hx, m, primes)
    src = list(range(n))
    rest = list(range(n, n + 11))
    allp = src + rest
    # addPrimes: n sources -> 11 new primes
    a = P.rand(src, 31, batch=B)
    d = hx.DoubleCRT(P.g, src, B, a)
    hx.profileBegin()
    d.addPrimes(rest)
    names = " ".join(k["kernel"] for k in hx.profileEnd()["kernels"])
    assert ("rns_extend_mfma_kernel" in names) == (form == "mfma"), names
    if valu_kernel:   # (None: the control's kernel is also the redo pass of the matrix-core form)
        assert (valu_kernel in names) == (form == "valu"), names
    got = d.download()
    assert d.getIndexSet() == allp
    for b in range(B):
        assert np.array_equal(got[:n, b], a[:, b])
        assert np.array_equal(got[n:, b], P.o.add_primes(src, a[:, b], rest))
    # scaleDownToSet: drop the n primes, keep the 11 (the plan that carries P^-1 and the ptxtSpace correction)
    x = P.rand(allp, 32, batch=B)
    for ptxt in (65537, 2, 1):
        d = hx.DoubleCRT(P.g, allp, B, x)
        nrm, fd = hx.scaleDownToSetMulti([d], rest, ptxt, norms=True, fdelta=True)
        got = d.download()
        assert sorted(d.getIndexSet()) == rest
        gi = d.getIndexSet()
        for b in range(B):
            want, wfd = P.o.scale_down(allp, x[:, b], src, ptxt, want_fdelta=True)
            for r, i in enumerate(gi):
                assert np.array_equal(got[r, b], want[rest.index(i)]), (ptxt, i, b)
            assert np.allclose(fd[0, b], wfd, rtol=0, atol=1e-9 * max(1.0, float(ptxt)))
            assert np.isclose(nrm[0, b], O.embedding_largest_coeff(m, wfd), rtol=1e-9)
    if not digits_too:
        return
    # breakIntoDigits: digits of n and 4 primes (in both orders), special primes = the rest
    own = src + rest[:4]
    sp = rest[4:]
    y = P.rand(own, 33, batch=B)
    for digits in ([src, rest[:4]], [rest[:4], src]):
        dg, nr = hx.DoubleCRT(P.g, own, B, y).breakIntoDigits(digits, sp, norms=True)
        got = dg.download()
        nall = len(own) + len(sp)
        for b in range(B):
            want, wn = P.o.break_into_digits(own, y[:, b], digits, own + sp, want_norms=True)
            assert np.array_equal(got[:, b].reshape(len(digits), nall, P.N), want)
            assert np.allclose(nr[:, b], wn, rtol=1e-9)




@pytest.mark.parametrize("form", ["mfma", "valu"])
@pytest.mark.parametrize("eps", ["default", "0.05", "1.0"])
@pytest.mark.parametrize("n", [17, 24, 25, 33, 36, 40])
def test_wide_rns_kernel_17_to_40_source_primes(hx, monkeypatch, n, eps, form):
    """rns_extend_mfma_kernel (rns_mfma_kernels.hip, mfma_ext.h: the target sums as an int8 matrix product on the
    matrix cores, V_MFMA_I32_32X32X32_I8 -- the default) and rns_extend_wide_kernel<24/32/40> (rns_kernels.h: the same
    sums as 30-bit-limb multiply-adds, HX_NO_MFMA_EXT=1 -- the control), engine.hip launch_extend: the exact basis extension from
    17..40 source primes -- the 36-prime digits and the 36 dropped special primes of the reference's own benchmark
    chain (benchmarks/bgv_basic.cpp:247, bits = 6400) -- in its HPS form with the Garner redo pass behind it
    (eps = 2^-30: list almost always empty; 0.05: a third of the coefficients redone next to trusted ones; 1.0:
    all of them).  On a small ring (m = 256) so that the oracle's O(n^2) Garner stays cheap: addPrimes (toPoly +
    FFT on the new primes, src/DoubleCRT.cpp:565-599), scaleDownToSet for ptxtSpace 65537 / 2 / 1 with fdelta and
    norms (:1464-1516, src/Ctxt.cpp:466-507), and breakIntoDigits with an n-prime digit whose fix-up updates the
    later digit's rows in place (:479-561) -- 60-bit sources onto 60-, 56- and 45-bit targets, every word."""
    if eps != "default":
        monkeypatch.setenv("HX_HPS_EPS", eps)
    if form == "valu":
        monkeypatch.setenv("HX_NO_MFMA_EXT", "1")
    _many_source_extension_cases(hx, n, form, "rns_extend_wide_kernel", True)


@pytest.mark.parametrize("form", ["mfma", "valu"])
@pytest.mark.parametrize("eps", ["default", "0.05", "1.0"])
@pytest.mark.parametrize("n", [6, 9, 11, 12, 16])
def test_mfma_rns_kernel_9_to_16_source_primes(hx, monkeypatch, n, eps, form):
    """The matrix-core basis extension (rns_extend_mfma_kernel, 3 .. 5 MFMA steps) on the plans of the fast kernels:
    9 .. 16 source primes -- the 8 + 2 / 8 + 3 dropped primes of a CKKS level-2 mod-switch (benchmarks/ckks_basic.cpp,
    src/Ctxt.cpp:466-507 through src/DoubleCRT.cpp:1464-1516) -- in the HPS form with rns_extend_fast_kernel<n, Garner>
    over its redo list; the control (HX_NO_MFMA_EXT=1) is rns_extend_fast_kernel<n, HPS>.  addPrimes and scaleDownToSet
    for ptxtSpace 65537 / 2 / 1 with fdelta and norms, 60-bit sources onto 60- / 56- / 45-bit targets, every word.
    n = 6 under HX_MFMA_MIN_N=4: the two-step instantiation (measured at the BGV chain's 8-source level-2 mod-switch:
    154 us either way, so the default threshold stays at nine sources)."""
    if eps != "default":
        monkeypatch.setenv("HX_HPS_EPS", eps)
    if form == "valu":
        monkeypatch.setenv("HX_NO_MFMA_EXT", "1")
    if n < 9:   # (below the default threshold: two MFMA steps; the control there is the Garner form)
        monkeypatch.setenv("HX_MFMA_MIN_N", "4")
    _many_source_extension_cases(hx, n, form, "rns_extend_fast_kernel<%d, true" % n if n >= 9 else None, False)


def test_tensor_bring_to_set_when_none_of_the_listed_primes_is_there(hx):
    """hx_tensor_bring_to_set with a drop list that does not meet the operands' prime set (nothing to mod-switch):
    the product parts must still be formed -- the plain tensor product on the operands' primes."""
    P, own, sp = setup_rns(hx, m=16384, L=4, K=2)
    B = 2
    ops = [P.rand(own, 900 + i, batch=B) for i in range(4)]
    c0, c1, d0, d1 = (hx.DoubleCRT(P.g, own, B, x) for x in ops)
    outs = [hx.DoubleCRT(P.g, own, B, zero=False) for _ in range(3)]
    drop = hx._i32([sp[0], sp[1]])
    hx._chk(hx.lib().hx_tensor_bring_to_set(c0.h, c1.h, d0.h, d1.h, outs[0].h, outs[1].h, outs[2].h, None, 0,
                                            hx._p(drop), 2, 65537))
    want = list(hx.tensorProduct(c0, c1, d0, d1))
    for o, w in zip(outs, want):
        assert o.getIndexSet() == w.getIndexSet() == own
        assert np.array_equal(o.download(), w.download())
