"""The oracle's canonical-embedding norm (embeddingLargestCoeff, src/norms.cpp:129-262,480-493)
against its mathematical definition in numpy, and the measured-noise pieces built on it:
breakIntoDigits' per-digit norms (src/DoubleCRT.cpp:538-545) and the mod-switch fdelta
(src/Ctxt.cpp:466-507)."""
from functools import reduce

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_oracle_rns import make_ctx, rand_rows

RTOL = 1e-9   # floating point: the reference's own PGFFT tests use bounds of this class


def embed_def(m, f):
    """max over j in Z_m^* of |sum_i f_i W^(ij)| by direct evaluation"""
    zs = np.array(O.zmstar(m), dtype=np.float64)
    W = np.exp(2j * np.pi * np.outer(zs, np.arange(len(f))) / m)
    return float(np.abs(W @ np.asarray(f, dtype=np.float64)).max())


@pytest.mark.parametrize("m", [4, 8, 64, 1024, 12, 15, 105, 1705])
def test_embedding_norm_matches_definition(m):
    N = len(O.zmstar(m))
    rng = np.random.default_rng(m)
    for f in (rng.normal(size=N), rng.integers(-3, 4, size=N).astype(float), np.ones(N),
              np.eye(1, N, N - 1)[0]):
        got = O.embedding_largest_coeff(m, f)
        assert got == pytest.approx(embed_def(m, f), rel=RTOL, abs=1e-300)
    assert O.embedding_largest_coeff(m, np.zeros(N)) == 0.0
    # fewer coefficients than phi(m) (the reference accepts sz <= m/2)
    f = rng.normal(size=max(1, N // 2))
    assert O.embedding_largest_coeff(m, f) == pytest.approx(
        embed_def(m, np.concatenate([f, np.zeros(N - len(f))])), rel=RTOL)


def test_embedding_norm_pow2_large_matches_numpy_fft():
    m = 32768
    N = m // 2
    f = np.random.default_rng(3).normal(size=N)
    g = f * np.exp(2j * np.pi * np.arange(N) / m)
    want = np.abs(np.fft.ifft(g) * N).max()          # sum_i g_i V^(+ij)
    assert O.embedding_largest_coeff(m, f) == pytest.approx(want, rel=RTOL)


@pytest.mark.parametrize("m,digits", [(32, [[0, 1], [2, 3], [4]]), (15, [[0, 1, 2], [3, 4]])])
def test_digit_norms_are_the_norms_of_the_centred_digits(m, digits):
    ctx = make_ctx(m, 7)
    own = sorted(p for d in digits for p in d)
    all_idx = own + [5, 6]
    rows = rand_rows(ctx, own, 4)
    plain = ctx.break_into_digits(own, rows, digits, all_idx)
    got, nrm = ctx.break_into_digits(own, rows, digits, all_idx, want_norms=True)
    assert np.array_equal(got, plain)
    # each digit recovered exactly from its own rows, then the definition
    for di, d in enumerate(digits):
        pos = [all_idx.index(i) for i in d]
        dig = ctx.to_poly(d, got[di][pos])
        P = reduce(lambda a, b: a * b, [ctx.primes[i] for i in d])
        frac = np.array([v / P for v in dig])
        assert max(abs(frac)) <= 0.5
        assert nrm[di] == pytest.approx(embed_def(m, frac), rel=1e-9)


@pytest.mark.parametrize("m,ptxt", [(32, 257), (15, 7), (16, 4), (32, 1)])
def test_fdelta_is_delta_over_the_dropped_product(m, ptxt):
    ctx = make_ctx(m, 5)
    own, drop = [0, 1, 2, 3, 4], [3, 4]
    rows = rand_rows(ctx, own, 11)
    out, fd = ctx.scale_down(own, rows, drop, ptxt, want_fdelta=True)
    assert np.array_equal(out, ctx.scale_down(own, rows, drop, ptxt))
    # sanity check of the reference (src/Ctxt.cpp:481-487): |fdelta| <= ptxtSpace/2
    assert np.abs(fd).max() <= ptxt / 2.0 + 1e-4
    # delta = (centred value v of the dropped rows) - P*k with k making it 0 mod ptxtSpace
    P = ctx.primes[3] * ctx.primes[4]
    v = ctx.to_poly(drop, rows[3:])
    from fractions import Fraction
    for j in range(ctx.N):
        k = round(Fraction(v[j], P) - Fraction(fd[j]))
        delta = v[j] - P * k
        assert abs(float(Fraction(delta, P)) - fd[j]) < 1e-9
        if ptxt > 1:
            assert delta % ptxt == 0
        else:
            assert k == 0
