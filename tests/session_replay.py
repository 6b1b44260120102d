"""The C++ host's benchmark session (helib_amd/csrc/host_session.cpp, what bench.py times) replayed on the CPU oracle:
the session's own operands, key-switching matrix and bookkeeping are downloaded (hxh_ctxt_rows / hxh_relin_matrix /
hxh_ctxt_info), the same host logic (helib_amd.ctxt, the python mirror) is driven over oracle/backend.py -- the
reference's unfused sequence -- once per batch element, and EVERY word of the session's kept products of level 1 and
level 2 is compared.  Test infrastructure (used by the mock-backed CPU test and by the GPU test)."""
import numpy as np

from helib_amd import ctxt as hc
from oracle import oracle as O
from oracle.backend import OKeySwitch, OPoly, OracleOps


def replay_and_compare(s, scheme, m, p, r, bits, measure=True, levels=(1, 2), elements=None):
    """s: a helib_amd.host.Session that has not multiplied yet.  Runs one multiply per level in the session and on
    the oracle; returns the number of words compared.  elements: the batch elements replayed on the oracle (default
    all) -- a stratified sample at the batch sizes bench.py times (128 / 64), where the work-to-workgroup maps of the
    kernels (md_tile, xcd_remap: functions of the launch size) differ from the small batches' and the oracle's
    ~9 multiplies per second allow a handful of elements."""
    ckks = scheme == "ckks"
    cc = hc.ChainContext(m, -1 if ckks else p, r, bits=bits, c=3, ckks=ckks)
    assert s.chain_primes() == [int(q) for q in cc.primes], "the C++ chain differs from the python mirror's"
    octx = O.Ctx(m)
    for q in cc.primes:
        octx.add_prime(q)
    ops = OracleOps(octx)
    widx, wb, wa = s.relin_matrix()
    assert widx == list(cc.ctxtPrimes) + list(cc.specialPrimes)
    oW = OKeySwitch(widx, wb, wa)
    old_measure = hc.Ctxt.measure
    hc.Ctxt.measure = bool(measure)
    compared = 0
    try:
        B = s.batch
        elems = list(range(B)) if elements is None else [int(b) for b in elements]
        assert all(0 <= b < B for b in elems) and len(set(elems)) == len(elems)
        operands = []
        for which in (0, 1):
            info = s.ctxt_info(0, which)
            parts = [s.ctxt_rows(0, which, part) for part in (0, 1)]
            assert all(idx == list(cc.ctxtPrimes) for idx, _ in parts)
            cts = []
            for b in elems:
                c = hc.Ctxt(cc, ops, oW, ksw_ptxtSpace=int(info["key_ptxtSpace"]), ksw_noise=info["key_lnNoise"])
                c.parts = {"1": OPoly(octx, parts[0][0], parts[0][1][:, b]), "s": OPoly(octx, parts[1][0], parts[1][1][:, b])}
                c.primeSet = frozenset(cc.ctxtPrimes)
                c.ptxtSpace, c.intFactor = int(info["ptxtSpace"]), int(info["intFactor"])
                c.ptxtMag, c.lnRatFactor = info["ptxtMag"], info["lnRatFactor"]
                c.lnNoise = info["lnNoise"]
                cts.append(c)
            operands.append(cts)
        cur = None
        for level in levels:
            s.multiply(level, 1, measure)
            if level == 1:
                cur = operands[0]
                for e in range(len(elems)):
                    cur[e].multiplyBy(operands[1][e])
            else:
                for e in range(len(elems)):
                    cur[e].multiplyBy(cur[e].clone())
            info = s.ctxt_info(level)
            assert int(info["nparts"]) == 2
            # the batch shares one prime-set decision: its estimate takes the largest norm of the batch per part and
            # digit, so it is the largest of its elements' estimates or slightly above
            # (a sample of the batch need not contain the elements with the largest norms, and the batch takes the
            # largest per part and digit separately -- at batch 128 the session's estimate sat 0.31 above the best of
            # five sampled elements: a wider margin above; the data comparison below is what this replay is for)
            worst = max(c.lnNoise for c in cur)
            assert worst - 1e-6 <= info["lnNoise"] <= worst + (0.05 if elements is None else 0.7), (level, worst, info["lnNoise"])
            for c in cur:
                assert sorted(c.primeSet) == s.result_primes(level)
                assert c.intFactor == int(info["intFactor"]) and abs(c.lnRatFactor - info["lnRatFactor"]) < 1e-9
                c.lnNoise = info["lnNoise"]
            for part, h in enumerate(("1", "s")):
                idx, rows = s.ctxt_rows(level, 0, part)
                for e, b in enumerate(elems):
                    oi, od = cur[e].parts[h].getIndexSet(), cur[e].parts[h].rows
                    assert sorted(idx) == sorted(oi)
                    for rr, i in enumerate(idx):
                        assert np.array_equal(rows[rr, b], od[oi.index(i)]), (level, h, i, b)
                        compared += rows.shape[2]
    finally:
        hc.Ctxt.measure = old_measure
    return compared
