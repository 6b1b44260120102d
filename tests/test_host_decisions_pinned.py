"""Prime-set decisions of the host side pinned on hand-derived cases (tests/golden/host_decisions.json,
derivation beside it): ModuliSizes::init / getSet4Size (src/primeChain.cpp:68-335) and
computeIntervalForMul (src/Ctxt.cpp:1610-1656).  The expected values come from the reference's source
worked through by hand, not from helib_amd/ctxt.py or include/helib_amd_ctxt.hpp -- both are checked
here.  CPU only."""
import json
import math
import os
import subprocess
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "host_decisions.json")))
LN2 = math.log(2.0)


def _table(pow2):
    from helib_amd import ctxt as hc
    ch = FIX["chain"]
    stub = types.SimpleNamespace(pow2=pow2, primes=ch["primes"], smallPrimes=ch["small"], ctxtPrimes=ch["ctxt"])
    return hc.ModuliSizes(stub)


def test_table_order_python():
    t = _table(True)
    assert [sorted(s) for _, s in t.sizes] == FIX["table_order"]
    assert t.iFFT_cost == 0 and _table(False).iFFT_cost == 20


@pytest.mark.parametrize("case", FIX["getSet4Size"], ids=lambda c: c["name"])
def test_getSet4Size_python(case):
    t = _table(case["pow2"])
    got = t.getSet4Size(case["low_bits"] * LN2, case["high_bits"] * LN2, case["from1"], case.get("from2"),
                        case["reverse"])
    assert sorted(got) == case["want"]


@pytest.mark.parametrize("case", FIX["computeIntervalForMul"], ids=lambda c: c["name"])
def test_computeIntervalForMul_python(case):
    from helib_amd import ctxt as hc

    def stub(d):
        return types.SimpleNamespace(logOfPrimeSet=lambda: d["logOfPrimeSet"], lnNoise=d["lnNoise"],
                                     modSwitchAddedNoiseBound=lambda: math.exp(d["ln_adn"]),
                                     context=types.SimpleNamespace(ckks=case["ckks"]))
    lo, hi = hc.Ctxt.computeIntervalForMul(stub(case["c1"]), stub(case["c2"]))
    assert abs(lo - case["lo"]) < 1e-9 and abs(hi - case["hi"]) < 1e-9
    assert abs((hi - lo) - 4 * LN2) < 1e-12


def test_getSet4Size_cpp(tmp_path):
    exe = str(tmp_path / "decisions_test")
    libdir = os.path.join(ROOT, "helib_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "decisions_test.cpp"), "-L" + libdir, "-lhelib_amd",
                           "-Wl,-rpath," + libdir, "-o", exe])
    ch = FIX["chain"]

    def ints(v):
        return f"{len(v)} " + " ".join(map(str, v))
    for pow2 in (True, False):
        cases = [c for c in FIX["getSet4Size"] if c["pow2"] == pow2]
        lines = [f"table {int(pow2)} {ints(ch['primes'])} {ints(ch['small'])} {ints(ch['ctxt'])}"]
        for c in cases:
            f2 = ints(c["from2"]) if "from2" in c else "-1"
            lines.append(f"set4 {c['low_bits'] * LN2!r} {c['high_bits'] * LN2!r} {int(c['reverse'])} {ints(c['from1'])} {f2}")
        run = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
        rows = run.stdout.splitlines()
        # the reference's hooks (include/helib_amd_timing.hpp): the window statistics were collected
        # (src/primeChain.cpp:207-208, 288-289) and the named timer ran once
        assert "timer decisions_test_tail calls 1" in run.stderr
        assert "window1-nchoices ave=" in run.stderr and "window1-out ave=" in run.stderr
        if any("from2" in c for c in cases):
            assert "window2-nchoices ave=" in run.stderr
        assert int(rows[0]) == len(FIX["table_order"])
        for c, r in zip(cases, rows[1:]):
            assert [int(x) for x in r.split()] == c["want"], c["name"]
