"""The C++ host side (include/helib_amd_ctxt.hpp, helib_amd_keys.hpp) on the CPU: the same C++ programs the
`-m gpu` suite runs against libhelib_amd.so are linked here against tests/cpp/hx_mock.cpp, a stand-in for the
C ABI over the CPU oracle (TEST INFRASTRUCTURE, built by this module into a temporary directory and linked by
nothing else).  What is under test is the host control flow -- prime-set decisions, noise bookkeeping, handle
algebra, key management, hoisting -- with every result decrypted and compared with plain arithmetic modulo
(Phi_m(X), p) by the programs themselves."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


@pytest.fixture(scope="module")
def mock(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("hxmock"))
    obj = os.path.join(d, "hx_oracle.o")
    subprocess.check_call(["gcc", "-O3", "-fPIC", "-std=c11", "-c", os.path.join(ROOT, "oracle", "hx_oracle.c"), "-o", obj])
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Werror", "-I" + INC,
                           "-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "cpp", "hx_mock.cpp"), obj,
                           "-lm", "-o", os.path.join(d, "libhx_mock.so")])

    def build(src, name):
        exe = os.path.join(d, name)
        if not os.path.exists(exe):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + INC, src, "-L" + d, "-lhx_mock",
                                   "-Wl,-rpath," + d, "-o", exe])
        return exe
    build.dir = d
    return build


def _env(defer):
    """HX_TEST_DEFER_NORMS: the programs switch Ctxt::deferNorms() on -- measured norms are read back lazily
    (LazyLn); the mock then leaves a *_norms array poisoned until hx_norms_flush, as the engine leaves it unwritten"""
    e = dict(os.environ)
    e.pop("HX_TEST_DEFER_NORMS", None)
    if defer:
        e["HX_TEST_DEFER_NORMS"] = "1"
    return e


@pytest.mark.parametrize("m,p,bits,measure", [(128, 257, 150, 0), (128, 2, 300, 1), (1024, 65537, 250, 1),
                                              (128, -1, 250, 0), (256, -1, 300, 1), (128, 2, 300, 2), (256, -1, 300, 2)])
def test_cpp_keys_encrypt_multiply_rotate_decrypt_over_the_mock(mock, m, p, bits, measure):
    """tests/cpp/keys_test.cpp (key generation, Encrypt, multiplyBy, addCtxt, smartAutomorph in one and two
    steps, Decrypt; CKKS: products over two levels, sums across scaling factors) -- the program of the GPU
    suite, here with the polynomial work done by the oracle."""
    exe = mock(os.path.join(ROOT, "tests", "cpp", "keys_test.cpp"), "keys_test")
    r = subprocess.run([exe, str(m), str(p), str(bits), str(min(measure, 1))], capture_output=True, text=True, timeout=300,
                       env=_env(measure == 2))          # measure = 2: measured noise with the lazy read-back
    assert r.returncode == 0 and "keys_test OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("m,p,bits,measure", [(128, 257, 300, 0), (128, 257, 300, 1), (128, 3, 300, 0),
                                              (256, 65537, 400, 1), (128, -1, 400, 0), (256, -1, 500, 1),
                                              (105, 2, 300, 0), (105, 257, 300, 1), (45, 2, 300, 1), (1705, 2, 300, 0),
                                              (128, 257, 300, 2), (256, -1, 500, 2), (105, 257, 300, 2)])
def test_cpp_ctxt_operations_over_the_mock(mock, m, p, bits, measure):
    """tests/cpp/ctxt_ops_test.cpp: multiplyBy2 / cube / power through the 4-part ciphertext and
    keySwitchPart (src/Ctxt.cpp:1776-1828, 720-842), totalProduct / incrementalProduct / innerProduct
    (:2803-2904), BasicAutomorphPrecon (src/matmul.cpp:48-184), frobeniusAutomorph (p = 3: order 32 modulo
    128), multByConstant / addConstant with scalars and DoubleCRT constants, capacity / isCorrect, and the
    CKKS forms.  General m (105, 45 = 9*5, 1705): keys and encryptions from the samplers that reduce modulo
    Phi_m, arithmetic checked modulo (Phi_m, p)."""
    exe = mock(os.path.join(ROOT, "tests", "cpp", "ctxt_ops_test.cpp"), "ctxt_ops_test")
    r = subprocess.run([exe, str(m), str(p), str(bits), str(min(measure, 1))], capture_output=True, text=True, timeout=300,
                       env=_env(measure == 2))          # measure = 2: measured noise with the lazy read-back
    assert r.returncode == 0 and "ctxt_ops_test OK" in r.stdout, r.stdout + r.stderr


def test_cpp_polyNormBnd_matches_the_python_mirror(mock, tmp_path):
    """calcPolyNormBnd (src/PAlgebra.cpp:215-434) in C++ against helib_amd.ctxt.polyNormBnd: power of two,
    prime-power odd part (closed form) and the general case (inverse Vandermonde row sums)."""
    from helib_amd.ctxt import polyNormBnd
    src = tmp_path / "pnb.cpp"
    src.write_text('#include <cstdio>\n#include <cstdlib>\n#include "helib_amd_ctxt.hpp"\n'
                   'int main(int c, char** v) { for (int i = 1; i < c; i++) '
                   'printf("%.15g\\n", helib_amd::polyNormBnd(atol(v[i]))); }\n')
    exe = mock(str(src), "pnb")
    ms = [128, 12, 45, 105, 1705, 4095]
    out = subprocess.run([exe] + [str(m) for m in ms], capture_output=True, text=True, timeout=300).stdout.split()
    assert len(out) == len(ms)
    for m, got in zip(ms, out):
        assert float(got) == pytest.approx(polyNormBnd(m), rel=1e-9), m


def test_cpp_cyclotomic_matches_the_python_side(mock, tmp_path):
    """Phi_m(X) by exact division in C++ (the samplers' reduceModPhimX and calcPolyNormBnd use it) against
    helib_amd.hostnt.phimx, incl. m with repeated prime factors"""
    from helib_amd import hostnt
    src = tmp_path / "cyc.cpp"
    src.write_text('#include <cstdio>\n#include <cstdlib>\n#include "helib_amd_ctxt.hpp"\n'
                   'int main(int c, char** v) { for (int i = 1; i < c; i++) { '
                   'for (long a : helib_amd::cyclotomic(atol(v[i]))) printf("%ld ", a); printf("\\n"); } }\n')
    exe = mock(str(src), "cyc")
    ms = [1, 2, 12, 45, 105, 128, 385, 1705, 4095]
    out = subprocess.run([exe] + [str(m) for m in ms], capture_output=True, text=True, timeout=300).stdout.strip().split("\n")
    assert len(out) == len(ms)
    for m, line in zip(ms, out):
        assert [int(x) for x in line.split()] == [int(c) for c in hostnt.phimx(m)], m


def test_the_mock_is_test_infrastructure_only():
    """nothing under helib_amd/, include/, bench.py or __graft_entry__.py refers to the mock"""
    hits = []
    for base, _, files in os.walk(ROOT):
        if any(part in base for part in (os.sep + "tests", os.sep + ".git", os.sep + "gpurun_out", "__pycache__")):
            continue
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".sh")):
                with open(os.path.join(base, f), errors="ignore") as fh:
                    if "hx_mock" in fh.read():
                        hits.append(os.path.join(base, f))
    assert not hits, hits


@pytest.mark.parametrize("m,p,bits,k,measure", [(128, 257, 150, 3, 0), (128, 257, 150, 5, 1), (1024, 65537, 250, 3, 1),
                                                (105, 2, 200, 2, 0), (128, 257, 150, 5, 2), (1024, 65537, 250, 3, 2)])
def test_cpp_ctxt_matches_the_python_mirror_over_the_oracle(mock, m, p, bits, k, measure, tmp_path, monkeypatch):
    """tests/cpp/ctxt_test.cpp (the program of the GPU suite's test_cpp_host_ctxt_matches_python_mirror) over the
    mock, against helib_amd.ctxt over the oracle backend on the same keys and ciphertexts: prime sets,
    intFactor, noise estimate and every word of every part of the product, product + product and the product
    rotated in one and in two steps; the error paths (LogicError, InvalidArgument) counted."""
    import struct

    import numpy as np

    from helib_amd import ctxt as hc
    from oracle import oracle as O
    from oracle.backend import OKeySwitch, OPoly, OracleOps
    from tests import test_ctxt_host as T
    exe = mock(os.path.join(ROOT, "tests", "cpp", "ctxt_test.cpp"), "ctxt_test")
    defer, measure = measure == 2, min(measure, 1)      # (2: the C++ side reads its measured norms back lazily)
    monkeypatch.setattr(hc.Ctxt, "measure", bool(measure))
    cc = hc.ChainContext(m, p, 1, bits=bits, c=3)
    octx = O.Ctx(m)
    for q in cc.primes:
        octx.add_prime(q)
    N = octx.N
    s, allp, kb, ka, rows = T.make_keys(cc, octx)
    _, _, kbk, kak, _ = T.make_keys(cc, octx, auto_k=k)
    rng = np.random.default_rng(4)
    ma, mb = rng.integers(0, p, size=N), rng.integers(0, p, size=N)
    ea, eb = T.encrypt(cc, octx, s, ma, 1, rows), T.encrypt(cc, octx, s, mb, 2, rows)
    L, D = len(cc.ctxtPrimes), len(cc.digits)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<10q", m, p, bits, k, measure, len(cc.primes), D, len(allp), L, N))
        f.write(np.array(octx.roots, dtype="<u8").tobytes())
        for arr in (kb, ka, kbk, kak, ea[0], ea[1], eb[0], eb[1]):
            f.write(np.ascontiguousarray(arr, dtype="<u8").tobytes())
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300, env=_env(defer))
    assert r.returncode == 0, r.stdout + r.stderr
    ops = OracleOps(octx)
    W, Wk = OKeySwitch(allp, kb, ka), OKeySwitch(allp, kbk, kak)
    mk = lambda e: hc.Ctxt.fresh(cc, ops, *(OPoly(octx, cc.ctxtPrimes, x) for x in e), ksw=W)  # noqa: E731
    ga, gb = mk(ea), mk(eb)
    ga.ksw_auto = {k: Wk}
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
    assert errs == 3 and off + 8 == len(buf)
    from tests import bgv_ref as B
    ab = [int(v) for v in B.polymul_mod_phi(ma, mb, m, p)]
    assert T.decrypt(cc, octx, s, prod, rows) == ab
    assert T.decrypt(cc, octx, s, ga, rows) == [int(v) for v in B.automorph_mod_phi(ab, m, pow(k, 3, m), p)]


@pytest.mark.parametrize("m,p,bits", [(128, 257, 300), (1024, 65537, 300), (128, 2, 300), (128, -1, 400)])
def test_cpp_ciphertexts_cross_the_reference_wire_format(mock, m, p, bits, tmp_path):
    """include/helib_amd_io.hpp (Ctxt::writeTo / Ctxt::read for the C++ host; tests/cpp/io_test.cpp): fresh,
    relinearised and 3-part ciphertexts written in the reference's 2.2.0 binary layout and read back as working
    ciphertexts (same bookkeeping, same decryption, usable in a further multiplication), IOError on malformed
    input.  Every blob the C++ side wrote is parsed here with helib_amd.wire -- whose layout is pinned on the
    reference's own binary fixture -- and written back byte for byte; the JSON text the C++ side wrote for the same
    ciphertext (Ctxt::writeToJSON) equals the python side's."""
    import struct

    from helib_amd import wire
    exe = mock(os.path.join(ROOT, "tests", "cpp", "io_test.cpp"), "io_test")
    out = str(tmp_path / "blobs.bin")
    r = subprocess.run([exe, str(m), str(p), str(bits), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "io_test OK" in r.stdout, r.stdout + r.stderr
    import json
    buf = open(out, "rb").read()
    off, items = 0, []
    while off < len(buf):
        (n,) = struct.unpack_from("<q", buf, off)
        items.append(buf[off + 8:off + 8 + n])
        off += 8 + n
    assert len(items) == 2 * (1 if p == -1 else 3)
    for blob, text in zip(items[0::2], items[1::2]):      # (binary object, JSON text of the same ciphertext)
        d, used = wire.read_ctxt(blob)
        assert used == len(blob) and wire.write_ctxt(d) == blob
        assert all(idx == d["primeSet"] for idx, _, _ in d["parts"])
        assert json.loads(text) == wire.ctxt_to_json(d)


def test_cpp_lazy_norm_read_back_under_the_address_sanitizer(mock, tmp_path):
    """Ctxt::deferNorms(): a *_norms array is written by the library at the NEXT flush, whatever has become of the
    ciphertext that asked for it (a result dropped without its noise estimate being read).  The operations test,
    built with -fsanitize=address against a sanitized mock whose deferral leaves the arrays unwritten until
    hx_norms_flush, must run clean: arrays outlive the flush (Context::normBuffer), estimates are completed before
    they are used (LazyLn)."""
    d = str(tmp_path)
    obj = os.path.join(d, "hx_oracle.o")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-c", os.path.join(ROOT, "oracle", "hx_oracle.c"), "-o", obj])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fPIC", "-shared", "-I" + INC,
                           "-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "cpp", "hx_mock.cpp"), obj,
                           "-lm", "-o", os.path.join(d, "libhx_mock_asan.so")])
    exe = os.path.join(d, "ctxt_ops_asan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-I" + INC,
                           os.path.join(ROOT, "tests", "cpp", "ctxt_ops_test.cpp"), "-L" + d, "-lhx_mock_asan",
                           "-Wl,-rpath," + d, "-o", exe])
    for args in (("128", "257", "300", "1"), ("128", "-1", "400", "1")):
        r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600, env=_env(True))
        assert r.returncode == 0 and "ctxt_ops_test OK" in r.stdout and "AddressSanitizer" not in r.stderr, \
            r.stdout + r.stderr[-2000:]


@pytest.mark.parametrize("scheme,m,p,r,bits,batch,measure", [("bgv", 128, 257, 1, 300, 3, False), ("bgv", 256, 65537, 1, 400, 2, True),
                                                           ("ckks", 256, -1, 20, 500, 2, True), ("ckks", 128, -1, 20, 400, 3, False),
                                                           # odd squarefree general m (config 5's kind of ring, small): the plaintext
                                                           # product modulo (Phi_m, p) by binomial passes (Session.expected)
                                                           ("bgv", 255, 2, 1, 300, 2, True), ("bgv", 1285, 7, 1, 300, 2, False)])
def test_cpp_host_session_library_over_the_mock(mock, scheme, m, p, r, bits, batch, measure):
    """helib_amd/csrc/host_session.cpp (the C++17 host behind include/helib_amd_host.h, the library bench.py times)
    linked against the CPU stand-in for the C ABI: keys and batched encryptions made in C++, the benchmark loop
    `copy = ctxt1; copy.multiplyBy(ctxt2)` at level 1 and level 2 (operands carrying special primes), every batch
    element of both products decrypted in C++ and compared here with the plaintext product -- BGV exactly, CKKS
    within the bound the ciphertext reports."""
    from helib_amd import build as hb, host
    so = hb.build_host(force=True, link_dir=mock.dir, link_lib="hx_mock", out=os.path.join(mock.dir, "libhelib_amd_host_mock.so"))
    s = host.Session(scheme, m, p, r, bits, batch, seed=11, lib_path=so)
    assert s.batch == batch and s.D >= 1 and (s.phim == m // 2 or m % 2 == 1)
    assert s.verify(0) == batch                                   # decrypt(encrypt(m)) == m
    s.multiply(1, 2, measure)
    assert s.verify(1) == batch
    s.multiply(2, 2, measure)
    assert s.verify(2) == batch
    assert set(s.result_primes(2)) < set(s.result_primes(1)) or len(s.result_primes(2)) <= len(s.result_primes(1))
    s.multiply_single(measure)
    with pytest.raises(host.HostError):
        s.multiply(3, 1)
    s.close()


@pytest.mark.parametrize("scheme,m,p,r,bits,batch,measure", [("bgv", 256, 65537, 1, 400, 2, True), ("ckks", 256, -1, 1, 500, 2, True),
                                                           ("bgv", 128, 257, 1, 300, 2, False)])
def test_cpp_session_products_equal_the_oracle_replay_and_one_key_pair_serves_two_sessions(mock, scheme, m, p, r, bits, batch, measure):
    """(a) tests/session_replay.py: the session's own operands, relinearisation matrix and bookkeeping downloaded
    (hxh_ctxt_rows / hxh_relin_matrix / hxh_ctxt_info), the python mirror driven over the oracle backend per batch
    element, every word of the kept products of level 1 and level 2 compared -- the C++ host's fused calls
    (hx_tensor_bring_to_set / hx_mul_relin through the mock) against the reference's unfused sequence.
    (b) hxh_export_keys / hxh_session_create_with_keys (one key pair replicated): a second session built
    from the first one's key material, with its own encryptions, holds the same matrix, and its products decrypt
    under the imported secret key."""
    from helib_amd import build as hb, host
    from tests.session_replay import replay_and_compare
    so = hb.build_host(force=True, link_dir=mock.dir, link_lib="hx_mock", out=os.path.join(mock.dir, "libhelib_amd_host_mock.so"))
    s = host.Session(scheme, m, p, r, bits, batch, seed=21, lib_path=so)
    keys = s.export_keys()
    assert replay_and_compare(s, scheme, m, p, r, bits, measure) > 0
    t = host.Session(scheme, m, p, r, bits, batch, seed=22, lib_path=so, keys=keys)
    si, sb, sa = s.relin_matrix()
    ti, tb, ta = t.relin_matrix()
    assert si == ti and np.array_equal(sb, tb) and np.array_equal(sa, ta)
    assert not np.array_equal(s.ctxt_rows(0, 0, 0)[1], t.ctxt_rows(0, 0, 0)[1])     # its own encryptions
    assert t.verify(0) == batch
    t.multiply(1, 1, measure)
    assert t.verify(1) == batch
    with pytest.raises(host.HostError):
        host.Session(scheme, m, p, r, bits + 200, batch, seed=23, lib_path=so, keys=keys)   # another chain
    for bad in (keys[:-5], np.concatenate([keys, keys[:3]]), keys[:40]):   # truncated / trailing words: refused as a whole
        with pytest.raises(host.HostError):
            host.Session(scheme, m, p, r, bits, batch, seed=24, lib_path=so, keys=bad)
    # batched encryption / decryption through the session (SecKey::EncryptBatch / DecryptBatch): BGV only
    if scheme == "bgv":
        ed = s.encrypt_decrypt_batch(3, reps=1)
        assert ed["batch"] == 3 and ed["all_elements_round_trip"] and ed["encrypt_ms_per_ciphertext"] > 0
    else:
        with pytest.raises(host.HostError):
            s.encrypt_decrypt_batch(3, reps=1)
    s.close()
    t.close()


@pytest.mark.parametrize("m", [128, 1024, 105, 1705])
def test_cpp_cmodulus_bignum_toPoly_and_namespace_intel_over_the_mock(mock, m):
    """tests/cpp/facade2_test.cpp: Cmodulus{FFT, iFFT} against the O(N^2) definition, DoubleCRT::toPoly with
    big-integer coefficients (every coefficient reduces to the inverse-transformed rows and lies in the centred
    range), and `namespace intel` (include/helib_amd_intel.hpp) through call sites shaped like the reference's
    USE_INTEL_HEXL ones -- the program of the GPU suite, here over the CPU stand-in for the C ABI."""
    exe = mock(os.path.join(ROOT, "tests", "cpp", "facade2_test.cpp"), "facade2_test")
    r = subprocess.run([exe, str(m)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "facade2_test OK" in r.stdout, r.stdout + r.stderr


# ---------------------------------------------------------------- round 6: ciphertexts cross process boundaries
def _blob_ctxts(blob):
    """the wire ciphertexts of a blob, parsed by the python reader of the reference's format (helib_amd/wire.py)"""
    from helib_amd import wire
    buf, off, out = bytes(blob), 0, []
    while off < len(buf):
        c, off = wire.read_ctxt(buf, off)
        out.append(c)
    return out


@pytest.mark.parametrize("scheme,m,p,r,bits,measure", [("bgv", 256, 65537, 1, 400, True), ("ckks", 256, -1, 20, 500, True),
                                                     ("bgv", 128, 257, 1, 300, False)])
def test_ciphertexts_scattered_multiplied_under_public_keys_and_gathered(mock, scheme, m, p, r, bits, measure):
    """The batch split as a SERVICE (north_star: "RCCL over xGMI only for the batch split"; SURVEY 2.3 row C1, 8e;
    VERDICT r5 missing 1), here in one process: a source session encrypts 5 pairs and keeps the secret key; its PUBLIC
    key material (hxh_export_public_keys) and two slices of its ciphertexts in the reference's binary format
    (hxh_export_ctxts = Ctxt::writeTo per element, src/Ctxt.cpp:2584-2611) make two worker sessions
    (hxh_session_create_from_ctxts) that multiply and cannot decrypt; their products come back as blobs and the
    source alone decrypts every one of them (hxh_decrypt_wire) against its own plaintexts.  The products' rows equal,
    word for word, those of the source multiplying the whole batch itself."""
    from helib_amd import build as hb, host
    so = hb.build_host(force=True, link_dir=mock.dir, link_lib="hx_mock", out=os.path.join(mock.dir, "libhelib_amd_host_mock.so"))
    total, slices = 5, [(0, 3), (3, 2)]
    src = host.Session(scheme, m, p, r, bits, total, seed=61, lib_path=so, source=True)
    pub, full = src.export_public_keys(), src.export_keys()
    assert pub.size == full.size and not np.array_equal(pub, full)
    whole = _blob_ctxts(src.export_ctxts(0, 0))
    assert len(whole) == total and all(len(c["parts"]) == 2 for c in whole)
    idx0, rows0 = src.ctxt_rows(0, 0, 0)
    assert whole[2]["parts"][0][0] == sorted(idx0)
    assert np.array_equal(np.asarray(whole[2]["parts"][0][1])[[sorted(idx0).index(i) for i in idx0]], rows0[:, 2])
    products = []
    for first, count in slices:
        a, b = src.export_ctxts(0, 0, first, count), src.export_ctxts(0, 1, first, count)
        assert len(_blob_ctxts(a)) == count
        w = host.Session(scheme, m, p, r, bits, count, lib_path=so, keys=pub, operands=(a, b))
        assert w.batch == count
        with pytest.raises(host.HostError):
            w.plaintext(0)                      # it was handed ciphertexts: no plaintexts ...
        with pytest.raises(host.HostError):
            w.decrypt(0, 0)                     # ... and no secret key
        w.multiply(1, 1, measure)
        with pytest.raises(host.HostError):
            w.decrypt(1, 0)
        w.multiply_single(measure)
        products.append(w.export_ctxts(1, 0))
        w.close()
    # rank 0 alone verifies: every product decrypts to its own plaintext product
    for (first, count), blob in zip(slices, products):
        assert src.verify_blob(blob, 1, first, count) == count
    with pytest.raises(host.HostError):
        src.verify_blob(products[0], 1, 1, 3)   # (the wrong elements: the check is not vacuous)
    with pytest.raises(host.HostError):
        src.verify_blob(products[0][:-9], 1, 0, 3)
    # the same multiplications by the source on the whole batch: identical rows (the products of a slice do not
    # depend on which other ciphertexts share its launch)
    src.multiply(1, 1, measure)
    mine = _blob_ctxts(src.export_ctxts(1, 0))
    theirs = [c for blob in products for c in _blob_ctxts(blob)]
    assert len(mine) == len(theirs) == total
    for x, y in zip(mine, theirs):
        assert x["primeSet"] == y["primeSet"] and x["intFactor"] == y["intFactor"] and x["ptxtSpace"] == y["ptxtSpace"]
        for (ia, ra, ha), (ib, rb, hb_) in zip(x["parts"], y["parts"]):
            assert ia == ib and ha == hb_ and np.array_equal(np.asarray(ra), np.asarray(rb))
    # a blob with a residue out of range, and key material of another chain, are refused
    bad = np.array(products[0], copy=True)
    a, b = src.export_ctxts(0, 0, 0, 2), src.export_ctxts(0, 1, 0, 2)
    with pytest.raises(host.HostError):
        host.Session(scheme, m, p, r, bits, 3, lib_path=so, keys=pub, operands=(a, b))          # two ciphertexts, batch 3
    with pytest.raises(host.HostError):
        host.Session(scheme, m, p, r, bits + 200, 2, lib_path=so, keys=pub, operands=(a, b))    # another chain
    a[len(a) // 2: len(a) // 2 + 8] = 0xff
    with pytest.raises(host.HostError):
        host.Session(scheme, m, p, r, bits, 2, lib_path=so, keys=pub, operands=(a, b))
    del bad
    src.close()


def test_two_ranks_scatter_multiply_gather_over_gloo(mock, tmp_path):
    """The same over a process group of world size 2 (gloo; RCCL on the GPUs): helib_amd.dist.Group.scatter_blobs /
    gather_blobs move the wire-format slices as one tensor per rank, rank 1 holds public key material only, rank 0
    decrypts and verifies all five products (what `bench.py --scatter` does at N > 1)."""
    import json
    import socket
    import sys
    import textwrap
    from helib_amd import build as hb
    so = hb.build_host(force=True, link_dir=mock.dir, link_lib="hx_mock", out=os.path.join(mock.dir, "libhelib_amd_host_mock.so"))
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(f"""
        import sys, json
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        from helib_amd import host
        from helib_amd.dist import Group, shard
        so = {so!r}
        P = ("bgv", 256, 65537, 1, 400)
        g = Group(backend="gloo")
        total = 5
        src = host.Session(*P, total, seed=71, lib_path=so, source=True) if g.rank == 0 else None
        keys, kb = g.broadcast_words(src.export_public_keys() if g.rank == 0 else None, src=0)
        parts = [shard(total, g.world, r) for r in range(g.world)]
        a, ma = g.scatter_blobs(None, src=0, produce=(lambda r: src.export_ctxts(0, 0, *parts[r])) if g.rank == 0 else None)
        b, mb = g.scatter_blobs(None, src=0, produce=(lambda r: src.export_ctxts(0, 1, *parts[r])) if g.rank == 0 else None)
        first, count = parts[g.rank]
        w = host.Session(*P, count, lib_path=so, keys=keys, operands=(a, b))
        w.multiply(1, 1, True)
        prod = w.export_ctxts(1, 0)
        got, moved = g.gather_blobs(prod, dst=0)
        out = {{"rank": g.rank, "count": count, "scattered": ma + mb, "gathered": moved, "prod_bytes": int(prod.size)}}
        if g.rank == 0:
            out["verified"] = sum(src.verify_blob(blob, 1, *parts[r]) for r, blob in enumerate(got))
            out["sizes"] = [int(x.size) for x in got]
        else:
            assert got is None
            try:
                w.decrypt(1, 0)
                out["decrypted_without_secret"] = True
            except host.HostError:
                out["decrypted_without_secret"] = False
        print(json.dumps(out))
        g.close()
    """))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        o, e = pr.communicate(timeout=300)
        assert pr.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["count"] == 3 and outs[1]["count"] == 2
    assert outs[0]["verified"] == 5 and outs[1]["decrypted_without_secret"] is False
    assert outs[0]["sizes"][1] == outs[1]["prod_bytes"] == outs[1]["gathered"]
    assert outs[0]["scattered"] == outs[1]["scattered"] > 0        # what rank 0 sent is what rank 1 received
