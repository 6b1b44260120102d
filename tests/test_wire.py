"""HElib's DoubleCRT wire formats (helib_amd/wire.py) against the reference's own
binary fixture: every DoubleCRT block of tests/test_resources/iotest_binLE.bin (cut out by
tests/golden/make_golden.py) must parse, re-serialise to the same bytes, and carry the rows the
ASCII fixture of the same objects lists."""
import json
import os

import numpy as np
import pytest

from helib_amd import wire

HERE = os.path.dirname(os.path.abspath(__file__))
BLOCKS = json.load(open(os.path.join(HERE, "golden", "iotest_m12_bin_blocks.json")))["blocks"]
ASCII = json.load(open(os.path.join(HERE, "golden", "iotest_m12.json")))


def test_every_binary_block_round_trips_bytewise():
    assert len(BLOCKS) >= 19
    for blk in BLOCKS:
        raw = bytes.fromhex(blk["hex"])
        idx, rows, off = wire.read_rows(raw)
        assert off == len(raw) and rows.shape == (len(idx), 4)       # phi(12) = 4
        assert idx == list(range(len(idx)))
        for r, i in zip(rows, idx):
            assert r.max() < ASCII["primes"][i]
        assert wire.write_rows(idx, rows) == raw
        # row order given in any order is serialised ascending, like IndexSet iterates
        perm = list(reversed(range(len(idx))))
        assert wire.write_rows([idx[k] for k in perm], rows[perm]) == raw


def test_binary_blocks_carry_the_ascii_fixture_rows():
    parsed = [wire.read_rows(bytes.fromhex(b["hex"]))[:2] for b in BLOCKS]
    have = {(tuple(i), tuple(map(tuple, r.tolist()))) for i, r in parsed}

    def key(obj_idx, obj_rows):
        return (tuple(obj_idx), tuple(tuple(r) for r in obj_rows))
    want = [key(ASCII["pubkey_b"]["idx"], ASCII["pubkey_b"]["rows"]),
            key(ASCII["pubkey_a"]["idx"], ASCII["pubkey_a"]["rows"]),
            key(ASCII["seckey"]["idx"], ASCII["seckey"]["rows"])]
    for k in ASCII["ksw"]:
        want.append(key(k["b0_idx"], k["b0"]))
        want.append(key(k["b1_idx"], k["b1"]))
    for w in want:
        assert w in have


def test_json_form_and_validation():
    idx, rows, _ = wire.read_rows(bytes.fromhex(BLOCKS[-1]["hex"]))
    j = wire.to_json(idx, rows)
    assert j["set"] == idx and len(j["map"]) == len(idx)
    i2, r2 = wire.from_json(json.dumps(j), primes=ASCII["primes"], phim=4)
    assert i2 == idx and np.array_equal(r2, rows)
    bad = dict(j, map=[list(r) for r in j["map"]])
    bad["map"][0][0] = ASCII["primes"][idx[0]]                       # == q: out of range
    with pytest.raises(ValueError):
        wire.from_json(bad, primes=ASCII["primes"], phim=4)
    with pytest.raises(ValueError):
        wire.from_json(j, primes=ASCII["primes"], phim=8)


def test_32bit_words_and_errors():
    rows = np.arange(12, dtype=np.uint64).reshape(3, 4)
    raw = wire.write_rows([7, 2, 5], rows, int_size=wire.BIT32)
    idx, back, off = wire.read_rows(raw)
    assert idx == [2, 5, 7] and off == len(raw)
    assert np.array_equal(back, rows[[1, 2, 0]])
    with pytest.raises(ValueError):
        wire.write_rows([0], np.array([[1 << 40]], dtype=np.uint64), int_size=wire.BIT32)
    with pytest.raises(ValueError):
        wire.write_rows([0, 0], rows[:2])
    with pytest.raises(ValueError):
        wire.write_rows([0], rows[:1], int_size=3)


# ---------------------------------------------------------------- Ctxt and KeySwitch objects
OBJECTS = json.load(open(os.path.join(HERE, "golden", "iotest_m12_bin_objects.json")))["objects"]


def test_fixture_keyswitch_objects_round_trip_bytewise():
    """The four key-switching matrices of the reference's binary fixture (legacy layout = 2.2.0's
    minus KeySwitch::noiseBound): SKHandle, toKeyID, ptxtSpace, vector<DoubleCRT>, prgSeed (ZZ)."""
    kms = [o for o in OBJECTS if o["kind"] == "KM"]
    assert len(kms) == 8                                   # 4 in the public key, the same 4 in the secret key
    seen = []
    for o in kms:
        raw = bytes.fromhex(o["hex"])
        k, off = wire.read_keyswitch(raw, legacy=True)
        assert off == len(raw)
        assert wire.write_keyswitch(k, legacy=True) == raw
        assert k["toKeyID"] == 0 and k["ptxtSpace"] == 7 and len(k["b"]) == 2
        assert 0 < k["prgSeed"] < (1 << 256)               # RandomBits(prgSeed, 256)
        for idx, rows in k["b"]:
            assert idx == [0, 1, 2, 3, 4] and rows.shape == (5, 4)
        seen.append(k["fromKey"])
    # the handles the ASCII fixture lists: s^2, s^3, s(X^5), s(X^7) -> s
    assert seen[:4] == [(2, 1, 0), (3, 1, 0), (1, 5, 0), (1, 7, 0)] == seen[4:]
    # and their b columns are the rows of the ASCII fixture
    for k_, a in zip([wire.read_keyswitch(bytes.fromhex(o["hex"]), legacy=True)[0] for o in kms[:4]], ASCII["ksw"]):
        assert k_["b"][0][1].tolist() == a["b0"] and k_["b"][1][1].tolist() == a["b1"]
    # the 2.2.0 layout appends the noise bound; JSON form round-trips too
    k["noiseBound"] = 1234.5
    raw2 = wire.write_keyswitch(k)
    k2, off = wire.read_keyswitch(raw2)
    assert off == len(raw2) == len(raw) + 16 and k2["noiseBound"] == 1234.5 and k2["prgSeed"] == k["prgSeed"]
    j = wire.keyswitch_to_json(k)
    assert j["type"] == "KeySwitch" and j["content"]["prgSeed"]["number"] == str(k["prgSeed"])
    k3 = wire.keyswitch_from_json(json.dumps(j), primes=ASCII["primes"], phim=4)
    assert k3["fromKey"] == k["fromKey"] and all(np.array_equal(x[1], y[1]) for x, y in zip(k3["b"], k["b"]))


def test_fixture_ctxt_objects_round_trip_bytewise():
    """pubEncrKey (a 2-part ciphertext over primes 0-2) and the empty recryptEkey, as serialised in
    the reference's fixture (legacy layout = 2.2.0's minus header/intFactor/ptxtMag/ratFactor)."""
    cxs = [o for o in OBJECTS if o["kind"] == "CX"]
    assert len(cxs) == 4
    for o in cxs:
        raw = bytes.fromhex(o["hex"])
        c, off = wire.read_ctxt(raw, legacy=True)
        assert off == len(raw)
        assert wire.write_ctxt(c, legacy=True) == raw
    pk, _ = wire.read_ctxt(bytes.fromhex(cxs[0]["hex"]), legacy=True)
    assert pk["ptxtSpace"] == 7 and pk["primeSet"] == [0, 1, 2] and len(pk["parts"]) == 2
    assert pk["noiseBound"] == pytest.approx(2007.04)       # the xdouble encoding (mantissa, exponent 0)
    assert [p[2] for p in pk["parts"]] == [(0, 1, 0), (1, 1, 0)]          # handles "1" and "s"
    assert pk["parts"][0][1].tolist() == ASCII["pubkey_b"]["rows"]
    assert pk["parts"][1][1].tolist() == ASCII["pubkey_a"]["rows"]
    empty, _ = wire.read_ctxt(bytes.fromhex(cxs[1]["hex"]), legacy=True)   # recryptEkey: no parts
    assert empty["parts"] == [] and empty["primeSet"] == [0, 1, 2]


def test_ctxt_2_2_0_layout_and_json():
    pk, _ = wire.read_ctxt(bytes.fromhex([o for o in OBJECTS if o["kind"] == "CX"][0]["hex"]), legacy=True)
    pk["intFactor"] = 3
    pk["noiseBound"] = 3.0 * 2.0 ** 300                     # needs a non-zero xdouble exponent
    raw = wire.write_ctxt(pk)
    assert raw[:4] == b"|HE[" and raw[4:8] == bytes([0, 0, 1, 0]) and raw[8:12] == bytes([2, 2, 0, 0])
    assert raw[12] == 20 and raw[20:28] == b"]HE||CX[" and raw[-4:] == b"]CX|"
    m, e = wire.xdouble_parts(pk["noiseBound"])
    assert e == 3 and 2.0 ** -57 <= m < 2.0 ** 57 and m * 2.0 ** (114 * e) == pk["noiseBound"]
    back, off = wire.read_ctxt(raw)
    assert off == len(raw) and back["intFactor"] == 3 and back["noiseBound"] == pk["noiseBound"]
    assert all(np.array_equal(a[1], b[1]) and a[2] == b[2] for a, b in zip(back["parts"], pk["parts"]))
    with pytest.raises(ValueError):
        wire.read_ctxt(b"|HX[" + raw[4:])
    with pytest.raises(ValueError):
        wire.read_ctxt(raw[:12] + bytes([10]) + raw[13:])   # a PubKey header in front of a Ctxt
    j = wire.ctxt_to_json(pk)
    assert j["type"] == "Ctxt" and j["HElibVersion"] == "2.2.0" and j["serializationVersion"] == "0.0.1"
    assert set(j["content"]) == {"ptxtSpace", "noiseBound", "primeSet", "intFactor", "ptxtMag", "ratFactor", "parts"}
    back = wire.ctxt_from_json(json.dumps(j), primes=ASCII["primes"], phim=4)
    assert back["noiseBound"] == pk["noiseBound"] and back["primeSet"] == [0, 1, 2]
    bad = json.loads(json.dumps(j))
    bad["content"]["primeSet"] = [0, 1]
    with pytest.raises(ValueError):
        wire.ctxt_from_json(bad)
    bad = dict(j, serializationVersion="0.0.2")
    with pytest.raises(ValueError):
        wire.ctxt_from_json(bad)


def test_ctxt_objects_through_the_wire_still_decrypt():
    """helib_amd.keys ciphertexts (oracle backend) -> Ctxt::writeTo bytes -> back -> Decrypt, before
    and after a multiplication (3 parts incl. the s^2 handle, special primes in the prime set)."""
    from helib_amd import ctxt as hc, keys as hk
    from oracle import oracle as O
    from oracle.backend import OPoly, OracleBackend
    m, p = 128, 257
    cc = hc.ChainContext(m, p, 1, bits=150, c=3)
    octx = O.Ctx(m)
    for q in cc.primes:
        octx.add_prime(q)
    be = OracleBackend(octx, cc)
    sk = hk.SecKey(cc, be, 4)
    sk.GenSecKey()
    rng = np.random.default_rng(6)
    ma, mb = rng.integers(0, p, size=cc.phim), rng.integers(0, p, size=cc.phim)
    ca, cb = sk.Encrypt(ma), sk.Encrypt(mb)

    def through(ct):
        raw = wire.write_ctxt(wire.from_ctxt(ct))
        desc, off = wire.read_ctxt(raw)
        assert off == len(raw)
        relin = sk.getKeySWmatrix(2, 1)
        return wire.to_ctxt(desc, hc.Ctxt, cc, be.ops, lambda idx, rows: OPoly(octx, idx, rows),
                            ksw=relin.W, ksw_ptxtSpace=relin.ptxtSpace)
    ca2 = through(ca)
    assert sk.Decrypt(ca2) == [int(v) for v in ma] and abs(ca2.lnNoise - ca.lnNoise) < 1e-12
    ca2.multLowLvl(cb)
    cm = through(ca2)
    assert set(cm.parts) == {"1", "s", "s2"}
    cm.reLinearize()
    cm = through(cm)
    from tests import bgv_ref as B
    assert sk.Decrypt(cm) == [int(v) for v in B.polymul_mod_phi(ma, mb, m, p)]


# ---------------------------------------------------------------------------------------------
# containers: Context, PubKey, SecKey
# ---------------------------------------------------------------------------------------------
WHOLE = bytes.fromhex(json.load(open(os.path.join(HERE, "golden", "iotest_m12_bin_whole.json")))["hex"])


def test_whole_fixture_parses_and_round_trips_bytewise():
    """The reference's binary fixture end to end: context base + context, the public key, the
    secret key (which embeds the same public key) -- every byte consumed, every byte re-emitted."""
    c, off = wire.read_context(WHOLE, 0, legacy=True)
    assert (c["m"], c["p"], c["r"], c["gens"], c["ords"]) == (12, 7, 1, [5], [2])
    assert c["qs"] == ASCII["primes"] and c["specialPrimes"] == [3, 4] and c["digits"] == [[0, 1], [2]]
    assert c["stdev"] == 3.2
    pk, off2 = wire.read_pubkey(WHOLE, off, legacy=True)
    sk, off3 = wire.read_seckey(WHOLE, off2, legacy=True)
    assert off3 == len(WHOLE) == 6704
    assert wire.write_context(c, legacy=True) == WHOLE[:off]
    assert wire.write_pubkey(pk, legacy=True) == WHOLE[off:off2]
    assert wire.write_seckey(sk, legacy=True) == WHOLE[off2:]
    # the secret key file embeds the public key byte for byte
    assert WHOLE[off:off2] == WHOLE[off2 + 4:off2 + 4 + (off2 - off)]
    # PubKey::setKeySwitchMap's BFS reproduces the stored map from the stored matrices
    assert [w["fromKey"] for w in pk["keySwitching"]] == [(2, 1, 0), (3, 1, 0), (1, 5, 0), (1, 7, 0)]
    assert pk["keySwitchMap"] == [wire.key_switch_map(12, pk["keySwitching"])]
    assert pk["recryptKeyID"] == -1 and pk["recryptEkey"]["parts"] == []
    # the secret key rows are the ASCII fixture's
    (idx, rows), = sk["sKeys"]
    assert idx == ASCII["seckey"]["idx"] and rows.tolist() == ASCII["seckey"]["rows"]
    # and the public encryption key is an RLWE sample under it: b + a*s = small (p*e) modulo every prime
    from oracle import oracle as O
    octx = O.Ctx(12)
    for q in c["qs"]:
        octx.add_prime(q)
    (bi, b, _), (ai, a, _) = pk["pubEncrKey"]["parts"]
    acc = np.array([[(int(x) + int(y) * int(s)) % c["qs"][i] for x, y, s in zip(b[r], a[r], rows[idx.index(i)])]
                    for r, i in enumerate(bi)], dtype=np.uint64)
    e = octx.to_poly(bi, acc)
    assert all(int(v) % 7 == 0 for v in e) and [int(v) // 7 for v in e] == ASCII["expect_e_coeffs"]


def _keys(m=128, p=257, bits=150):
    from helib_amd import ctxt as hc, keys as hk
    from oracle import oracle as O
    from oracle.backend import OracleBackend
    cc = hc.ChainContext(m, p, 1, bits=bits, c=3)
    octx = O.Ctx(m)
    for q in cc.primes:
        octx.add_prime(q)
    be = OracleBackend(octx, cc)
    sk = hk.SecKey(cc, be, 9)
    sk.GenSecKey()
    sk.GenKeySWmatrix(1, 3)
    sk.GenKeySWmatrix(1, 5)
    sk.setKeySwitchMap()
    return cc, octx, be, sk


def test_keys_with_explicit_a_columns_are_not_written_as_reference_blobs_by_default():
    """The a-columns of a key-switching matrix are not on the reference's wire (KeySwitch::writeTo,
    src/keySwitching.cpp:196-240, stores prgSeed and HElib re-derives them with NTL's PRG, src/Ctxt.cpp:
    196-206); this engine's keys hold them explicitly.  A well-formed reference blob of such a key would
    load into HElib without complaint and key-switch to garbage, so the writers refuse it unless the
    caller says engine_only=True; sk_only blobs (no matrices) and blobs read from the wire are not
    affected."""
    cc, octx, be, sk = _keys()
    d = wire.from_seckey(sk)
    assert all(w["explicit_a"] for w in d["keySwitching"])
    for call in (lambda: wire.write_pubkey(d), lambda: wire.write_seckey(d), lambda: wire.pubkey_to_json(d),
                 lambda: wire.seckey_to_json(d), lambda: wire.write_keyswitch(d["keySwitching"][0]),
                 lambda: wire.keyswitch_to_json(d["keySwitching"][0])):
        with pytest.raises(ValueError, match="engine_only=True"):
            call()
    assert wire.write_seckey(d, sk_only=True)                       # no matrices inside
    raw = wire.write_seckey(d, engine_only=True)
    back, _ = wire.read_seckey(raw)
    assert wire.write_seckey(back) == raw                           # what came off the wire goes back on it


def test_context_pubkey_seckey_2_2_0_layout_round_trip():
    cc, octx, be, sk = _keys()
    d = wire.from_seckey(sk, gens=[3, 127], ords=[32, -2])
    # Context
    raw = wire.write_context(d["context"])
    assert raw[:4] == b"|HE[" and raw[12] == 5 and raw[24:28] == b"|CN[" and raw[-4:] == b"]CN|"
    c2, off = wire.read_context(raw)
    assert off == len(raw) and wire.write_context(c2) == raw
    assert c2["qs"] == [int(q) for q in cc.primes] and c2["ords"] == [32, -2] and c2["scale"] == 10.0
    assert wire.context_from_json(json.dumps(wire.context_to_json(d["context"]))) == c2
    # PubKey / SecKey, binary
    for writer, reader, sid in ((wire.write_pubkey, wire.read_pubkey, 10), (wire.write_seckey, wire.read_seckey, 15)):
        raw = writer(d, engine_only=True)
        assert raw[12] == sid
        k2, off = reader(raw, context=d["context"])
        assert off == len(raw) and writer(k2) == raw
        assert k2["skBounds"] == [float(b) for b in sk.skBounds]
        assert [w["fromKey"] for w in k2["keySwitching"]] == [(2, 1, 0), (3, 1, 0), (1, 3, 0), (1, 5, 0)]
        assert k2["keySwitchMap"] == [wire.key_switch_map(cc.m, k2["keySwitching"])]
        # the stored map (matrix indices) and helib_amd.keys' map (powers of X) say the same thing
        for k in range(2, cc.m):
            i = k2["keySwitchMap"][0][k]
            assert (k2["keySwitching"][i]["fromKey"][1] if i >= 0 else 0) == sk.keySwitchMap[k]
        with pytest.raises(ValueError, match="Context mismatch"):
            reader(raw, context=dict(d["context"], m=64))
        with pytest.raises(ValueError, match="header"):
            reader(b"|XX[" + raw[4:])
        with pytest.raises(ValueError, match="structId"):
            reader(raw[:12] + bytes([20]) + raw[13:])
        bad = raw[:-4] + b"]XX|"
        with pytest.raises(ValueError, match="eye catcher"):
            reader(bad)
    # sk_only: context + secret rows only (src/keys.cpp:1741-1750)
    raw = wire.write_seckey(d, sk_only=True)
    k3, off = wire.read_seckey(raw, sk_only=True, context=d["context"])
    assert off == len(raw) and set(k3) == {"context", "sKeys"} and len(raw) < 40000
    assert np.array_equal(k3["sKeys"][0][1], d["sKeys"][0][1])
    # JSON
    j = json.dumps(wire.seckey_to_json(d, engine_only=True))
    k4 = wire.seckey_from_json(j, primes={i: q for i, q in enumerate(cc.primes)}, phim=cc.phim)
    assert wire.write_seckey(k4) == wire.write_seckey(d, engine_only=True)
    jj = json.loads(j)
    assert jj["type"] == "SecKey" and jj["content"]["PubKey"]["type"] == "PubKey"
    assert jj["content"]["PubKey"]["content"]["recryptEkey"] == "nullptr"


def test_keys_through_the_wire_still_work():
    """SecKey -> bytes -> SecKey: the reloaded key decrypts what the original encrypted, the
    original decrypts what the reloaded public key encrypts, and the reloaded relinearisation /
    rotation matrices give the right products and rotations."""
    from helib_amd import keys as hk
    from oracle.backend import OPoly
    from tests import bgv_ref as B
    cc, octx, be, sk = _keys()
    p, m = cc.p, cc.m
    raw = wire.write_seckey(wire.from_seckey(sk), engine_only=True)
    desc, _ = wire.read_seckey(raw)
    a_cols = {k: w.a for k, w in sk.keySwitching.items()}
    sk2 = wire.to_seckey(desc, hk.SecKey, cc, be, lambda idx, rows: OPoly(octx, idx, rows), ksw_a=a_cols, seed=77)
    assert np.array_equal(sk2.sKeys[0], sk.sKeys[0]) and sk2.skBounds == sk.skBounds
    assert sk2.keySwitchMap == sk.keySwitchMap
    rng = np.random.default_rng(12)
    ma, mb = rng.integers(0, p, size=cc.phim), rng.integers(0, p, size=cc.phim)
    assert sk2.Decrypt(sk.Encrypt(ma)) == [int(v) for v in ma]
    ca, cb = sk2.Encrypt(ma), sk2.Encrypt(mb)
    assert sk.Decrypt(ca) == [int(v) for v in ma]
    ca.multiplyBy(cb)
    want = [int(v) for v in B.polymul_mod_phi(ma, mb, m, p)]
    assert sk.Decrypt(ca) == want
    ca.smartAutomorph(15)       # 3 * 5: two steps along the reloaded map
    assert sk.Decrypt(ca) == [int(v) for v in B.automorph_mod_phi(want, m, 15, p)]
    # without the a-columns the matrices are not installed (they are not derivable here: NTL's PRG)
    sk3 = wire.to_seckey(desc, hk.SecKey, cc, be, lambda idx, rows: OPoly(octx, idx, rows))
    assert sk3.keySwitching == {} and sk3.Decrypt(sk.Encrypt(mb)) == [int(v) for v in mb]
    with pytest.raises(ValueError, match="Context mismatch"):
        wire.to_seckey(dict(desc, context=dict(desc["context"], m=64)), hk.SecKey, cc, be, None)


# ---------------------------------------------------------------------------------------------
# the same formats from C++ (include/helib_amd_wire.hpp), no device involved
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def wire_exe(tmp_path_factory):
    import subprocess
    root = os.path.dirname(HERE)
    exe = str(tmp_path_factory.mktemp("wire") / "wire_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "wire_test.cpp"), "-o", exe])

    def run(mode, blob, tmp=exe + ".bin"):
        with open(tmp, "wb") as f:
            f.write(blob)
        r = subprocess.run([exe, mode, tmp], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        return json.loads(r.stdout)
    return run


def test_cpp_wire_header_on_the_reference_fixture(wire_exe):
    """helib_amd_wire.hpp reads the reference's whole binary fixture to the last byte, writes the
    same 6704 bytes back, and sees what helib_amd/wire.py sees."""
    got = wire_exe("legacy", WHOLE)
    assert got["consumed"] and got["same_bytes"] and got["embedded_pk_equal"]
    c, off = wire.read_context(WHOLE, 0, legacy=True)
    pk, off2 = wire.read_pubkey(WHOLE, off, legacy=True)
    assert got["offsets"] == [off, off2, len(WHOLE)]
    assert (got["context"]["m"], got["context"]["p"], got["context"]["gens"], got["context"]["ords"]) == \
        (12, 7, [5], [2])
    assert [tuple(h) for h in got["handles"]] == [w["fromKey"] for w in pk["keySwitching"]]
    assert got["keySwitchMap"] == pk["keySwitchMap"][0] == got["keySwitchMap_bfs"]
    assert got["KS_strategy"] == pk["KS_strategy"] and got["skBounds"] == [64.0] and got["recryptKeyID"] == -1
    assert got["pubEncrKey_primeSet"] == pk["pubEncrKey"]["primeSet"] and got["pubEncrKey_parts"] == 2
    assert got["pubEncrKey_noise"] == [pk["pubEncrKey"]["noiseBound"], 0]
    assert got["sk_idx"] == ASCII["seckey"]["idx"] and got["sk_row0"] == ASCII["seckey"]["rows"][0]


def test_cpp_wire_header_2_2_0_layout_against_the_python_writer(wire_exe):
    """Blobs written by helib_amd/wire.py in the 2.2.0 layout (Context, PubKey, SecKey, sk_only,
    Ctxt incl. a CKKS one with a non-trivial ratFactor) are consumed and reproduced byte for byte
    by the C++ reader/writer; header / eye-catcher / truncation / context errors are raised."""
    cc, octx, be, sk = _keys()
    d = wire.from_seckey(sk)
    for mode, blob in (("context", wire.write_context(d["context"])), ("pubkey", wire.write_pubkey(d, engine_only=True)),
                       ("seckey", wire.write_seckey(d, engine_only=True)), ("skonly", wire.write_seckey(d, sk_only=True))):
        got = wire_exe(mode, blob)
        assert got["consumed"] and got["same_bytes"], mode
        assert got["context"]["qs"] == [int(q) for q in cc.primes] and got["context"]["gens"] == d["context"]["gens"]
        assert got["context"]["specialPrimes"] == list(cc.specialPrimes) and got["context"]["ndigits"] == len(cc.digits)
        if mode in ("pubkey", "seckey"):
            assert [tuple(h) for h in got["handles"]] == [w["fromKey"] for w in d["keySwitching"]]
            assert got["keySwitchMap"] == d["keySwitchMap"][0] == got["keySwitchMap_bfs"]
            assert got["skBounds"] == d["skBounds"]
        if mode in ("seckey", "skonly"):
            assert got["nsk"] == 1
    rng = np.random.default_rng(1)
    ca, cb = sk.Encrypt(rng.integers(0, 257, size=cc.phim)), sk.Encrypt(rng.integers(0, 257, size=cc.phim))
    ca.multiplyBy(cb)
    got = wire_exe("ctxt", wire.write_ctxt(wire.from_ctxt(ca)))
    assert got["consumed"] and got["same_bytes"] and got["poly_round_trip"]
    assert got["primeSet"] == sorted(ca.primeSet) and got["intFactor"] == ca.intFactor and got["parts"] == 2
    # a CKKS ciphertext: ratFactor far outside a double's range travels as (mantissa, exponent)
    from helib_amd import ctxt as hc, keys as hk
    from oracle import oracle as O
    from oracle.backend import OracleBackend
    ck = hc.ChainContext(128, -1, 20, bits=250, c=2, ckks=True)
    oc = O.Ctx(128)
    for q in ck.primes:
        oc.add_prime(q)
    sk2 = hk.SecKey(ck, OracleBackend(oc, ck), 3)
    sk2.GenSecKey()
    f = float(1 << 20)
    pt = np.rint(rng.uniform(-1, 1, ck.phim) / ck.phim * f).astype(np.int64)
    c1, c2 = sk2.CKKSencrypt(pt, 1.0, f), sk2.CKKSencrypt(pt, 1.0, f)
    c1.multiplyBy(c2)
    desc = wire.from_ctxt(c1)
    got = wire_exe("ctxt", wire.write_ctxt(desc))
    assert got["consumed"] and got["same_bytes"] and got["ptxtSpace"] == 1
    assert got["ratFactor"] == [desc["ratFactor"][0], desc["ratFactor"][1]] and got["ratFactor"][1] >= 1
    assert all(wire_exe("errors", wire.write_seckey(d, engine_only=True))["errors"])


def test_cpp_and_python_wire_agree_on_random_objects(wire_exe):
    """Random ciphertext descriptions (ragged prime sets, 1-3 parts, xdouble fields across many
    exponents, negative handles excluded as in the reference) -> python writer -> C++ reader/writer:
    the same bytes come back, for 2.2.0 Ctxt objects embedded in nothing and inside a SecKey."""
    rng = np.random.default_rng(2024)
    for trial in range(12):
        n = int(rng.choice([1, 4, 16]))
        nprimes = int(rng.integers(1, 7))
        idx = sorted(rng.choice(40, size=nprimes, replace=False).tolist())
        parts = []
        for h in [(0, 1, 0), (1, 1, 0), (int(rng.integers(1, 4)), int(rng.integers(2, 50)), 0)][:int(rng.integers(1, 4))]:
            rows = rng.integers(0, 1 << 60, size=(nprimes, n), dtype=np.uint64)
            parts.append((idx, rows, h))
        xd = lambda: (float(rng.uniform(1.0, 2.0 ** 56)), int(rng.integers(-3, 9)))   # noqa: E731
        desc = {"ptxtSpace": int(rng.choice([1, 2, 257, 65537])), "intFactor": int(rng.integers(1, 200)),
                "ptxtMag": xd(), "ratFactor": xd(), "noiseBound": xd(), "primeSet": idx, "parts": parts}
        blob = wire.write_ctxt(desc)
        got = wire_exe("ctxt", blob)
        assert got["consumed"] and got["same_bytes"] and got["primeSet"] == idx and got["parts"] == len(parts)
        assert got["ratFactor"] == list(desc["ratFactor"]) and got["poly_round_trip"]
        back, off = wire.read_ctxt(blob)
        assert off == len(blob) and wire.write_ctxt(back) == blob


# ---------------------------------------------------------------------------------------------
# the JSON forms from C++ (include/helib_amd_json.hpp)
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def json_exe(tmp_path_factory):
    import subprocess
    root = os.path.dirname(HERE)
    d = tmp_path_factory.mktemp("json")
    exe = str(d / "json_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "json_test.cpp"), "-o", exe])

    def run(*args, data=None):
        tmp = str(d / "in.dat")
        if data is not None:
            with open(tmp, "wb") as f:
                f.write(data if isinstance(data, bytes) else data.encode())
        argv = [exe] + [tmp if a == "@" else str(d / a) if a.startswith("out") else a for a in args]
        r = subprocess.run(argv, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout[-400:] + r.stderr
        return r.stdout, d
    return run


def test_cpp_json_forms_agree_with_the_python_side(json_exe):
    """helib_amd_json.hpp: every object kind of the path written as JSON by helib_amd/wire.py is parsed in C++
    into the description the binary reader produces -- the C++ binary writer gives the python writer's bytes --
    and written back as JSON equal to the original; the other way round, binary blobs read in C++ come out as
    the JSON python writes for them.  xdouble fields, the 256-bit prgSeed as a decimal string, the typed
    wrapper and the "nullptr" recryption key included."""
    cc, octx, be, sk = _keys()
    d = wire.from_seckey(sk)
    rng = np.random.default_rng(5)
    ca, cb = sk.Encrypt(rng.integers(0, 257, size=cc.phim)), sk.Encrypt(rng.integers(0, 257, size=cc.phim))
    ca.multiplyBy(cb)
    ct = wire.from_ctxt(ca)
    ksw = dict(d["keySwitching"][0], prgSeed=(1 << 255) + 12345678901234567890)     # a reference-style seed
    ksw.pop("explicit_a", None)
    cases = [("ctxt", wire.ctxt_to_json(ct), wire.write_ctxt(ct)),
             ("keyswitch", wire.keyswitch_to_json(ksw), wire.write_keyswitch(ksw)),
             ("context", wire.context_to_json(d["context"]), wire.write_context(d["context"])),
             ("pubkey", wire.pubkey_to_json(d, engine_only=True), wire.write_pubkey(d, engine_only=True)),
             ("seckey", wire.seckey_to_json(d, engine_only=True), wire.write_seckey(d, engine_only=True)),
             ("skonly", wire.seckey_to_json(d, sk_only=True), wire.write_seckey(d, sk_only=True))]
    for kind, j, blob in cases:
        out, tmpd = json_exe(kind, "j2b", "@", "out.bin", data=json.dumps(j))
        assert json.loads(out) == j, kind
        assert open(str(tmpd / "out.bin"), "rb").read() == blob, kind
        out, _ = json_exe(kind, "b2j", "@", data=blob)
        assert json.loads(out) == j, kind
    # a CKKS ciphertext: ratFactor as (mantissa, exponent)
    from helib_amd import ctxt as hc, keys as hk
    from oracle import oracle as O
    from oracle.backend import OracleBackend
    ck = hc.ChainContext(128, -1, 20, bits=250, c=2, ckks=True)
    oc = O.Ctx(128)
    for q in ck.primes:
        oc.add_prime(q)
    sk2 = hk.SecKey(ck, OracleBackend(oc, ck), 3)
    sk2.GenSecKey()
    f = float(1 << 20)
    pt = np.rint(rng.uniform(-1, 1, ck.phim) / ck.phim * f).astype(np.int64)
    c1, c2 = sk2.CKKSencrypt(pt, 1.0, f), sk2.CKKSencrypt(pt, 1.0, f)
    c1.multiplyBy(c2)
    ct2 = wire.from_ctxt(c1)
    j2 = wire.ctxt_to_json(ct2)
    out, tmpd = json_exe("ctxt", "j2b", "@", "out.bin", data=json.dumps(j2))
    assert json.loads(out) == j2 and j2["content"]["ratFactor"]["exponent"] >= 1
    assert open(str(tmpd / "out.bin"), "rb").read() == wire.write_ctxt(ct2)
    # the reference's own (legacy-layout) fixture objects: read by python, handed to C++ as 2.2.0 JSON
    pk, _ = wire.read_pubkey(WHOLE, wire.read_context(WHOLE, 0, legacy=True)[1], legacy=True)
    for w in pk["keySwitching"]:
        jw = wire.keyswitch_to_json(w)
        out, _ = json_exe("keyswitch", "j2b", "@", "out.bin", data=json.dumps(jw))
        assert json.loads(out) == jw and jw["content"]["prgSeed"]["number"] == str(w["prgSeed"])
    out, _ = json_exe("errors")
    assert out.strip() == "raised 25 of 25"
