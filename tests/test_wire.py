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
