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
