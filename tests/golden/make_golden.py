#!/usr/bin/env python3
"""Extract the known-answer vectors of the reference's legacy I/O fixture.

Source (read-only, reference-authored DATA, not code):
  /root/reference/tests/test_resources/iotest_asciiLE.txt
It is the only value-level DoubleCRT fixture in the HElib 2.2.0 tree
(SURVEY.md section 0 fact 9): context m=12, p=7, five primes; a public
encryption key (b, a) on primes 0-2 and the secret key on primes 0-4.

Run once in the build container (the GPU box has no /root/reference):
    python tests/golden/make_golden.py
writes tests/golden/iotest_m12.json, which is committed.
"""
import json
import os
import re
import sys

SRC = "/root/reference/tests/test_resources/iotest_asciiLE.txt"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "iotest_m12.json")


def rows_after(lines, start):
    """Parse '[[idx...]\n [row]\n [row] ... ]' starting at line `start` (0-based)."""
    idx = [int(t) for t in re.findall(r"-?\d+", lines[start])]
    rows = []
    i = start + 1
    while lines[i].strip().startswith("[") and not lines[i].strip().startswith("[["):
        rows.append([int(t) for t in re.findall(r"\d+", lines[i])])
        i += 1
    return idx, rows


def main():
    if not os.path.exists(SRC):
        sys.exit("reference fixture not present (expected in the build container only)")
    lines = open(SRC).read().split("\n")
    hdr = [int(t) for t in re.findall(r"\d+", lines[0])]
    m, p, r = hdr[0], hdr[1], hdr[2]
    primes = [int(t) for t in lines[4].split()]
    # public encryption key: part b (handle one) at line 12, part a (handle s) at line 18
    idx_b, rows_b = rows_after(lines, 11)
    idx_a, rows_a = rows_after(lines, 17)
    # secret key DoubleCRT at line 181 (1-based)
    idx_s, rows_s = rows_after(lines, 180)
    # four key-switching matrices (fromKey handle, two b columns each) -- kept for
    # layout tests only: the matching a-columns come from NTL's PRG and are not
    # reproducible without NTL.
    ksw = []
    for ln in (27, 43, 59, 75):
        handle = [int(t) for t in re.findall(r"-?\d+", lines[ln - 1])]
        i0, b0 = rows_after(lines, ln)
        i1, b1 = rows_after(lines, ln + 7)
        ksw.append({"header": handle, "b0_idx": i0, "b0": b0, "b1_idx": i1, "b1": b1})
    out = {
        "source": "HElib 2.2.0 tests/test_resources/iotest_asciiLE.txt",
        "m": m, "p": p, "r": r, "primes": primes,
        "pubkey_b": {"idx": idx_b, "rows": rows_b},
        "pubkey_a": {"idx": idx_a, "rows": rows_a},
        "seckey": {"idx": idx_s, "rows": rows_s},
        "ksw": ksw,
        # facts verified when the fixture was analysed:
        "expect_s_coeffs": [1, -1, 1, 1],
        "expect_e_coeffs": [-2, -4, 0, 1],
    }
    with open(DST, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", DST)
    bin_blocks()
    bin_objects()
    bin_whole()


def bin_whole():
    """The whole binary fixture (6704 bytes: context base + context + public key + secret key in
    the older layout), for the container readers/writers of helib_amd/wire.py."""
    src = os.path.join(os.path.dirname(SRC), "iotest_binLE.bin")
    b = open(src, "rb").read()
    dst = os.path.join(os.path.dirname(DST), "iotest_m12_bin_whole.json")
    with open(dst, "w") as f:
        json.dump({"source": "HElib 2.2.0 tests/test_resources/iotest_binLE.bin", "bytes": len(b),
                   "hex": b.hex()}, f)
    print("wrote", dst, len(b), "bytes")


def bin_objects():
    """Whole serialised objects of the binary fixture, cut between their eye catchers: the four
    key-switching matrices ("|KM[" .. "]KM|") and the ciphertext-typed members ("|CX[" .. "]CX|":
    pubEncrKey and the empty recryptEkey of the public and of the secret key).  The file was
    written by an older HElib: same field encodings as 2.2.0 (SKHandle, IndexSet, vec_long rows,
    ZZ, xdouble), without the 2.2.0 additions (SerializeHeader, intFactor/ptxtMag/ratFactor,
    KeySwitch::noiseBound) -- helib_amd/wire.py reads it with legacy=True."""
    src = os.path.join(os.path.dirname(SRC), "iotest_binLE.bin")
    b = open(src, "rb").read()
    objs = []
    for tag in (b"KM", b"CX"):
        beg, end = b"|" + tag + b"[", b"]" + tag + b"|"
        pos = 0
        while True:
            i = b.find(beg, pos)
            if i < 0:
                break
            j = b.index(end, i) + 4
            objs.append({"kind": tag.decode(), "offset": i, "hex": b[i:j].hex()})
            pos = j
    dst = os.path.join(os.path.dirname(DST), "iotest_m12_bin_objects.json")
    with open(dst, "w") as f:
        json.dump({"source": "HElib 2.2.0 tests/test_resources/iotest_binLE.bin", "objects": objs}, f, indent=1)
    print("wrote", dst, len(objs), "objects")


def bin_blocks():
    """The same fixture in HElib's binary format (iotest_binLE.bin): every DoubleCRT block
    (DoubleCRT::writeTo: IndexSet then one vec_long per prime) cut out byte-for-byte, for the
    wire-format tests of helib_amd/wire.py."""
    import struct
    src = os.path.join(os.path.dirname(SRC), "iotest_binLE.bin")
    b = open(src, "rb").read()
    blocks = []
    for card in (3, 5):
        pat = struct.pack(f"<{card + 1}q", card, *range(card))
        i = b.find(pat)
        while i >= 0:
            end = i + 8 * (card + 1)
            ok = True
            for _ in range(card):
                n, sz = struct.unpack_from("<ii", b, end)
                if n != 4 or sz != 8:
                    ok = False
                    break
                end += 8 + n * sz
            if ok:
                blocks.append({"offset": i, "hex": b[i:end].hex()})
            i = b.find(pat, i + 1)
    blocks.sort(key=lambda x: x["offset"])
    dst = os.path.join(os.path.dirname(DST), "iotest_m12_bin_blocks.json")
    with open(dst, "w") as f:
        json.dump({"source": "HElib 2.2.0 tests/test_resources/iotest_binLE.bin", "blocks": blocks}, f, indent=1)
    print("wrote", dst, len(blocks), "blocks")


if __name__ == "__main__":
    main()
