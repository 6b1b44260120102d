"""HElib 2.2.0 wire formats of a DoubleCRT -- host side only.

binary  DoubleCRT::writeTo / read (src/DoubleCRT.cpp:1530-1561):
          IndexSet::writeTo (src/IndexSet.cpp:288-297): int64 card, then the indices ascending
          per index: write_ntl_vec_long (src/binio.cpp:103-146): int32 length, int32 intSize
          (8 or 4), then `length` little-endian words of that size
JSON    DoubleCRT::writeToJSON / readJSON (src/DoubleCRT.cpp:1568-1633):
          {"set": [indices], "map": [[row], ...]} with every entry in [0, ithPrime(i))

Rows cross this boundary as numpy uint64 [nrows, phi(m)] in the order of the index set (ascending
prime index, as IndexSet iterates); helib_amd.capi.DoubleCRT keeps its rows in getIndexSet() order,
so readers/writers below permute where needed.  A ciphertext produced by a real HElib build can be
loaded this way (and vice versa) -- the route to cross-implementation parity once somebody has an
NTL build at hand.
"""
import json
import struct

import numpy as np

BIT32, BIT64 = 4, 8      # Binio::BIT32 / BIT64 (bytes)


def write_rows(idx, rows, int_size=BIT64):
    """-> bytes.  idx: prime indices (any order); rows: [len(idx), N] residues."""
    idx = [int(i) for i in idx]
    rows = np.asarray(rows, dtype=np.uint64)
    if rows.shape[0] != len(idx) or len(set(idx)) != len(idx):
        raise ValueError("one row per distinct prime index")
    if int_size not in (BIT32, BIT64):
        raise ValueError("intSize must be 32 or 64 bit for binary IO")
    order = np.argsort(idx)
    out = [struct.pack("<q", len(idx))]
    out += [struct.pack("<q", idx[k]) for k in order]
    for k in order:
        r = rows[k]
        out.append(struct.pack("<ii", r.shape[0], int_size))
        if int_size == BIT64:
            out.append(r.astype("<u8").tobytes())
        else:
            if r.max(initial=0) >= (1 << 31):
                raise ValueError("row does not fit 32-bit words")
            out.append(r.astype("<u4").tobytes())
    return b"".join(out)


def read_rows(buf, off=0):
    """-> (idx ascending, rows uint64 [n, N], offset after the object)."""
    (card,) = struct.unpack_from("<q", buf, off)
    off += 8
    if card < 0 or card > 4096:
        raise ValueError("implausible IndexSet cardinality")
    idx = list(struct.unpack_from(f"<{card}q", buf, off))
    off += 8 * card
    idx_sorted = sorted(set(idx))              # IndexSet::insert: a set, iterated ascending
    rows = []
    for _ in idx_sorted:
        n, int_size = struct.unpack_from("<ii", buf, off)
        off += 8
        if int_size not in (BIT32, BIT64) or n < 0:
            raise ValueError("intSize must be 32 or 64 bit for binary IO")
        dt = "<u8" if int_size == BIT64 else "<u4"
        rows.append(np.frombuffer(buf, dtype=dt, count=n, offset=off).astype(np.uint64))
        off += n * int_size
    if rows and any(r.shape != rows[0].shape for r in rows):
        raise ValueError("rows of unequal length")
    return idx_sorted, (np.stack(rows) if rows else np.zeros((0, 0), dtype=np.uint64)), off


def to_json(idx, rows):
    idx = [int(i) for i in idx]
    order = np.argsort(idx)
    rows = np.asarray(rows, dtype=np.uint64)
    return {"set": [idx[k] for k in order], "map": [[int(v) for v in rows[k]] for k in order]}


def from_json(j, primes=None, phim=None):
    """-> (idx, rows); with `primes` (index -> q) and `phim` the reference's validity checks
    (src/DoubleCRT.cpp:1616-1631) are applied."""
    if isinstance(j, (str, bytes)):
        j = json.loads(j)
    idx = sorted(set(int(i) for i in j["set"]))
    rows = np.array(j["map"], dtype=np.uint64).reshape(len(idx), -1)
    if phim is not None and rows.shape[1] != phim:
        raise ValueError("Data not valid: d.map[i].length() != phim")
    if primes is not None:
        for r, i in zip(rows, idx):
            if i >= len(primes):
                raise ValueError("Stream does not contain subset of the context's primes")
            if r.max(initial=0) >= primes[i]:
                raise ValueError("this->map[i][j] invalid: must be between 0 and context.ithPrime(i)")
    return idx, rows


# ---- DoubleCRT objects of the GPU backend -------------------------------------------------
def writeTo(poly, b=0, int_size=BIT64):
    """DoubleCRT::writeTo for batch element b of a capi.DoubleCRT."""
    return write_rows(poly.getIndexSet(), poly.download()[:, b], int_size)


def readFrom(capi, context, buf, off=0, batch=1):
    """DoubleCRT::readFrom: a new capi.DoubleCRT (the same rows in every batch element) and the
    offset after it."""
    idx, rows, off = read_rows(buf, off)
    for i in idx:
        if i >= len(context.primes):
            raise ValueError("Stream does not contain subset of the context's primes")
    data = np.repeat(rows[:, None, :], batch, axis=1)
    return capi.DoubleCRT(context, idx, batch, data), off
