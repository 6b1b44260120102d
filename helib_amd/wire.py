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


# =============================================================================================
# Ctxt and KeySwitch objects (HElib 2.2.0)
#
# binary  SerializeHeader<T> (src/binio.h:88-146, 24 bytes): "|HE[" version{0,0,1,0}
#           helibVersion{2,2,0,0} structId (Ctxt = 20) 7 reserved zero bytes "]HE|"
#         Ctxt::writeTo (src/Ctxt.cpp:2584-2608): header, "|CX[", ptxtSpace, intFactor (raw
#           int64), ptxtMag, ratFactor, noiseBound (write_raw_xdouble, src/binio.cpp:165-171:
#           double mantissa + int64 exponent), primeSet (IndexSet), int64 #parts, per part
#           CtxtPart::writeTo (:2715-2719) = DoubleCRT::writeTo + SKHandle::writeTo
#           (powerOfS, powerOfX, secretKeyID as int64, :49-54), "]CX|"
#         KeySwitch::writeTo (src/keySwitching.cpp:196-219; no header): "|KM[", fromKey
#           (SKHandle), toKeyID, ptxtSpace, int64 #b, the b DoubleCRTs, prgSeed (write_raw_ZZ,
#           src/binio.cpp:180-191: int64 byte count + little-endian magnitude bytes), noiseBound
#           (xdouble), "]KM|"
# JSON    Ctxt::writeToJSON (src/Ctxt.cpp:2642-2660) / KeySwitch::writeToJSON
#           (src/keySwitching.cpp:243-262) inside toTypedJson (src/io.h:161-167):
#           {"type", "HElibVersion": "2.2.0", "serializationVersion": "0.0.1", "content": {...}},
#           xdouble = {"mantissa", "exponent"}, ZZ = {"number": "<decimal>"} (src/io.cpp:20-44)
#
# A Ctxt here is the plain description {ptxtSpace, intFactor, ptxtMag, ratFactor, noiseBound,
# primeSet, parts: [(idx, rows, (powerOfS, powerOfX, secretKeyID))]}; from_ctxt / to_ctxt convert
# from / to helib_amd.ctxt.Ctxt (handles "1", "s", "s2", ("s", k)).
# =============================================================================================
HELIB_VERSION = (2, 2, 0)
STRUCT_ID = {"Context": 5, "PubKey": 10, "SecKey": 15, "Ctxt": 20}
JSON_SERIALIZATION_VERSION = "0.0.1"
_XD_HBOUND = float(2 ** 57)        # NTL_XD_HBOUND = NTL_FDOUBLE_PRECISION * 32 = 2^52 * 32
_XD_BOUND_BITS = 114               # NTL_XD_BOUND = NTL_XD_HBOUND^2


def xdouble_parts(v):
    """NTL::xdouble normal form of a non-negative python float or (mantissa, exponent) pair:
    value = x * 2^(114 e) with 2^-57 <= |x| < 2^57 (NTL xdouble::normalize)."""
    if isinstance(v, tuple):
        return float(v[0]), int(v[1])
    x, e = float(v), 0
    if x == 0.0:
        return 0.0, 0
    while abs(x) >= _XD_HBOUND:
        x /= 2.0 ** _XD_BOUND_BITS
        e += 1
    while abs(x) < 1.0 / _XD_HBOUND:
        x *= 2.0 ** _XD_BOUND_BITS
        e -= 1
    return x, e


def _xd_from_ln(ln):
    """exp(ln) as an xdouble (mantissa, exponent) pair without overflowing a double"""
    import math
    if ln == -math.inf:
        return (0.0, 0)
    step, half, e = _XD_BOUND_BITS * math.log(2.0), 57 * math.log(2.0), 0
    while ln >= half:
        ln -= step
        e += 1
    while ln < -half:
        ln += step
        e -= 1
    return (math.exp(ln), e)


def xdouble_value(m, e):
    return m * 2.0 ** (_XD_BOUND_BITS * e) if abs(e) < 8 else (m, e)


def header(kind):
    return (b"|HE[" + bytes([0, 0, 1, 0]) + bytes(HELIB_VERSION + (0,)) + bytes([STRUCT_ID[kind]])
            + bytes(7) + b"]HE|")


def _read_header(buf, off, kind):
    h = buf[off:off + 24]
    if h[:4] != b"|HE[" or h[20:24] != b"]HE|":
        raise ValueError("Eye catchers for header mismatch")
    if h[4:8] != bytes([0, 0, 1, 0]):
        raise ValueError("Header: version not supported")
    if h[12] != STRUCT_ID[kind]:
        raise ValueError("Header: wrong structId")
    return off + 24


def _xd(v):
    m, e = xdouble_parts(v)
    return struct.pack("<dq", m, e)


def _handle_of(h):
    if h == "1":
        return (0, 1, 0)
    if h == "s":
        return (1, 1, 0)
    if h == "s2":
        return (2, 1, 0)
    return (1, int(h[1]), 0)


def _handle_name(sp, xp):
    if sp == 0:
        return "1"
    if xp == 1:
        return {1: "s", 2: "s2"}.get(sp, ("s^", sp))
    return ("s", xp) if sp == 1 else ("s^", sp, xp)


def write_ctxt(c, legacy=False):
    """legacy=True: the layout of the reference's own fixture tests/test_resources/iotest_bin*.bin
    (written by an older HElib): no SerializeHeader, no intFactor / ptxtMag / ratFactor."""
    if legacy:
        out = [b"|CX[", struct.pack("<q", int(c["ptxtSpace"])), _xd(c["noiseBound"])]
    else:
        out = [header("Ctxt"), b"|CX[", struct.pack("<qq", int(c["ptxtSpace"]), int(c["intFactor"])),
               _xd(c.get("ptxtMag", 0.0)), _xd(c.get("ratFactor", 1.0)), _xd(c["noiseBound"])]
    ps = sorted(int(i) for i in c["primeSet"])
    out.append(struct.pack(f"<q{len(ps)}q", len(ps), *ps))
    out.append(struct.pack("<q", len(c["parts"])))
    for idx, rows, hnd in c["parts"]:
        out.append(write_rows(idx, rows))
        out.append(struct.pack("<qqq", *hnd))
    out.append(b"]CX|")
    return b"".join(out)


def read_ctxt(buf, off=0, legacy=False):
    if not legacy:
        off = _read_header(buf, off, "Ctxt")
    if buf[off:off + 4] != b"|CX[":
        raise ValueError("Could not find pre-ciphertext eye catcher")
    off += 4
    if legacy:
        (ptxtSpace,) = struct.unpack_from("<q", buf, off)
        intFactor = 1
        off += 8
    else:
        ptxtSpace, intFactor = struct.unpack_from("<qq", buf, off)
        off += 16
    xs = []
    for _ in range(1 if legacy else 3):
        m, e = struct.unpack_from("<dq", buf, off)
        xs.append(xdouble_value(m, e))
        off += 16
    if legacy:
        xs = [0.0, 1.0, xs[0]]
    (card,) = struct.unpack_from("<q", buf, off)
    ps = list(struct.unpack_from(f"<{card}q", buf, off + 8))
    off += 8 + 8 * card
    (nparts,) = struct.unpack_from("<q", buf, off)
    off += 8
    parts = []
    for _ in range(nparts):
        idx, rows, off = read_rows(buf, off)
        hnd = struct.unpack_from("<qqq", buf, off)
        off += 24
        parts.append((idx, rows, tuple(int(v) for v in hnd)))
    if buf[off:off + 4] != b"]CX|":
        raise ValueError("Could not find post-ciphertext eye catcher")
    return {"ptxtSpace": ptxtSpace, "intFactor": intFactor, "ptxtMag": xs[0], "ratFactor": xs[1],
            "noiseBound": xs[2], "primeSet": sorted(ps), "parts": parts}, off + 4


def _typed(kind, content):
    return {"type": kind, "HElibVersion": ".".join(str(v) for v in HELIB_VERSION),
            "serializationVersion": JSON_SERIALIZATION_VERSION, "content": content}


def _untyped(j, kind):
    if isinstance(j, (str, bytes)):
        j = json.loads(j)
    if j["serializationVersion"] != JSON_SERIALIZATION_VERSION:
        raise ValueError("Serialization version mismatch")
    if j["HElibVersion"] != ".".join(str(v) for v in HELIB_VERSION):
        raise ValueError("HElib version mismatch")
    if j["type"] != kind:
        raise ValueError("Type mismatch deserializing json object")
    return j["content"]


def _xdj(v):
    m, e = xdouble_parts(v)
    return {"mantissa": m, "exponent": e}


def _skh_json(h):
    return {"powerOfS": h[0], "powerOfX": h[1], "secretKeyID": h[2]}


def ctxt_to_json(c):
    return _typed("Ctxt", {
        "ptxtSpace": int(c["ptxtSpace"]), "noiseBound": _xdj(c["noiseBound"]),
        "primeSet": sorted(int(i) for i in c["primeSet"]), "intFactor": int(c["intFactor"]),
        "ptxtMag": _xdj(c.get("ptxtMag", 0.0)), "ratFactor": _xdj(c.get("ratFactor", 1.0)),
        "parts": [{"DoubleCRT": to_json(idx, rows), "skHandle": _skh_json(h)} for idx, rows, h in c["parts"]]})


def ctxt_from_json(j, primes=None, phim=None):
    c = _untyped(j, "Ctxt")
    parts = []
    for p in c["parts"]:
        idx, rows = from_json(p["DoubleCRT"], primes, phim)
        if sorted(idx) != sorted(c["primeSet"]):   # Ctxt::readJSON's sanity check (:2697-2702)
            raise ValueError("Ciphertext part's index set does not match prime set")
        h = p["skHandle"]
        parts.append((idx, rows, (h["powerOfS"], h["powerOfX"], h["secretKeyID"])))
    xd = lambda o: xdouble_value(o["mantissa"], o["exponent"])   # noqa: E731
    return {"ptxtSpace": c["ptxtSpace"], "intFactor": c["intFactor"], "ptxtMag": xd(c["ptxtMag"]),
            "ratFactor": xd(c["ratFactor"]), "noiseBound": xd(c["noiseBound"]),
            "primeSet": sorted(c["primeSet"]), "parts": parts}


def write_keyswitch(k, legacy=False):
    """k = {fromKey: (powerOfS, powerOfX, keyID), toKeyID, ptxtSpace, b: [(idx, rows)], prgSeed:
    int > 0, noiseBound}; legacy=True (the reference fixture's layout): no noiseBound."""
    seed = int(k["prgSeed"])
    if seed <= 0:
        raise ValueError("Number of bytes to write must be non-negative")   # write_raw_ZZ's assert
    nb = (seed.bit_length() + 7) // 8
    out = [b"|KM[", struct.pack("<qqq", *k["fromKey"]), struct.pack("<qq", int(k["toKeyID"]), int(k["ptxtSpace"])),
           struct.pack("<q", len(k["b"]))]
    out += [write_rows(idx, rows) for idx, rows in k["b"]]
    out += [struct.pack("<q", nb), seed.to_bytes(nb, "little")]
    if not legacy:
        out.append(_xd(k["noiseBound"]))
    out.append(b"]KM|")
    return b"".join(out)


def read_keyswitch(buf, off=0, legacy=False):
    if buf[off:off + 4] != b"|KM[":
        raise ValueError("Could not find pre-secret key eyecatcher")
    off += 4
    fk = struct.unpack_from("<qqq", buf, off)
    toKey, ptxt, nb = struct.unpack_from("<qqq", buf, off + 24)
    off += 48
    b = []
    for _ in range(nb):
        idx, rows, off = read_rows(buf, off)
        b.append((idx, rows))
    (nbytes,) = struct.unpack_from("<q", buf, off)
    if nbytes <= 0:
        raise ValueError("Number of bytes to write must be non-negative")
    seed = int.from_bytes(buf[off + 8:off + 8 + nbytes], "little")
    off += 8 + nbytes
    m, e = (0.0, 0)
    if not legacy:
        m, e = struct.unpack_from("<dq", buf, off)
        off += 16
    if buf[off:off + 4] != b"]KM|":
        raise ValueError("Could not find post-secret key eyecatcher")
    return {"fromKey": tuple(int(v) for v in fk), "toKeyID": toKey, "ptxtSpace": ptxt, "b": b,
            "prgSeed": seed, "noiseBound": xdouble_value(m, e)}, off + 4


def keyswitch_to_json(k):
    return _typed("KeySwitch", {"fromKey": _skh_json(k["fromKey"]), "toKeyID": int(k["toKeyID"]),
                                "ptxtSpace": int(k["ptxtSpace"]),
                                "b": [to_json(idx, rows) for idx, rows in k["b"]],
                                "prgSeed": {"number": str(int(k["prgSeed"]))},
                                "noiseBound": _xdj(k["noiseBound"])})


def keyswitch_from_json(j, primes=None, phim=None):
    c = _untyped(j, "KeySwitch")
    h = c["fromKey"]
    return {"fromKey": (h["powerOfS"], h["powerOfX"], h["secretKeyID"]), "toKeyID": c["toKeyID"],
            "ptxtSpace": c["ptxtSpace"], "b": [from_json(x, primes, phim) for x in c["b"]],
            "prgSeed": int(c["prgSeed"]["number"]),
            "noiseBound": xdouble_value(c["noiseBound"]["mantissa"], c["noiseBound"]["exponent"])}


# ---- helib_amd.ctxt.Ctxt objects (any backend whose parts have getIndexSet()/download()) -----
def from_ctxt(ct, b=0):
    """helib_amd.ctxt.Ctxt -> the plain description (batch element b); parts in the reference's
    order: the part of "1" first, then "s", then the rest (Ctxt::addPart appends, the constant
    part is created first)."""
    import math
    order = sorted(ct.parts, key=lambda h: _handle_of(h)[:2])
    parts = [(ct.parts[h].getIndexSet(), ct.parts[h].download()[:, b], _handle_of(h)) for h in order]
    return {"ptxtSpace": ct.ptxtSpace, "intFactor": ct.intFactor, "ptxtMag": 0.0, "ratFactor": 1.0,
            "noiseBound": _xd_from_ln(ct.lnNoise),
            "primeSet": sorted(ct.primeSet), "parts": parts}


def to_ctxt(desc, ctxt_cls, context, ops, make_poly, **kw):
    """the plain description -> ctxt_cls (helib_amd.ctxt.Ctxt); make_poly(idx, rows) builds a
    backend DoubleCRT."""
    import math
    ct = ctxt_cls(context, ops, **kw)
    ct.parts = {_handle_name(h[0], h[1]): make_poly(idx, rows) for idx, rows, h in desc["parts"]}
    ct.primeSet = frozenset(desc["primeSet"])
    ct.ptxtSpace, ct.intFactor = desc["ptxtSpace"], desc["intFactor"]
    nb = desc["noiseBound"]
    ct.lnNoise = (math.log(nb[0]) + 114 * math.log(2.0) * nb[1]) if isinstance(nb, tuple) else \
        (math.log(nb) if nb > 0 else -math.inf)
    return ct
