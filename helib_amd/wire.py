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
    from .ctxt import handle_powers
    sp, xp = handle_powers(h)
    return (sp, xp, 0)


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
               _xd(c.get("ptxtMag", 1.0)), _xd(c.get("ratFactor", 1.0)), _xd(c["noiseBound"])]
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
        xs = [1.0, 1.0, xs[0]]
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
        "ptxtMag": _xdj(c.get("ptxtMag", 1.0)), "ratFactor": _xdj(c.get("ratFactor", 1.0)),
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


ENGINE_ONLY_MSG = ("this key-switching matrix keeps its a-columns explicitly (helib_amd.keys): they are not on the "
                   "reference's wire, which carries only prgSeed -- a reference HElib would load the blob, "
                   "regenerate DIFFERENT a-columns from the seed and every reLinearize / rotation would silently "
                   "decrypt to garbage.  Pass engine_only=True to write it anyway (loadable by this engine with "
                   "to_seckey / to_pubkey(..., ksw_a=...) only)")


def _check_engine_only(k, engine_only):
    if k.get("explicit_a") and not engine_only:
        raise ValueError(ENGINE_ONLY_MSG)


def write_keyswitch(k, legacy=False, engine_only=False):
    """k = {fromKey: (powerOfS, powerOfX, keyID), toKeyID, ptxtSpace, b: [(idx, rows)], prgSeed:
    int > 0, noiseBound}; legacy=True (the reference fixture's layout): no noiseBound.
    A description marked explicit_a (made by from_pubkey / from_seckey from this engine's keys) is
    refused unless engine_only=True: see ENGINE_ONLY_MSG."""
    _check_engine_only(k, engine_only)
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


def keyswitch_to_json(k, engine_only=False):
    _check_engine_only(k, engine_only)
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
    ckks = getattr(ct.context, "ckks", False)
    return {"ptxtSpace": ct.ptxtSpace, "intFactor": ct.intFactor,
            "ptxtMag": float(ct.ptxtMag),
            "ratFactor": _xd_from_ln(ct.lnRatFactor) if ckks else 1.0,
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
    ct.ptxtMag = desc.get("ptxtMag", 1.0) if not isinstance(desc.get("ptxtMag", 1.0), tuple) else \
        xdouble_value(*desc["ptxtMag"])
    rf = desc.get("ratFactor", 1.0)
    ct.lnRatFactor = (math.log(rf[0]) + 114 * math.log(2.0) * rf[1]) if isinstance(rf, tuple) else \
        (math.log(rf) if rf > 0 else 0.0)
    return ct


# ---- containers: Context, PubKey, SecKey ---------------------------------------------------------
def _vec_long(v, int_size=BIT64):
    """write_ntl_vec_long (src/binio.cpp:103-125)"""
    fmt = "<u8" if int_size == BIT64 else "<u4"
    return struct.pack("<ii", len(v), int_size) + np.asarray(v, dtype=np.int64).astype(fmt).tobytes()


def _read_vec_long(buf, off):
    n, int_size = struct.unpack_from("<ii", buf, off)
    if int_size not in (BIT32, BIT64) or n < 0:
        raise ValueError("intSize must be 32 or 64 bit for binary IO")
    dt = "<i8" if int_size == BIT64 else "<i4"
    v = [int(x) for x in np.frombuffer(buf, dtype=dt, count=n, offset=off + 8)]
    return v, off + 8 + n * int_size


def _longs(v):
    return struct.pack(f"<q{len(v)}q", len(v), *[int(x) for x in v])     # write_raw_vector<long>


def _read_longs(buf, off):
    (n,) = struct.unpack_from("<q", buf, off)
    if n < 0 or n > (len(buf) - off) // 8:
        raise ValueError("implausible vector length")
    return list(struct.unpack_from(f"<{n}q", buf, off + 8)), off + 8 + 8 * n


def _eye(buf, off, tag, what):
    if buf[off:off + 4] != tag:
        raise ValueError(f"Could not find {what} eye catcher")
    return off + 4


def _base(c):
    return [struct.pack("<qqq", int(c["p"]), int(c["r"]), int(c["m"])), _longs(c["gens"]), _longs(c["ords"])]


def write_context_base(c):
    return b"".join([b"|BS["] + _base(c) + [b"]BS|"])


def write_context(c, legacy=False):
    """Context::writeTo (src/Context.cpp:324-387).  c = {p, r, m, gens, ords (negated = bad
    dimension), stdev, scale, smallPrimes, specialPrimes, qs, digits, hwt_param, e_param,
    ePrime_param, mvec, build_cache, alsoThick}.  gens/ords describe Zm*/<p> (PAlgebra), which
    this engine never computes: they are carried through as data.
    legacy=True: the two blocks of the reference's fixture (context base |BS[..]BS| = p r m gens
    ords, then |CN[..]CN| = stdev as a double, the index sets, primes, digits, mvec and two flags)."""
    base = _base(c)
    sets = [_longs(sorted(c["smallPrimes"])), _longs(sorted(c["specialPrimes"])), _longs(c["qs"]),
            struct.pack("<q", len(c["digits"]))] + [_longs(sorted(d)) for d in c["digits"]]
    tail = [_vec_long(c.get("mvec", [])), struct.pack("<qq", int(c.get("build_cache", 0)), int(c.get("alsoThick", 0)))]
    if legacy:
        return b"".join([b"|BS["] + base + [b"]BS|", b"|CN[", struct.pack("<d", float(c["stdev"]))] + sets
                        + tail + [b"]CN|"])
    return b"".join([header("Context"), b"|CN["] + base + [_xd(c["stdev"]), struct.pack("<d", float(c["scale"]))]
                    + sets + [struct.pack("<qqq", int(c.get("hwt_param", 0)), int(c.get("e_param", 0)),
                                          int(c.get("ePrime_param", 0)))] + tail + [b"]CN|"])


def _read_base(buf, off):
    p, r, m = struct.unpack_from("<qqq", buf, off)
    gens, off = _read_longs(buf, off + 24)
    ords, off = _read_longs(buf, off)
    return {"p": p, "r": r, "m": m, "gens": gens, "ords": ords}, off


def read_context_base(buf, off=0):
    """the legacy |BS[ p r m gens ords ]BS| block"""
    off = _eye(buf, off, b"|BS[", "pre-context-base")
    c, off = _read_base(buf, off)
    return c, _eye(buf, off, b"]BS|", "post-context-base")


def read_context(buf, off=0, legacy=False):
    """Context::readParamsFrom (src/Context.cpp:389-442) -> (dict, offset)"""
    if legacy:
        c, off = read_context_base(buf, off)
        off = _eye(buf, off, b"|CN[", "pre-context")
        (c["stdev"],) = struct.unpack_from("<d", buf, off)
        off += 8
        c["scale"] = 10.0
    else:
        off = _read_header(buf, off, "Context")
        off = _eye(buf, off, b"|CN[", "pre-context")
        c, off = _read_base(buf, off)
        mnt, e, c["scale"] = struct.unpack_from("<dqd", buf, off)
        c["stdev"] = xdouble_value(mnt, e)
        off += 24
    c["smallPrimes"], off = _read_longs(buf, off)
    c["specialPrimes"], off = _read_longs(buf, off)
    c["qs"], off = _read_longs(buf, off)
    (nd,) = struct.unpack_from("<q", buf, off)
    off += 8
    c["digits"] = []
    for _ in range(nd):
        d, off = _read_longs(buf, off)
        c["digits"].append(d)
    if not legacy:
        c["hwt_param"], c["e_param"], c["ePrime_param"] = struct.unpack_from("<qqq", buf, off)
        off += 24
    c["mvec"], off = _read_vec_long(buf, off)
    c["build_cache"], c["alsoThick"] = struct.unpack_from("<qq", buf, off)
    return c, _eye(buf, off + 16, b"]CN|", "post-context")


def context_to_json(c):
    idx = lambda s: sorted(int(i) for i in s)   # IndexSet::writeToJSON: a plain array (src/IndexSet.cpp:319-330)  # noqa: E731
    return _typed("Context", {
        "m": int(c["m"]), "p": int(c["p"]), "r": int(c["r"]), "gens": list(c["gens"]), "ords": list(c["ords"]),
        "stdev": _xdj(c["stdev"]), "scale": float(c["scale"]),
        "smallPrimes": idx(c["smallPrimes"]), "specialPrimes": idx(c["specialPrimes"]),
        "qs": [int(q) for q in c["qs"]], "digits": [idx(d) for d in c["digits"]],
        "hwt_param": int(c.get("hwt_param", 0)), "e_param": int(c.get("e_param", 0)),
        "ePrime_param": int(c.get("ePrime_param", 0)), "mvec": list(c.get("mvec", [])),
        "build_cache": bool(c.get("build_cache", 0)), "alsoThick": bool(c.get("alsoThick", 0))})


def context_from_json(j):
    c = dict(_untyped(j, "Context"))
    for k in ("smallPrimes", "specialPrimes"):
        c[k] = sorted(c[k])
    c["digits"] = [sorted(d) for d in c["digits"]]
    c["stdev"] = xdouble_value(c["stdev"]["mantissa"], c["stdev"]["exponent"])
    c["build_cache"], c["alsoThick"] = int(c["build_cache"]), int(c["alsoThick"])
    return c


def context_of(cc, gens=(), ords=()):
    """helib_amd.ctxt.ChainContext -> the Context description"""
    return {"p": cc.p, "r": cc.r, "m": cc.m, "gens": list(gens), "ords": list(ords), "stdev": cc.stdev,
            "scale": float(cc.scale), "smallPrimes": list(cc.smallPrimes), "specialPrimes": list(cc.specialPrimes),
            "qs": [int(q) for q in cc.primes], "digits": [list(d) for d in cc.digits], "hwt_param": int(cc.hwt),
            "e_param": 0, "ePrime_param": 0, "mvec": [], "build_cache": 0, "alsoThick": 0}


def write_pubkey(k, legacy=False, engine_only=False):
    """PubKey::writeTo (src/keys.cpp:888-921).  k = {context, pubEncrKey (ctxt description),
    skBounds [double], keySwitching [key-switch descriptions], keySwitchMap [[long]], KS_strategy
    [long], recryptKeyID, recryptEkey (ctxt description)}.
    legacy=True (the fixture): context base only, skBounds still the integer Hamming weights."""
    out = [b"|PK["] if legacy else [header("PubKey"), b"|PK["]
    if legacy:
        out.append(write_context_base(k["context"]))
    else:
        out.append(write_context(k["context"]))
    out.append(write_ctxt(k["pubEncrKey"], legacy))
    if legacy:
        out.append(_longs(k["skBounds"]))
    else:
        out.append(struct.pack(f"<q{len(k['skBounds'])}d", len(k["skBounds"]), *[float(x) for x in k["skBounds"]]))
    out.append(struct.pack("<q", len(k["keySwitching"])))
    out += [write_keyswitch(w, legacy, engine_only) for w in k["keySwitching"]]
    out.append(struct.pack("<q", len(k["keySwitchMap"])))
    out += [_longs(v) for v in k["keySwitchMap"]]
    out += [_vec_long(k["KS_strategy"]), struct.pack("<q", int(k["recryptKeyID"])),
            write_ctxt(k["recryptEkey"], legacy), b"]PK|"]
    return b"".join(out)


def read_pubkey(buf, off=0, legacy=False, context=None):
    """PubKey::readFrom (src/keys.cpp:923-974); with `context` (a description) the reference's
    "Context mismatch" check is applied."""
    if not legacy:
        off = _read_header(buf, off, "PubKey")
    off = _eye(buf, off, b"|PK[", "pre-public key")
    k = {}
    if legacy:
        k["context"], off = read_context_base(buf, off)
    else:
        k["context"], off = read_context(buf, off)
    if context is not None:
        for f in k["context"]:
            if f in context and k["context"][f] != context[f]:
                raise ValueError("Context mismatch")
    k["pubEncrKey"], off = read_ctxt(buf, off, legacy)
    if legacy:
        k["skBounds"], off = _read_longs(buf, off)
    else:
        (n,) = struct.unpack_from("<q", buf, off)
        k["skBounds"] = list(struct.unpack_from(f"<{n}d", buf, off + 8))
        off += 8 + 8 * n
    (n,) = struct.unpack_from("<q", buf, off)
    off += 8
    k["keySwitching"] = []
    for _ in range(n):
        w, off = read_keyswitch(buf, off, legacy)
        k["keySwitching"].append(w)
    (n,) = struct.unpack_from("<q", buf, off)
    off += 8
    k["keySwitchMap"] = []
    for _ in range(n):
        v, off = _read_longs(buf, off)
        k["keySwitchMap"].append(v)
    k["KS_strategy"], off = _read_vec_long(buf, off)
    (k["recryptKeyID"],) = struct.unpack_from("<q", buf, off)
    k["recryptEkey"], off = read_ctxt(buf, off + 8, legacy)
    return k, _eye(buf, off, b"]PK|", "post-public key")


def write_seckey(k, legacy=False, sk_only=False, engine_only=False):
    """SecKey::writeTo (src/keys.cpp:1736-1753): the public key (or, sk_only, just the context),
    then vector<DoubleCRT> sKeys.  k = the PubKey description + {"sKeys": [(idx, rows)]}."""
    out = [b"|SK["] if legacy else [header("SecKey"), b"|SK["]
    out.append(write_context(k["context"], legacy) if sk_only else write_pubkey(k, legacy, engine_only))
    out.append(struct.pack("<q", len(k["sKeys"])))
    out += [write_rows(idx, rows) for idx, rows in k["sKeys"]]
    out.append(b"]SK|")
    return b"".join(out)


def read_seckey(buf, off=0, legacy=False, sk_only=False, context=None):
    if not legacy:
        off = _read_header(buf, off, "SecKey")
    off = _eye(buf, off, b"|SK[", "pre-secret key")
    if sk_only:
        c, off = read_context(buf, off, legacy)
        if context is not None and any(c[f] != context[f] for f in c if f in context):
            raise ValueError("Context mismatch")
        k = {"context": c}
    else:
        k, off = read_pubkey(buf, off, legacy, context)
    (n,) = struct.unpack_from("<q", buf, off)
    off += 8
    k["sKeys"] = []
    for _ in range(n):
        idx, rows, off = read_rows(buf, off)
        k["sKeys"].append((idx, rows))
    return k, _eye(buf, off, b"]SK|", "post-secret key")


def pubkey_to_json(k, engine_only=False):
    return _typed("PubKey", {
        "context": context_to_json(k["context"]), "pubEncrKey": ctxt_to_json(k["pubEncrKey"]),
        "skBounds": [float(x) for x in k["skBounds"]],
        "keySwitching": [keyswitch_to_json(w, engine_only) for w in k["keySwitching"]],
        "keySwitchMap": [list(v) for v in k["keySwitchMap"]], "KS_strategy": list(k["KS_strategy"]),
        "recryptKeyID": int(k["recryptKeyID"]),
        "recryptEkey": ctxt_to_json(k["recryptEkey"]) if k["recryptKeyID"] >= 0 else "nullptr"})


def pubkey_from_json(j, primes=None, phim=None):
    c = _untyped(j, "PubKey")
    k = {"context": context_from_json(c["context"]), "pubEncrKey": ctxt_from_json(c["pubEncrKey"], primes, phim),
         "skBounds": list(c["skBounds"]),
         "keySwitching": [keyswitch_from_json(w, primes, phim) for w in c["keySwitching"]],
         "keySwitchMap": [list(v) for v in c["keySwitchMap"]], "KS_strategy": list(c["KS_strategy"]),
         "recryptKeyID": c["recryptKeyID"]}
    if c["recryptKeyID"] >= 0:
        k["recryptEkey"] = ctxt_from_json(c["recryptEkey"], primes, phim)
    else:   # left as constructed by Ctxt(pubKey): empty, over the ctxt primes
        k["recryptEkey"] = {"ptxtSpace": k["pubEncrKey"]["ptxtSpace"], "intFactor": 1, "ptxtMag": 1.0,
                            "ratFactor": 1.0, "noiseBound": 0.0, "primeSet": list(k["pubEncrKey"]["primeSet"]),
                            "parts": []}
    return k


def seckey_to_json(k, sk_only=False, engine_only=False):
    body = {"context": context_to_json(k["context"])} if sk_only else {"PubKey": pubkey_to_json(k, engine_only)}
    body["sKeys"] = [to_json(idx, rows) for idx, rows in k["sKeys"]]
    return _typed("SecKey", body)


def seckey_from_json(j, sk_only=False, primes=None, phim=None):
    c = _untyped(j, "SecKey")
    k = {"context": context_from_json(c["context"])} if sk_only else pubkey_from_json(c["PubKey"], primes, phim)
    k["sKeys"] = [from_json(x, primes, phim) for x in c["sKeys"]]
    return k


def key_switch_map(m, keySwitching, keyId=0):
    """PubKey::setKeySwitchMap in the stored form (src/keys.cpp:122-172): entry k = the INDEX in
    keySwitching of the matrix for the first step of X -> X^k, -1 if unreachable."""
    edges = [(w["fromKey"][1], i) for i, w in enumerate(keySwitching)
             if w["toKeyID"] == keyId and w["fromKey"][0] == 1 and w["fromKey"][2] == keyId]
    kmap = [-1] * m
    queue, head = [1], 0
    while head < len(queue):
        cur = queue[head]
        head += 1
        for n, idx in edges:
            nxt = cur * n % m
            if kmap[nxt] == -1:
                kmap[nxt] = idx
                queue.append(nxt)
    return kmap


# ---- helib_amd.keys objects ----------------------------------------------------------------------
def from_pubkey(pk, gens=(), ords=()):
    """helib_amd.keys.PubKey/SecKey -> the PubKey description.  The a-columns of a key-switching
    matrix are regenerated by the reference from prgSeed with NTL's PRG; this engine keeps them
    explicitly (helib_amd.keys.KeySwitchInfo.a), so the matrix descriptions made here carry prgSeed = 1
    and the mark explicit_a: the writers (write_pubkey / write_seckey / *_to_json) refuse them unless
    engine_only=True, because a reference HElib would load such a blob without complaint and
    regenerate different a-columns (ENGINE_ONLY_MSG)."""
    import math
    cc = pk.cc
    enc = {"ptxtSpace": pk.ptxtSpace, "intFactor": 1, "ptxtMag": 1.0, "ratFactor": 1.0,
           "noiseBound": float(pk.pubEncrKeyNoise), "primeSet": sorted(cc.ctxtPrimes),
           "parts": [(pk.pubEncrKey[0].getIndexSet(), pk.pubEncrKey[0].download()[:, 0], (0, 1, 0)),
                     (pk.pubEncrKey[1].getIndexSet(), pk.pubEncrKey[1].download()[:, 0], (1, 1, 0))]}
    allp = list(cc.ctxtPrimes) + list(cc.specialPrimes)
    ksw = [{"fromKey": (sp, xp, 0), "toKeyID": 0, "ptxtSpace": w.ptxtSpace,
            "b": [(allp, np.asarray(w.b[i])) for i in range(len(w.b))], "prgSeed": 1, "explicit_a": True,
            "noiseBound": float(w.noiseBound)} for (sp, xp), w in pk.keySwitching.items()]
    kmap = [key_switch_map(cc.m, ksw)] if getattr(pk, "keySwitchMap", None) else []
    empty = {"ptxtSpace": pk.ptxtSpace, "intFactor": 1, "ptxtMag": 1.0, "ratFactor": 1.0, "noiseBound": 0.0,
             "primeSet": sorted(cc.ctxtPrimes), "parts": []}
    z = getattr(pk, "zMStar", None)
    if z is not None and not gens:
        gens, ords = z.gens, z.signedOrds()
    return {"context": context_of(cc, gens, ords), "pubEncrKey": enc, "skBounds": [float(b) for b in pk.skBounds],
            "keySwitching": ksw, "keySwitchMap": kmap, "KS_strategy": list(getattr(pk, "KS_strategy", [])),
            "recryptKeyID": -1, "recryptEkey": empty}


def from_seckey(sk, gens=(), ords=()):
    k = from_pubkey(sk, gens, ords)
    allp = list(sk.cc.ctxtPrimes) + list(sk.cc.specialPrimes)
    k["sKeys"] = []
    for s in sk.sKeys:
        d = sk.be.fromCoeffs(allp, s)
        k["sKeys"].append((d.getIndexSet(), d.download()[:, 0]))
    return k


def to_seckey(desc, key_cls, cc, be, make_poly, ksw_a=None, seed=None):
    """the SecKey (or PubKey) description -> key_cls (helib_amd.keys.SecKey / PubKey) over backend
    `be`; make_poly(idx, rows) builds a backend DoubleCRT.  Key-switching matrices are installed
    only when their a-columns are supplied (ksw_a[(powerOfS, powerOfX)] = [D][rows][N]) -- see
    from_pubkey."""
    from . import keys as hk
    ctx = desc["context"]
    if (ctx["m"], ctx["p"], ctx["r"]) != (cc.m, cc.p, cc.r) or \
            ("qs" in ctx and [int(q) for q in ctx["qs"]] != [int(q) for q in cc.primes]):
        raise ValueError("Context mismatch")
    key = key_cls(cc, be, seed)
    enc = desc["pubEncrKey"]
    byh = {h[:2]: make_poly(idx, rows) for idx, rows, h in enc["parts"]}
    key.pubEncrKey = (byh[(0, 1)], byh[(1, 1)])
    nb = enc["noiseBound"]
    key.pubEncrKeyNoise = xdouble_value(*nb) if isinstance(nb, tuple) else float(nb)
    key.ptxtSpace = enc["ptxtSpace"]
    key.skBounds = [float(b) for b in desc["skBounds"]]
    for w in desc["keySwitching"]:
        k2 = tuple(w["fromKey"][:2])
        if ksw_a is not None and k2 in ksw_a:
            hb = np.stack([rows for _, rows in w["b"]])
            ha = np.asarray(ksw_a[k2])
            idx = w["b"][0][0]
            nbw = w["noiseBound"]
            key.keySwitching[k2] = hk.KeySwitchInfo(k2[0], k2[1], be.keySwitch(idx, hb, ha), w["ptxtSpace"],
                                                    xdouble_value(*nbw) if isinstance(nbw, tuple) else float(nbw),
                                                    hb, ha)
    if desc.get("keySwitchMap") and any(k[0] == 1 and k[1] > 1 for k in key.keySwitching):
        key.setKeySwitchMap()
    if "sKeys" in desc and hasattr(key, "sKeys"):
        for idx, rows in desc["sKeys"]:
            poly = be.toPoly(make_poly(idx, rows))
            key.sKeys.append(np.array([int(v) for v in poly], dtype=np.int64))
    return key
