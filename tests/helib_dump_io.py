"""A dump directory of tools/helib_dump/dump.cpp (genuine HElib 2.2.0): reading it, replaying it on this repository's
host logic, and -- so that the consumer is exercised without NTL -- writing a directory of the same layout from this
engine's own objects (make_synthetic_dump; keys and ciphertexts through helib_amd/wire.py's writers, which emit the
reference's binary formats byte for byte).  TEST INFRASTRUCTURE."""
import json
import os
import struct

import numpy as np

from helib_amd import ctxt as hc, keys as hk, wire


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def read_modsizes(buf):
    """ModuliSizes::write (src/primeChain.cpp:353-358): count, then per row write_raw_double(size) + IndexSet::writeTo."""
    (n,) = struct.unpack_from("<q", buf, 0)
    off, rows = 8, []
    for _ in range(n):
        (size,) = struct.unpack_from("<d", buf, off)
        (card,) = struct.unpack_from("<q", buf, off + 8)
        idx = list(struct.unpack_from(f"<{card}q", buf, off + 16))
        off += 16 + 8 * card
        rows.append((size, idx))
    assert off == len(buf)
    return rows


def write_modsizes(ms):
    out = struct.pack("<q", len(ms.sizes))
    for size, st in ms.sizes:
        idx = sorted(st)
        out += struct.pack(f"<dq{len(idx)}q", size, len(idx), *idx)
    return out


def load(dump_dir):
    d = {"dir": dump_dir}
    with open(os.path.join(dump_dir, "params.json")) as f:
        d["params"] = json.load(f)
    with open(os.path.join(dump_dir, "chain.json")) as f:
        d["chain"] = json.load(f)
    d["modsizes"] = read_modsizes(_read(os.path.join(dump_dir, "modsizes.bin")))
    d["context"], _ = wire.read_context(_read(os.path.join(dump_dir, "context.bin")))
    d["seckey"], _ = wire.read_seckey(_read(os.path.join(dump_dir, "seckey.bin")))
    for name in ("ct1", "ct2", "prod", "prod2", "rot"):
        d[name], _ = wire.read_ctxt(_read(os.path.join(dump_dir, name + ".bin")))
    for name in ("ptxt1", "ptxt2", "rot", "decisions"):
        with open(os.path.join(dump_dir, name + ".json")) as f:
            d[name + "_json"] = json.load(f)
    d["ksw_a"] = {}
    for w in d["seckey"]["keySwitching"]:
        sp, xp = w["fromKey"][:2]
        path = os.path.join(dump_dir, f"ksw_a_{sp}_{xp}.bin")
        if not os.path.exists(path):
            continue
        buf, off, cols = _read(path), 0, []
        while off < len(buf):
            idx, rows, off = wire.read_rows(buf, off)
            cols.append((idx, rows))
        assert len(cols) == len(w["b"]) and all(c[0] == w["b"][0][0] for c in cols)
        d["ksw_a"][(sp, xp)] = np.stack([rows for _, rows in cols])
    return d


def backends(d, kind, hx=None):
    """(chain context of this repository's host, backend, make_poly) with every prime registered under the DUMPED root"""
    P = d["params"]
    cc = hc.ChainContext(P["m"], P["p"], P["r"], bits=P["bits"], c=P["c"])
    primes, roots = d["chain"]["primes"], d["chain"]["roots"]
    if kind == "oracle":
        from oracle import oracle as O
        from oracle.backend import OPoly, OracleBackend
        octx = O.Ctx(P["m"])
        for q, r in zip(primes, roots):
            octx.add_prime(q, r)
        return cc, OracleBackend(octx, cc), (lambda idx, rows: OPoly(octx, idx, rows))
    g = hx.Context(P["m"])
    for q, r in zip(primes, roots):
        g.add_prime(q, r)
    return cc, hk.HxBackend(g, cc), (lambda idx, rows: hx.DoubleCRT(g, list(idx), 1, np.asarray(rows)[:, None, :]))


def check_chain(d, cc):
    """chain order, prime sets, digits and the ModuliSizes table: the decisions of Context::buildModChain"""
    ch = d["chain"]
    assert [int(q) for q in cc.primes] == ch["primes"], "prime chain differs from HElib's"
    assert sorted(cc.smallPrimes) == ch["smallPrimes"] and sorted(cc.ctxtPrimes) == ch["ctxtPrimes"]
    assert sorted(cc.specialPrimes) == ch["specialPrimes"]
    assert [sorted(dg) for dg in cc.digits] == ch["digits"]
    assert d["context"]["qs"] == ch["primes"] and d["context"]["m"] == cc.m
    mine = [(s, sorted(st)) for s, st in cc.modSizes.sizes]
    assert len(mine) == len(d["modsizes"])
    for (s1, i1), (s2, i2) in zip(mine, d["modsizes"]):
        assert i1 == i2 and abs(s1 - s2) <= 1e-9 * max(1.0, abs(s2)), (s1, i1, s2, i2)


def same_ctxt(got_desc, want_desc, what, noise_rtol=1e-6):
    """every word of every part, the prime set, intFactor and ptxtSpace; the noise bound to a relative tolerance
    (the estimate is floating point on both sides: HElib in xdouble, this host in log-doubles)"""
    assert got_desc["primeSet"] == want_desc["primeSet"], what
    assert (got_desc["ptxtSpace"], got_desc["intFactor"]) == (want_desc["ptxtSpace"], want_desc["intFactor"]), what
    gp = {h[:2]: (idx, rows) for idx, rows, h in got_desc["parts"]}
    wp = {h[:2]: (idx, rows) for idx, rows, h in want_desc["parts"]}
    assert set(gp) == set(wp), what
    for h in wp:
        assert gp[h][0] == wp[h][0], (what, h)
        assert np.array_equal(gp[h][1], wp[h][1]), (what, h)
    g, w = got_desc["noiseBound"], want_desc["noiseBound"]
    g = wire.xdouble_value(*g) if isinstance(g, tuple) else float(g)
    w = wire.xdouble_value(*w) if isinstance(w, tuple) else float(w)
    assert abs(g - w) <= noise_rtol * abs(w), (what, g, w)


def attach(desc, sk, cc, be, make_poly):
    """a loaded ciphertext wired to the key's matrices the way PubKey::Encrypt's result is (helib_amd.keys._newCtxt)"""
    import math
    relin = sk.keySwitching.get((2, 1))
    ct = wire.to_ctxt(desc, hc.Ctxt, cc, be.ops, make_poly, ksw=relin.W if relin else None,
                      ksw_ptxtSpace=relin.ptxtSpace if relin else None,
                      ksw_noise=math.log(relin.noiseBound) if relin else None)
    for (sp, xp), ks in sk.keySwitching.items():
        if sp == 1 and xp > 1:
            ct.ksw_auto[xp] = ks.W
        elif sp > 2 and xp == 1:
            ct.ksw_pow[sp] = ks.W
    ct.ksw_map = getattr(sk, "keySwitchMap", None)
    return ct


def replay(d, kind, hx=None, measured_noise=None):
    """ct1.multiplyBy(ct2), the level-2 product and one smartAutomorph with this repository's host logic on the dumped
    keys; every result against the dumped one.  Returns the secret key object (for decryption checks)."""
    cc, be, make_poly = backends(d, kind, hx)
    check_chain(d, cc)
    sk = wire.to_seckey(d["seckey"], hk.SecKey, cc, be, make_poly, ksw_a=d["ksw_a"], seed=1)
    assert (2, 1) in sk.keySwitching, "the dump has no relinearisation matrix with its a-columns"
    ct1, ct2 = (attach(d[n], sk, cc, be, make_poly) for n in ("ct1", "ct2"))
    p2r = cc.ptxtSpace
    assert sk.Decrypt(ct1) == [int(v) % p2r for v in d["ptxt1_json"]]      # the dumped key decrypts the dumped ciphertext
    assert sk.Decrypt(ct2) == [int(v) % p2r for v in d["ptxt2_json"]]
    ct1.multiplyBy(ct2)
    same_ctxt(wire.from_ctxt(ct1), d["prod"], "ct1.multiplyBy(ct2)")
    p2, other, rot = (attach(d["prod"], sk, cc, be, make_poly) for _ in range(3))
    p2.multiplyBy(other)
    same_ctxt(wire.from_ctxt(p2), d["prod2"], "prod.multiplyBy(prod)")
    rot.smartAutomorph(d["rot_json"]["k"])
    same_ctxt(wire.from_ctxt(rot), d["rot"], "prod.smartAutomorph(k)")
    return sk


def make_synthetic_dump(dump_dir, m, p, bits, c=3, seed=5):
    """The same files from THIS engine's objects (oracle backend): exercises load() / replay() where there is no HElib.
    Roots are the oracle's (FindPrimRootT); a-columns are the engine's explicit ones."""
    from oracle import oracle as O
    from oracle.backend import OPoly, OracleBackend
    os.makedirs(dump_dir, exist_ok=True)
    cc = hc.ChainContext(m, p, 1, bits=bits, c=c)
    octx = O.Ctx(m)
    for q in cc.primes:
        octx.add_prime(q)
    be = OracleBackend(octx, cc)
    sk = hk.SecKey(cc, be, seed=seed)
    sk.GenSecKey(maxDegKswitch=2)
    import math
    k = next(x for x in (3, 2, 5, 7) if math.gcd(x, m) == 1)
    sk.GenKeySWmatrix(1, k)
    sk.setKeySwitchMap()
    with open(os.path.join(dump_dir, "params.json"), "w") as f:
        json.dump({"m": m, "p": p, "r": 1, "bits": bits, "c": c, "helib_version": "synthetic (this engine)", "phim": cc.phim}, f)
    with open(os.path.join(dump_dir, "chain.json"), "w") as f:
        json.dump({"primes": [int(q) for q in cc.primes], "roots": [int(r) for r in octx.roots],
                   "smallPrimes": sorted(cc.smallPrimes), "ctxtPrimes": sorted(cc.ctxtPrimes),
                   "specialPrimes": sorted(cc.specialPrimes), "digits": [sorted(dg) for dg in cc.digits]}, f)
    with open(os.path.join(dump_dir, "modsizes.bin"), "wb") as f:
        f.write(write_modsizes(cc.modSizes))
    desc = wire.from_seckey(sk)
    with open(os.path.join(dump_dir, "context.bin"), "wb") as f:
        f.write(wire.write_context(desc["context"]))
    with open(os.path.join(dump_dir, "seckey.bin"), "wb") as f:
        f.write(wire.write_seckey(desc, engine_only=True))
    with open(os.path.join(dump_dir, "pubkey.bin"), "wb") as f:
        f.write(wire.write_pubkey(wire.from_pubkey(sk), engine_only=True))
    allp = list(cc.ctxtPrimes) + list(cc.specialPrimes)
    for (sp, xp), w in sk.keySwitching.items():
        with open(os.path.join(dump_dir, f"ksw_a_{sp}_{xp}.bin"), "wb") as f:
            for i in range(len(w.a)):
                f.write(wire.write_rows(allp, np.asarray(w.a[i])))
    rng = np.random.default_rng(seed)
    msgs = [rng.integers(0, p, size=cc.phim) for _ in range(2)]
    for j, mm in enumerate(msgs):
        with open(os.path.join(dump_dir, f"ptxt{j + 1}.json"), "w") as f:
            json.dump([int(v) for v in mm], f)
    ct1, ct2 = sk.Encrypt(msgs[0]), sk.Encrypt(msgs[1])

    def put(name, ct):
        with open(os.path.join(dump_dir, name + ".bin"), "wb") as f:
            f.write(wire.write_ctxt(wire.from_ctxt(ct)))
    put("ct1", ct1)
    put("ct2", ct2)
    ct1.multiplyBy(ct2)
    put("prod", ct1)
    d_prod = wire.from_ctxt(ct1)
    mk = lambda idx, rows: OPoly(octx, idx, rows)   # noqa: E731
    a, b, r = (attach(d_prod, sk, cc, be, mk) for _ in range(3))
    a.multiplyBy(b)
    put("prod2", a)
    r.smartAutomorph(k)
    put("rot", r)
    with open(os.path.join(dump_dir, "rot.json"), "w") as f:
        json.dump({"k": int(k)}, f)
    with open(os.path.join(dump_dir, "decisions.json"), "w") as f:
        json.dump({n: {"primeSet": sorted(x.primeSet), "ptxtSpace": x.ptxtSpace, "lnNoiseBound": x.lnNoise}
                   for n, x in (("prod", ct1), ("prod2", a), ("rot", r))}, f)
