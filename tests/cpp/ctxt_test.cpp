// GPU check of the C++ host side (include/helib_amd_ctxt.hpp): the reference's multiplyBy
// sequence (bringToSet x2 -> tensorProduct -> dropSmallAndSpecialPrimes -> reLinearize), addCtxt
// and smartAutomorph driven from C++ over the C ABI.  Inputs (keys, ciphertext parts, roots) come
// from a file written by tests/test_gpu_parity.py, which runs the python mirror on the same data
// and compares every output word, the prime sets, intFactor and the noise estimate.
//   ctxt_test <in.bin> <out.bin>
// in : int64 m, p, bits, k, measure, nprimes, D, nall, L, N ; uint64 roots[nprimes] ;
//      uint64 kb[D][nall][N], ka[D][nall][N], kbk[..], kak[..] (matrix for s(X^k)) ;
//      uint64 a0[L][N], a1[L][N], b0[L][N], b1[L][N]
// out: per result (product, product+product, rotated product, rotated twice more through the map):
//      int64 nprimes_in_set, intFactor, nparts ; double lnNoise ; int64 primeSet[] ;
//      per part: int64 powerOfS, powerOfX, nrows, idx[nrows] ; uint64 rows[nrows][N]
#include <cstdio>
#include <cstdlib>

#include "helib_amd_ctxt.hpp"

using namespace helib_amd;

static std::vector<uint64_t> rd(FILE* f, size_t n)
{
  std::vector<uint64_t> v(n);
  if (fread(v.data(), 8, n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
  return v;
}
static void wr(FILE* f, const void* p, size_t bytes) { fwrite(p, 1, bytes, f); }
static void dump(FILE* f, const Ctxt& c)
{
  int64_t hdr[3] = {(int64_t)c.primeSet.size(), c.intFactor, (int64_t)c.parts.size()};
  wr(f, hdr, sizeof hdr);
  double ln = c.lnNoise;
  wr(f, &ln, 8);
  for (int i : c.primeSet) {
    int64_t v = i;
    wr(f, &v, 8);
  }
  for (auto& kv : c.parts) {
    IndexSet idx = kv.second.getIndexSet();
    int64_t h[3] = {kv.first.powerOfS, kv.first.powerOfX, (int64_t)idx.size()};
    wr(f, h, sizeof h);
    for (int i : idx) {
      int64_t v = i;
      wr(f, &v, 8);
    }
    std::vector<uint64_t> rows = kv.second.getRows();
    wr(f, rows.data(), rows.size() * 8);
  }
}

int main(int argc, char** argv)
{
  if (argc < 3)
    return 2;
  Ctxt::deferNorms() = getenv("HX_TEST_DEFER_NORMS") != nullptr;   // measured norms read back lazily (LazyLn)
  FILE* f = fopen(argv[1], "rb");
  if (!f)
    return 2;
  auto H = rd(f, 10);
  long m = (long)H[0], p = (long)H[1], bits = (long)H[2], k = (long)H[3];
  bool measure = H[4] != 0;
  size_t nprimes = H[5], D = H[6], nall = H[7], L = H[8], N = H[9];
  try {
    ChainContext cc(m, p, 1, bits, 3);
    if (cc.primes.size() != nprimes || cc.digits.size() != D || cc.ctxtPrimes.size() != L ||
        cc.ctxtPrimes.size() + cc.specialPrimes.size() != nall || (size_t)cc.phim != N) {
      fprintf(stderr, "chain mismatch\n");
      return 3;
    }
    auto roots = rd(f, nprimes);
    auto dev = cc.makeDeviceContext(0, &roots);
    IndexSet allp = cc.ctxtPrimes;
    allp.insert(allp.end(), cc.specialPrimes.begin(), cc.specialPrimes.end());
    auto kb = rd(f, D * nall * N), ka = rd(f, D * nall * N), kbk = rd(f, D * nall * N), kak = rd(f, D * nall * N);
    KeySwitch W(*dev, (int)D, allp, kb, ka), Wk(*dev, (int)D, allp, kbk, kak);
    KeySet keys;
    keys.relin = &W;
    keys.automorph[k] = &Wk;
    keys.ptxtSpace = cc.ptxtSpace;
    keys.lnNoise = std::log(cc.gaussBound() * cc.ptxtSpace);
    auto mk = [&](FILE* fp) {
      DoubleCRT d(*dev, cc.ctxtPrimes, 1);
      d.setRows(rd(fp, L * N));
      return d;
    };
    DoubleCRT a0 = mk(f), a1 = mk(f), b0 = mk(f), b1 = mk(f);
    fclose(f);
    Ctxt ca = Ctxt::fresh(cc, *dev, keys, std::move(a0), std::move(a1));
    Ctxt cb = Ctxt::fresh(cc, *dev, keys, std::move(b0), std::move(b1));
    ca.measure = cb.measure = measure;
    FILE* o = fopen(argv[2], "wb");
    ca.multiplyBy(cb);  // Ctxt::multiplyBy
    dump(o, ca);
    Ctxt sum = ca;       // copy, then Ctxt::addCtxt
    sum.addCtxt(ca);
    dump(o, sum);
    ca.smartAutomorph(k);
    dump(o, ca);
    keys.setKeySwitchMap(m);  // BFS over Zm*: k*k is reachable in two steps
    ca.smartAutomorph((long)((unsigned __int128)k * (unsigned long)k % (unsigned long)m));
    dump(o, ca);
    // error behaviour: no matrix for this rotation -> LogicError, k outside Zm* -> InvalidArgument
    int errs = 0;
    try {
      Ctxt t = ca;
      t.smartAutomorph(m - 1);   // -1 is not a power of k
    } catch (const LogicError&) {
      errs |= 1;
    }
    try {
      Ctxt t = ca;
      t.automorph(m % 2 == 0 ? 2 : m);
    } catch (const InvalidArgument&) {
      errs |= 2;
    }
    int64_t e = errs;
    wr(o, &e, 8);
    fclose(o);
    dev->sync();
  } catch (const std::exception& ex) {
    fprintf(stderr, "exception: %s\n", ex.what());
    return 1;
  }
  return 0;
}
