// CPU check of include/helib_amd_wire.hpp (no device, no library): parse a binary object, write it
// back, compare the bytes, and print what was read as JSON for tests/test_wire.py to compare with
// helib_amd/wire.py's reading of the same bytes.
//   wire_test legacy  <file>     the reference's whole fixture: context + public key + secret key
//   wire_test seckey  <file>     a 2.2.0 SecKey blob          wire_test skonly <file>   (sk_only)
//   wire_test pubkey  <file>     a 2.2.0 PubKey blob          wire_test ctxt   <file>   a 2.2.0 Ctxt
//   wire_test context <file>     a 2.2.0 Context              wire_test errors <file>   a 2.2.0 SecKey
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "helib_amd_wire.hpp"

using namespace helib_amd::wire;

static std::string slurp(const char* path)
{
  std::ifstream f(path, std::ios::binary);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
static void jl(const char* name, const std::vector<long>& v, bool last = false)
{
  printf("\"%s\": [", name);
  for (size_t i = 0; i < v.size(); i++)
    printf("%s%ld", i ? ", " : "", v[i]);
  printf("]%s", last ? "" : ", ");
}
static void jctx(const ContextDesc& c)
{
  printf("\"context\": {\"m\": %ld, \"p\": %ld, \"r\": %ld, ", c.m, c.p, c.r);
  jl("gens", c.gens);
  jl("ords", c.ords);
  jl("qs", c.qs);
  jl("smallPrimes", c.smallPrimes);
  jl("specialPrimes", c.specialPrimes);
  printf("\"stdev\": %.17g, \"scale\": %.17g, \"ndigits\": %zu, \"hwt\": %ld}, ", c.stdev.mantissa, c.scale,
         c.digits.size(), c.hwt_param);
}
static void jkey(const PubKeyDesc& k)
{
  jctx(k.context);
  printf("\"handles\": [");
  for (size_t i = 0; i < k.keySwitching.size(); i++)
    printf("%s[%ld, %ld, %ld]", i ? ", " : "", k.keySwitching[i].fromKey[0], k.keySwitching[i].fromKey[1],
           k.keySwitching[i].fromKey[2]);
  printf("], \"skBounds\": [");
  for (size_t i = 0; i < k.skBounds.size(); i++)
    printf("%s%.17g", i ? ", " : "", k.skBounds[i]);
  printf("], ");
  jl("KS_strategy", k.KS_strategy);
  if (!k.keySwitchMap.empty()) {
    jl("keySwitchMap", k.keySwitchMap[0]);
    jl("keySwitchMap_bfs", keySwitchMapOf(k.context.m, k.keySwitching));
  }
  printf("\"recryptKeyID\": %ld, \"pubEncrKey_noise\": [%.17g, %ld], \"pubEncrKey_parts\": %zu, ", k.recryptKeyID,
         k.pubEncrKey.noiseBound.mantissa, (long)k.pubEncrKey.noiseBound.exponent, k.pubEncrKey.parts.size());
  jl("pubEncrKey_primeSet", k.pubEncrKey.primeSet);
}

struct MockPoly {   // what fromPoly / toRows need of a DoubleCRT
  std::vector<int> idx;
  int b;
  std::vector<uint64_t> rows;
  std::vector<int> getIndexSet() const { return idx; }
  std::vector<uint64_t> getRows() const { return rows; }
  int batch() const { return b; }
};

int main(int argc, char** argv)
{
  if (argc < 3)
    return 2;
  std::string mode = argv[1], in = slurp(argv[2]);
  try {
    Reader rd(in.data(), in.size());
    Writer w;
    printf("{");
    if (mode == "legacy") {
      ContextDesc c = readContext(rd, true);
      size_t a = rd.pos;
      PubKeyDesc pk = readPubKey(rd, true, &c);
      size_t b = rd.pos;
      SecKeyDesc sk = readSecKey(rd, true, false, &c);
      write(w, c, true);
      write(w, pk, true);
      write(w, sk, true);
      jkey(pk);
      printf("\"offsets\": [%zu, %zu, %zu], \"nsk\": %zu, ", a, b, rd.pos, sk.sKeys.size());
      jl("sk_idx", sk.sKeys.at(0).idx);
      printf("\"sk_row0\": [");
      for (size_t j = 0; j < sk.sKeys[0].n; j++)
        printf("%s%llu", j ? ", " : "", (unsigned long long)sk.sKeys[0].data[j]);
      printf("], \"embedded_pk_equal\": %s, ", in.substr(a, b - a) == in.substr(b + 4, b - a) ? "true" : "false");
    } else if (mode == "seckey" || mode == "skonly") {
      SecKeyDesc sk = readSecKey(rd, false, mode == "skonly");
      write(w, sk, false, mode == "skonly");
      if (mode == "seckey")
        jkey(sk);
      else
        jctx(sk.context);
      printf("\"nsk\": %zu, ", sk.sKeys.size());
    } else if (mode == "pubkey") {
      PubKeyDesc pk = readPubKey(rd);
      write(w, pk);
      jkey(pk);
    } else if (mode == "context") {
      ContextDesc c = readContext(rd);
      write(w, c);
      jctx(c);
    } else if (mode == "ctxt") {
      CtxtDesc c = readCtxt(rd);
      write(w, c);
      printf("\"ptxtSpace\": %ld, \"intFactor\": %ld, \"parts\": %zu, \"ratFactor\": [%.17g, %ld], ", c.ptxtSpace,
             c.intFactor, c.parts.size(), c.ratFactor.mantissa, (long)c.ratFactor.exponent);
      jl("primeSet", c.primeSet);
      // through fromPoly / toRows with a stand-in polynomial: rows out of order, two batch elements
      if (!c.parts.empty()) {
        const Rows& r = c.parts[0].rows;
        MockPoly mp;
        mp.b = 2;
        for (size_t k = r.idx.size(); k-- > 0;)
          mp.idx.push_back((int)r.idx[k]);   // descending
        for (size_t k = r.idx.size(); k-- > 0;)
          for (int bb = 0; bb < 2; bb++)
            for (size_t j = 0; j < r.n; j++)
              mp.rows.push_back(bb == 1 ? r.data[k * r.n + j] : ~r.data[k * r.n + j]);
        Rows back = fromPoly(mp, r.n, 1);
        bool ok = back.idx == r.idx && back.data == r.data;
        std::vector<uint64_t> rr = toRows(r, mp.idx);
        for (size_t k = 0; k < r.idx.size(); k++)
          for (size_t j = 0; j < r.n; j++)
            ok = ok && rr[k * r.n + j] == r.data[(r.idx.size() - 1 - k) * r.n + j];
        printf("\"poly_round_trip\": %s, ", ok ? "true" : "false");
      }
    } else if (mode == "errors") {
      auto expect = [&](std::string bytes, const char* what) {
        try {
          Reader r2(bytes.data(), bytes.size());
          readSecKey(r2);
        } catch (const IOError& e) {
          return std::string(e.what()).find(what) != std::string::npos;
        }
        return false;
      };
      std::string bad = in;
      bad[0] = 'X';
      bool e1 = expect(bad, "header");
      bad = in;
      bad[12] = 20;
      bool e2 = expect(bad, "structId");
      bad = in;
      bad[bad.size() - 2] = 'X';
      bool e3 = expect(bad, "eye catcher");
      bool e4 = expect(in.substr(0, in.size() / 2), "end of stream");
      ContextDesc other = readSecKey(rd).context;
      other.m += 1;
      bool e5 = false;
      try {
        Reader r3(in.data(), in.size());
        readSecKey(r3, false, false, &other);
      } catch (const IOError& e) {
        e5 = std::string(e.what()) == "Context mismatch";
      }
      printf("\"errors\": [%d, %d, %d, %d, %d], ", e1, e2, e3, e4, e5);
      w.out = in;
    } else {
      return 2;
    }
    printf("\"consumed\": %s, \"same_bytes\": %s}\n", rd.done() ? "true" : "false", w.out == in ? "true" : "false");
  } catch (const std::exception& e) {
    fprintf(stderr, "exception: %s\n", e.what());
    return 1;
  }
  return 0;
}
