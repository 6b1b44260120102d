// dump.cpp -- built against GENUINE HElib 2.2.0 (with NTL and GMP), NOT part of this repository's build:
// writes everything tests/test_against_helib_dump.py needs to pin this engine against the real library on the
// parameters of benchmarks/bgv_basic.cpp -- the one thing that cannot be produced in the build container, which
// has no NTL: the power-of-two roots come out of NTL's PRG (src/CModulus.cpp:93-98, 118-119) and
// the chain decisions out of NTL::ProbPrime / xdouble arithmetic.
//
//   tools/helib_dump/build.sh /path/to/helib_install && ./helib_dump OUTDIR [m p r bits c]
//   HELIB_DUMP_DIR=OUTDIR python -m pytest tests/test_against_helib_dump.py
//
// Files (binary = HElib's own wire format, src/Ctxt.cpp:2584-2641, src/DoubleCRT.cpp writeTo, src/keys.cpp writeTo):
//   params.json     m, p, r, bits, c, HElib version
//   chain.json      primes in chain order (Context::ithPrime), Cmodulus::getRoot() of each, the small / ctxt / special
//                   prime sets, the digits
//   modsizes.bin    the ModuliSizes table as ModuliSizes::write puts it: count, then (double size, IndexSet) per row
//   context.bin, pubkey.bin, seckey.bin
//   ksw_a_<s>_<x>.bin   per key-switching matrix from (s^s)(X^x): its D pseudorandom a-columns as DoubleCRT blobs,
//                   regenerated from prgSeed exactly as Ctxt::keySwitchDigits does (src/Ctxt.cpp:191-206) -- the PRG
//                   stream is NTL's and cannot be re-derived without it
//   ptxt1.json, ptxt2.json     the plaintext polynomials (coefficients)
//   ct1.bin, ct2.bin           their encryptions
//   prod.bin                   ct1 after multiplyBy(ct2)         (benchmarks/bgv_basic.cpp:158-164)
//   prod2.bin                  prod after multiplyBy(prod)       (a level-2 product: operands carry special primes)
//   rot.bin, rot.json          prod after smartAutomorph(k), and k
//   decisions.json             primeSet / ptxtSpace / noise bound / capacity of every ciphertext above (intFactor: in the .bin)
#include <helib/helib.h>

#include <fstream>
#include <iomanip>
#include <sstream>

using namespace helib;

static std::string setToJson(const IndexSet& s)
{
  std::ostringstream os;
  os << "[";
  bool first = true;
  for (long i : s) {
    os << (first ? "" : ",") << i;
    first = false;
  }
  os << "]";
  return os.str();
}

static std::string ctxtInfo(const char* name, const Ctxt& c)
{
  std::ostringstream os;
  os << std::setprecision(17);
  // (intFactor has no accessor: the test reads it from the ciphertext's binary form, where writeTo puts it)
  os << "\"" << name << "\": {\"primeSet\": " << setToJson(c.getPrimeSet()) << ", \"ptxtSpace\": " << c.getPtxtSpace()
     << ", \"lnNoiseBound\": " << NTL::conv<double>(log(c.getNoiseBound())) << ", \"capacity\": " << c.capacity() << "}";
  return os.str();
}

static void writeBin(const std::string& path, const std::function<void(std::ostream&)>& f)
{
  std::ofstream os(path, std::ios::binary);
  f(os);
}

int main(int argc, char** argv)
{
  if (argc < 2) {
    std::cerr << "usage: helib_dump OUTDIR [m p r bits c]   (defaults: 32768 65537 1 950 3)\n";
    return 2;
  }
  const std::string dir = argv[1];
  const long m = argc > 2 ? atol(argv[2]) : 32768, p = argc > 3 ? atol(argv[3]) : 65537, r = argc > 4 ? atol(argv[4]) : 1,
             bits = argc > 5 ? atol(argv[5]) : 950, c = argc > 6 ? atol(argv[6]) : 3;

  Context context = ContextBuilder<BGV>().m(m).p(p).r(r).bits(bits).c(c).build();
  SecKey secretKey(context);
  secretKey.GenSecKey();            // incl. the s^2 -> s relinearisation matrix
  addSome1DMatrices(secretKey);     // the rotation matrices, as benchmarks/bgv_common.h:60-75
  const PubKey& publicKey = secretKey;

  {
    std::ofstream os(dir + "/params.json");
    os << "{\"m\": " << m << ", \"p\": " << p << ", \"r\": " << r << ", \"bits\": " << bits << ", \"c\": " << c
       << ", \"helib_version\": \"" << version::asString << "\", \"phim\": " << context.getPhiM() << "}\n";
  }
  {
    std::ofstream os(dir + "/chain.json");
    os << std::setprecision(17);
    os << "{\"primes\": [";
    for (long i = 0; i < context.numPrimes(); i++)
      os << (i ? "," : "") << context.ithPrime(i);
    os << "],\n \"roots\": [";
    for (long i = 0; i < context.numPrimes(); i++)
      os << (i ? "," : "") << context.ithModulus(i).getRoot();
    os << "],\n \"smallPrimes\": " << setToJson(context.getSmallPrimes()) << ",\n \"ctxtPrimes\": "
       << setToJson(context.getCtxtPrimes()) << ",\n \"specialPrimes\": " << setToJson(context.getSpecialPrimes())
       << ",\n \"digits\": [";
    for (size_t d = 0; d < context.getDigits().size(); d++)
      os << (d ? "," : "") << setToJson(context.getDigit((long)d));
    os << "]}\n";
  }
  // the ModuliSizes table in its own binary form (src/primeChain.cpp:55-59, 353-358): count, then (double size, IndexSet)
  writeBin(dir + "/modsizes.bin", [&](std::ostream& os) { context.getModSizeTable().write(os); });
  writeBin(dir + "/context.bin", [&](std::ostream& os) { context.writeTo(os); });
  writeBin(dir + "/pubkey.bin", [&](std::ostream& os) { publicKey.writeTo(os); });
  writeBin(dir + "/seckey.bin", [&](std::ostream& os) { secretKey.writeTo(os); });

  // the a-columns of every key-switching matrix, regenerated from its seed as Ctxt::keySwitchDigits does
  for (const KeySwitch& W : publicKey.keySWlist()) {
    if (W.isDummy())
      continue;
    std::ostringstream name;
    name << dir << "/ksw_a_" << W.fromKey.getPowerOfS() << "_" << W.fromKey.getPowerOfX() << ".bin";
    std::ofstream os(name.str(), std::ios::binary);
    DoubleCRT ai(context, context.getCtxtPrimes() | context.getSpecialPrimes());
    RandomState state;   // restores NTL's PRG on destruction
    NTL::SetSeed(W.prgSeed);
    for (size_t i = 0; i < W.b.size(); i++) {
      ai.randomize();
      ai.writeTo(os);
    }
  }

  // two plaintext polynomials with coefficients in [0, p^r), encrypted as polynomials (no slot encoding on this path)
  const long phim = context.getPhiM(), p2r = context.getPPowR();
  NTL::ZZX pt[2];
  for (int j = 0; j < 2; j++) {
    std::ofstream os(dir + (j ? "/ptxt2.json" : "/ptxt1.json"));
    os << "[";
    for (long i = 0; i < phim; i++) {
      const long v = NTL::RandomBnd(p2r);
      NTL::SetCoeff(pt[j], i, v);
      os << (i ? "," : "") << v;
    }
    os << "]\n";
  }
  Ctxt ct1(publicKey), ct2(publicKey);
  publicKey.Encrypt(ct1, pt[0]);
  publicKey.Encrypt(ct2, pt[1]);
  writeBin(dir + "/ct1.bin", [&](std::ostream& os) { ct1.writeTo(os); });
  writeBin(dir + "/ct2.bin", [&](std::ostream& os) { ct2.writeTo(os); });

  Ctxt prod(ct1);
  prod.multiplyBy(ct2);
  writeBin(dir + "/prod.bin", [&](std::ostream& os) { prod.writeTo(os); });
  Ctxt prod2(prod);
  prod2.multiplyBy(prod);
  writeBin(dir + "/prod2.bin", [&](std::ostream& os) { prod2.writeTo(os); });

  // one rotation step that has a matrix: the first generator of Z_m^* / <p>, or 3 for a power of two
  const long k = context.getZMStar().numOfGens() > 0 ? context.getZMStar().ZmStarGen(0) : 3;
  Ctxt rot(prod);
  rot.smartAutomorph(k);
  writeBin(dir + "/rot.bin", [&](std::ostream& os) { rot.writeTo(os); });
  {
    std::ofstream os(dir + "/rot.json");
    os << "{\"k\": " << k << "}\n";
  }
  {
    std::ofstream os(dir + "/decisions.json");
    os << "{" << ctxtInfo("ct1", ct1) << ",\n " << ctxtInfo("ct2", ct2) << ",\n " << ctxtInfo("prod", prod) << ",\n "
       << ctxtInfo("prod2", prod2) << ",\n " << ctxtInfo("rot", rot) << "}\n";
  }
  // self-check on the way out: the dump is of a working computation
  NTL::ZZX dec;
  secretKey.Decrypt(dec, prod);
  std::cout << "helib_dump: wrote " << dir << " (m=" << m << " p=" << p << " bits=" << bits << ", " << context.numPrimes()
            << " primes, decrypt(prod) has degree " << NTL::deg(dec) << ")\n";
  return 0;
}
