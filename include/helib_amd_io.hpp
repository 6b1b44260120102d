// helib_amd_io.hpp -- Ctxt::writeTo / Ctxt::read (src/Ctxt.cpp:2584-2641) for the C++ host: a live
// helib_amd::Ctxt to and from the reference's binary layout (helib_amd_wire.hpp has the layout itself; this
// header is the bridge between its plain descriptions and the device objects of helib_amd_ctxt.hpp).
//
//   wire::describe(ct)                    Ctxt -> CtxtDesc (rows downloaded, batch element b)
//   wire::restore(desc, cc, dev, keys)    CtxtDesc -> Ctxt (rows uploaded)
//   writeTo(ct) / readCtxtFrom(bytes, ...) the 2.2.0 blob itself
//   writeToJSON(ct) / readCtxtFromJSON(text, ...) the typed JSON object (helib_amd_json.hpp)
//
// Noise and CKKS factor cross the wire as NTL xdoubles (mantissa * 2^(114 e), 2^-57 <= |mantissa| < 2^57);
// the host keeps their natural logarithms, so the conversion never overflows a double.
#pragma once
#include "helib_amd_ctxt.hpp"
#include "helib_amd_json.hpp"
#include "helib_amd_wire.hpp"

namespace helib_amd {
namespace wire {

constexpr int XD_BOUND_BITS = 114;   // NTL: xdouble's exponent counts multiples of 2^114

// exp(ln) in xdouble normal form (exponent in closed form: a crafted noise estimate cannot make this spin)
inline XDouble xdFromLn(double ln)
{
  if (ln == -INFINITY)
    return XDouble{0.0, 0};
  if (!std::isfinite(ln))
    throw IOError("xdouble: logarithm is not finite");
  const double step = XD_BOUND_BITS * std::log(2.0), half = 57 * std::log(2.0);
  double ef = std::floor((ln + half) / step);   // ln - e*step in [-half, half)
  double r = ln - ef * step;
  if (r >= half) {   // (rounding at the boundary)
    r -= step;
    ef += 1;
  } else if (r < -half) {
    r += step;
    ef -= 1;
  }
  return XDouble{std::exp(r), (int64_t)ef};
}
// a non-negative double in xdouble normal form (NTL xdouble::normalize)
inline XDouble xdOf(double x)
{
  if (x == 0.0)
    return XDouble{0.0, 0};
  if (!std::isfinite(x))
    throw IOError("xdouble: value is not finite");
  // |x| = f * 2^k with f in [0.5, 1): the normal form wants 2^-57 <= |mantissa| < 2^57
  int k = 0;
  (void)std::frexp(x, &k);
  const int64_t e = (int64_t)std::floor((double)(k + 56) / XD_BOUND_BITS);
  return XDouble{std::ldexp(x, (int)(-e * XD_BOUND_BITS)), e};
}
// what a reader accepts: finite mantissa, exponent within what a double's logarithm can carry
inline void xdCheck(const XDouble& x, const char* what)
{
  if (!std::isfinite(x.mantissa) || x.exponent > ((int64_t)1 << 40) || x.exponent < -((int64_t)1 << 40))
    throw IOError(std::string(what) + ": xdouble out of range");
}
inline double lnOf(const XDouble& x)
{
  return x.mantissa > 0 ? std::log(x.mantissa) + (double)x.exponent * XD_BOUND_BITS * std::log(2.0) : -INFINITY;
}
inline double valueOf(const XDouble& x)   // (exponents beyond a double's range saturate to 0 / inf)
{
  const int64_t e = std::max<int64_t>(-16, std::min<int64_t>(16, x.exponent));
  return std::ldexp(x.mantissa, (int)(e * XD_BOUND_BITS));
}

// parts in the reference's order: the part of 1 first, then s, then the rest (Ctxt::addPart appends and
// the constant part is created first) -- the order of SKHandle's operator<
inline CtxtDesc describe(const Ctxt& ct, int b = 0)
{
  CtxtDesc d;
  d.ptxtSpace = ct.ptxtSpace;
  d.intFactor = ct.intFactor;
  d.ptxtMag = xdOf(ct.ptxtMag);
  d.ratFactor = ct.context->ckks ? xdFromLn(ct.lnRatFactor) : XDouble{1.0, 0};
  d.noiseBound = xdFromLn(ct.lnNoise);
  d.primeSet.assign(ct.primeSet.begin(), ct.primeSet.end());
  for (auto& kv : ct.parts) {
    Part p;
    p.rows = fromPoly(kv.second, (size_t)ct.context->phim, b);
    p.handle[0] = kv.first.powerOfS;
    p.handle[1] = kv.first.powerOfX;
    p.handle[2] = 0;
    d.parts.push_back(std::move(p));
  }
  return d;
}
// Ctxt::read's checks (src/Ctxt.cpp:2620-2641, DoubleCRT::read :1530-1566): every part on exactly the
// ciphertext's prime set, primes known to the context, residues below their prime
inline Ctxt restore(const CtxtDesc& d, const ChainContext& cc, const Context& dev, const KeySet& keys)
{
  xdCheck(d.ptxtMag, "ptxtMag");
  xdCheck(d.ratFactor, "ratFactor");
  xdCheck(d.noiseBound, "noiseBound");
  Ctxt ct(cc, dev, keys);
  ct.ptxtSpace = d.ptxtSpace;
  ct.intFactor = d.intFactor;
  ct.ptxtMag = valueOf(d.ptxtMag);
  ct.lnRatFactor = cc.ckks ? lnOf(d.ratFactor) : 0.0;
  ct.lnNoise = lnOf(d.noiseBound);
  for (long i : d.primeSet) {
    if (i < 0 || (size_t)i >= cc.primes.size())
      throw IOError("Stream does not contain subset of the context's primes");
    ct.primeSet.insert((int)i);
  }
  const std::vector<long> want(ct.primeSet.begin(), ct.primeSet.end());   // ascending, as DoubleCRT::read leaves a part's
  for (auto& p : d.parts) {
    if (p.rows.idx != want)
      throw IOError("Ciphertext part's index set does not match prime set");
    if (p.rows.n != (size_t)cc.phim)
      throw IOError("Data not valid: d.map[i].length() != phim");
    for (size_t r = 0; r < p.rows.idx.size(); r++) {
      const uint64_t q = cc.primes[(size_t)p.rows.idx[r]];
      for (size_t j = 0; j < p.rows.n; j++)
        if (p.rows.data[r * p.rows.n + j] >= q)
          throw IOError("this->map[i][j] invalid: must be between 0 and context.ithPrime(i)");
    }
    IndexSet idx(p.rows.idx.begin(), p.rows.idx.end());
    DoubleCRT poly(dev, idx, 1, DoubleCRT::Uninitialized{});
    poly.setRows(toRows(p.rows, idx));
    SKHandle h{p.handle[0], p.handle[1]};
    if (!ct.parts.emplace(h, std::move(poly)).second)
      throw IOError("two ciphertext parts with one handle");
  }
  return ct;
}

}  // namespace wire

// Ctxt::writeTo: the reference's 2.2.0 binary object (header, "|CX[" ... "]CX|")
inline std::string writeTo(const Ctxt& ct, int b = 0)
{
  wire::Writer w;
  wire::write(w, wire::describe(ct, b));
  return w.out;
}
// Ctxt::read: `used` (optional) receives the number of bytes consumed
inline Ctxt readCtxtFrom(const void* data, size_t size, const ChainContext& cc, const Context& dev, const KeySet& keys,
                         size_t* used = nullptr)
{
  wire::Reader rd(data, size);
  wire::CtxtDesc d = wire::readCtxt(rd);
  if (used)
    *used = rd.pos;
  return wire::restore(d, cc, dev, keys);
}

// Ctxt::writeToJSON / Ctxt::readJSON (src/Ctxt.cpp:2642-2712): the typed JSON object as text
inline std::string writeToJSON(const Ctxt& ct, int b = 0) { return wire::toJson(wire::describe(ct, b)).dump(); }
inline Ctxt readCtxtFromJSON(const std::string& text, const ChainContext& cc, const Context& dev, const KeySet& keys)
{
  return wire::restore(wire::ctxtFromJson(wire::Json::parse(text)), cc, dev, keys);
}

}  // namespace helib_amd
