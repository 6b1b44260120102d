// include/helib_amd_json.hpp against helib_amd/wire.py (tests/test_wire.py drives it; no device involved):
//   json_test <kind> j2b <in.json> <out.bin>   JSON text -> description -> the 2.2.0 binary object in <out.bin>,
//                                              and the description's JSON again on stdout
//   json_test <kind> b2j <in.bin>              binary object -> description -> JSON on stdout
//   json_test errors                           malformed inputs: one line "raised <n> of <total>"
// kind: ctxt | keyswitch | context | pubkey | seckey | skonly
#include <cstdio>
#include <fstream>
#include <sstream>

#include "helib_amd_json.hpp"

using namespace helib_amd::wire;

static std::string slurp(const char* path)
{
  std::ifstream f(path, std::ios::binary);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
template <class F>
static int raises(F f)
{
  try {
    f();
  } catch (const IOError&) {
    return 1;
  }
  return 0;
}

int main(int argc, char** argv)
{
  try {
    if (argc >= 2 && std::string(argv[1]) == "errors") {
      int n = 0, total = 0;
      const char* bad[] = {"", "{", "[1,2", "{\"a\" 1}", "{\"a\":1,}", "nul", "\"abc", "{\"a\":1} x", "[1 2]", "-", "1e",
                           "{\"type\":\"Ctxt\",\"HElibVersion\":\"2.2.0\",\"serializationVersion\":\"0.0.1\"}"};
      for (const char* t : bad) {
        total++;
        n += raises([&] { ctxtFromJson(Json::parse(t)); });
      }
      // type / version mismatches and the part-versus-primeSet check
      CtxtDesc c;
      c.ptxtSpace = 2;
      c.primeSet = {0, 1};
      Part p;
      p.rows.idx = {0, 1};
      p.rows.n = 2;
      p.rows.data = {1, 2, 3, 4};
      c.parts.push_back(p);
      Json good = toJson(c);
      total++, n += 1 - raises([&] { ctxtFromJson(good); });                 // (the good one must pass)
      Json j = good;
      j["type"] = Json::string("KeySwitch");
      total++, n += raises([&] { ctxtFromJson(j); });
      j = good;
      j["HElibVersion"] = Json::string("2.1.0");
      total++, n += raises([&] { ctxtFromJson(j); });
      j = good;
      j["serializationVersion"] = Json::string("0.0.2");
      total++, n += raises([&] { ctxtFromJson(j); });
      j = good;
      j["content"]["primeSet"] = jsonOf(std::vector<long>{0, 1, 2});
      total++, n += raises([&] { ctxtFromJson(j); });
      j = good;
      j["content"]["parts"].a[0]["DoubleCRT"]["map"].a[1].a.pop_back();      // ragged rows
      total++, n += raises([&] { ctxtFromJson(j); });
      j = good;
      j["content"]["parts"].a[0]["DoubleCRT"]["map"].a[0].a[0] = Json::integer(-1);
      total++, n += raises([&] { ctxtFromJson(j); });
      total++, n += raises([&] { bytesOfDecimal("12x"); });
      total++, n += raises([&] { bytesOfDecimal(std::string(5000, '9')); });
      total++, n += raises([&] { Json::parse("1e300").asInt(); });
      total++, n += raises([&] { Json::parse(std::string(100, '[')); });
      total++, n += (decimalOf(bytesOfDecimal("340282366920938463463374607431768211457")) ==
                     "340282366920938463463374607431768211457");
      total++, n += (decimalOf({}) == "0" && bytesOfDecimal("0").empty());
      printf("raised %d of %d\n", n, total);
      return n == total ? 0 : 1;
    }
    if (argc < 4)
      return 2;
    const std::string kind = argv[1], dir = argv[2], in = slurp(argv[3]);
    Writer w;
    Json out;
    if (dir == "j2b") {
      Json j = Json::parse(in);
      if (kind == "ctxt") {
        CtxtDesc d = ctxtFromJson(j);
        write(w, d);
        out = toJson(d);
      } else if (kind == "keyswitch") {
        KeySwitchDesc d = keySwitchFromJson(j);
        write(w, d);
        out = toJson(d);
      } else if (kind == "context") {
        ContextDesc d = contextFromJson(j);
        write(w, d);
        out = toJson(d);
      } else if (kind == "pubkey") {
        PubKeyDesc d = pubKeyFromJson(j);
        write(w, d);
        out = toJson(d);
      } else if (kind == "seckey" || kind == "skonly") {
        SecKeyDesc d = secKeyFromJson(j, kind == "skonly");
        write(w, d, false, kind == "skonly");
        out = toJson(d, kind == "skonly");
      } else {
        return 2;
      }
      if (argc >= 5) {
        std::ofstream f(argv[4], std::ios::binary);
        f.write(w.out.data(), (std::streamsize)w.out.size());
      }
    } else {
      Reader rd(in.data(), in.size());
      if (kind == "ctxt")
        out = toJson(readCtxt(rd));
      else if (kind == "keyswitch")
        out = toJson(readKeySwitch(rd));
      else if (kind == "context")
        out = toJson(readContext(rd));
      else if (kind == "pubkey")
        out = toJson(readPubKey(rd));
      else if (kind == "seckey" || kind == "skonly")
        out = toJson(readSecKey(rd, false, kind == "skonly"), kind == "skonly");
      else
        return 2;
      if (!rd.done())
        throw IOError("trailing bytes");
    }
    std::string text = out.dump();
    fwrite(text.data(), 1, text.size(), stdout);
    printf("\n");
  } catch (const std::exception& ex) {
    fprintf(stderr, "exception: %s\n", ex.what());
    return 1;
  }
  return 0;
}
