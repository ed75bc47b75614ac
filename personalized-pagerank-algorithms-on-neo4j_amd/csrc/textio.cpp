// textio.cpp — the reference's preprocessing result format: one text file per source node, lines
// "<nodeId>\t<Double.toString(ppr)>\n" (Base_Whole_Graph.java:118-126,152-156; readers :167-186).
#include <sys/stat.h>

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

#include "common.hpp"

namespace pprhip {

// java.lang.Double.toString: shortest decimal that round-trips, plain notation for
// 1e-3 <= |d| < 1e7 and "d.dddE[-]n" otherwise, always at least one digit after the point.
std::string java_double_to_string(double d) {
  if (std::isnan(d)) return "NaN";
  if (std::isinf(d)) return d > 0 ? "Infinity" : "-Infinity";
  if (d == 0.0) return std::signbit(d) ? "-0.0" : "0.0";
  char buf[64];
  int prec = 1;
  for (; prec <= 17; ++prec) {
    snprintf(buf, sizeof buf, "%.*e", prec - 1, d);
    if (strtod(buf, nullptr) == d) break;
  }
  // buf = [-]D[.DDDD]e[+-]XX
  std::string s(buf);
  const bool neg = s[0] == '-';
  if (neg) s.erase(0, 1);
  const size_t epos = s.find('e');
  std::string mant = s.substr(0, epos);
  const int exp10 = atoi(s.c_str() + epos + 1);
  std::string digits;
  for (char c : mant)
    if (c != '.') digits.push_back(c);
  while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
  std::string out;
  const double ad = std::fabs(d);
  if (ad >= 1e-3 && ad < 1e7) {
    if (exp10 >= 0) {
      std::string ip = digits.substr(0, std::min(digits.size(), (size_t)exp10 + 1));
      while (ip.size() < (size_t)exp10 + 1) ip.push_back('0');
      std::string fp = digits.size() > (size_t)exp10 + 1 ? digits.substr(exp10 + 1) : "0";
      out = ip + "." + fp;
    } else {
      out = "0." + std::string((size_t)(-exp10 - 1), '0') + digits;
    }
  } else {
    out = digits.substr(0, 1) + "." + (digits.size() > 1 ? digits.substr(1) : "0") + "E" + std::to_string(exp10);
  }
  return neg ? "-" + out : out;
}

}  // namespace pprhip

extern "C" {

int pprhip_format_double(double d, char* buf, size_t cap) {
  if (!buf || cap == 0) return PPRHIP_ERR_INVALID;
  const std::string s = pprhip::java_double_to_string(d);
  if (s.size() + 1 > cap) return PPRHIP_ERR_INVALID;
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}

int pprhip_index_write_dir(const pprhip_index_t* ix, const char* dir) {
  using namespace pprhip;
  if (!ix || !dir) {
    set_error("pprhip_index_write_dir: null argument");
    return PPRHIP_ERR_INVALID;
  }
  uint32_t n = 0;
  uint64_t entries = 0;
  const uint64_t* off = nullptr;
  const int32_t* tg = nullptr;
  const double* vl = nullptr;
  PPRHIP_TRY(pprhip_index_info(ix, &n, &entries));
  PPRHIP_TRY(pprhip_index_arrays(ix, &off, &tg, &vl));
  // mkdir -p
  std::string path(dir);
  for (size_t i = 1; i <= path.size(); ++i) {
    if (i == path.size() || path[i] == '/') {
      std::string sub = path.substr(0, i);
      if (mkdir(sub.c_str(), 0777) != 0 && errno != EEXIST) {
        set_error("cannot create directory %s: %s", sub.c_str(), strerror(errno));
        return PPRHIP_ERR_IO;
      }
    }
  }
  for (uint32_t v = 0; v < n; ++v) {
    if (off[v] == off[v + 1]) continue;  // the reference only writes sources that own entries (:113)
    std::string fn = path + "/" + std::to_string(v) + ".txt";
    FILE* f = fopen(fn.c_str(), "w");
    if (!f) {
      set_error("cannot write %s: %s", fn.c_str(), strerror(errno));
      return PPRHIP_ERR_IO;
    }
    for (uint64_t i = off[v]; i < off[v + 1]; ++i)
      fprintf(f, "%d\t%s\n", tg[i], java_double_to_string(vl[i]).c_str());
    fclose(f);
  }
  return PPRHIP_OK;
}

}  // extern "C"
