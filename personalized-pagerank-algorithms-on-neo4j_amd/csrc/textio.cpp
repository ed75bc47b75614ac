// textio.cpp — the reference's preprocessing result format: one text file per source node, lines
// "<nodeId>\t<Double.toString(ppr)>\n" (Base_Whole_Graph.java:118-126,152-156; readers :167-186).
#include <sys/stat.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <charconv>
#include <cmath>
#include <mutex>
#include <thread>
#include <vector>
#include <cstdlib>
#include <cstring>
#include <string>

#include "common.hpp"

namespace pprhip {
namespace detail {
unsigned host_threads();  // lift.cpp
}

// java.lang.Double.toString: shortest decimal that round-trips, plain notation for
// 1e-3 <= |d| < 1e7 and "d.dddE[-]n" otherwise, always at least one digit after the point.
std::string java_double_to_string(double d) {
  if (std::isnan(d)) return "NaN";
  if (std::isinf(d)) return d > 0 ? "Infinity" : "-Infinity";
  if (d == 0.0) return std::signbit(d) ? "-0.0" : "0.0";
  char buf[64];
  // The first precision whose correctly rounded decimal reads back as d.  No decimal shorter than the shortest
  // round-tripping one (std::to_chars) can do that, so the search starts at its length: one or two trials instead of
  // the sixteen a PPR value used to take (5 us per value: minutes for an All-Pair index of 3e7 entries).
  int prec = 1;
  {
    char sh[64];
    const auto r = std::to_chars(sh, sh + sizeof sh, d, std::chars_format::scientific);
    if (r.ec == std::errc()) {
      int digits = 0;
      for (const char* c = sh; c < r.ptr && *c != 'e'; ++c) digits += (*c >= '0' && *c <= '9') ? 1 : 0;
      prec = std::max(1, std::min(17, digits));
    }
  }
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
  // One file per source that owns entries (the reference's layout: millions of small files for a large graph).  The
  // lines are formatted on all host threads, a range of sources at a time into one buffer; the files themselves are
  // created by one thread at a time - creating files in ONE directory from several threads at once serialises on the
  // directory and ran six times slower than a single thread (8 threads: 42 s against 7 s for 194 K files).
  const unsigned T = (entries < (1u << 14)) ? 1u : detail::host_threads();
  const unsigned parts = T == 1 ? 1u : std::max(T * 16u, (unsigned)std::min<uint64_t>(entries >> 16, 4096));
  std::vector<std::string> errs(parts);
  std::atomic<unsigned> next{0};
  std::mutex dir_mu;
  auto work = [&]() {
    std::string text, fn;
    std::vector<size_t> end;  // end[i]: where the text of the range's i-th written source ends
    std::vector<uint32_t> who;
    for (unsigned p = next.fetch_add(1); p < parts; p = next.fetch_add(1)) {
      const uint32_t v_lo = (uint32_t)((uint64_t)n * p / parts), v_hi = (uint32_t)((uint64_t)n * (p + 1) / parts);
      try {
        text.clear();
        end.clear();
        who.clear();
        for (uint32_t v = v_lo; v < v_hi; ++v) {
          if (off[v] == off[v + 1]) continue;  // the reference only writes sources that own entries (:113)
          for (uint64_t i = off[v]; i < off[v + 1]; ++i) {
            text += std::to_string(tg[i]);
            text += '\t';
            text += java_double_to_string(vl[i]);
            text += '\n';
          }
          who.push_back(v);
          end.push_back(text.size());
        }
        std::lock_guard<std::mutex> lk(dir_mu);
        size_t begin = 0;
        for (size_t j = 0; j < who.size() && errs[p].empty(); ++j) {
          fn = path + "/" + std::to_string(who[j]) + ".txt";
          const int fd = open(fn.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
          bool ok = fd >= 0;
          for (size_t done = begin; ok && done < end[j];) {
            const ssize_t w = write(fd, text.data() + done, end[j] - done);
            if (w < 0 && errno == EINTR) continue;
            ok = w > 0;
            if (ok) done += (size_t)w;
          }
          const int saved = errno;
          if (fd >= 0 && close(fd) != 0 && ok) ok = false;
          if (!ok) errs[p] = "cannot write " + fn + ": " + strerror(saved ? saved : errno);
          begin = end[j];
        }
      } catch (const std::bad_alloc&) {
        errs[p] = "out of host memory";
      }
    }
  };
  {
    std::vector<std::thread> th;
    for (unsigned w = 1; w < std::min(T, parts); ++w) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
  }
  for (const std::string& e : errs)
    if (!e.empty()) {
      set_error("%s", e.c_str());
      return PPRHIP_ERR_IO;
    }
  return PPRHIP_OK;
}

}  // extern "C"
