// Library plumbing: version, arch string, thread-local error text.
#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ubv_common.h"

namespace ubv {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ubv

// ---- optional per-kernel timing with HIP events on the launch stream ---------------------------------
namespace ubv {
namespace {
struct ProfRec { char name[96]; double bytes; hipEvent_t e0, e1; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
bool g_prof_ops_only = false;   // level 2: whole operators only — the scopes of the kernels inside an op put their own
                                // event records between its launches and lengthen what the op-level scope measures
}  // namespace

ProfScope::ProfScope(const char* name, hipStream_t st, double bytes) : idx_(-1), st_(st) {
  if (!g_prof_on) return;
  if (g_prof_ops_only && strncmp(name, "bev_lift_fwd<", 13) != 0 && strstr(name, "_op<") == nullptr) return;
  ProfRec r;
  snprintf(r.name, sizeof(r.name), "%s", name);
  r.bytes = bytes;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
  (void)hipEventRecord(r.e0, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(r);
  idx_ = (long)g_prof.size() - 1;
}

ProfScope::~ProfScope() {
  if (idx_ < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_prof[idx_].e1, st_);
}
}  // namespace ubv

extern "C" int ubv_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(ubv::g_prof_mu);
  for (auto& r : ubv::g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  ubv::g_prof.clear();
  ubv::g_prof_on = on != 0;
  ubv::g_prof_ops_only = on == 2;
  return UBV_OK;
}

extern "C" int64_t ubv_profile_read(char* out, int64_t capacity) {
  std::lock_guard<std::mutex> lk(ubv::g_prof_mu);
  struct Agg { long n = 0; double ms = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : ubv::g_prof) {
    if (hipEventSynchronize(r.e1) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    Agg& a = agg[r.name];
    a.n += 1; a.ms += ms; a.bytes += r.bytes;
  }
  std::string text;
  char line[256];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s\t%ld\t%.6f\t%.1f\n", kv.first.c_str(), kv.second.n, kv.second.ms,
             kv.second.bytes / (double)kv.second.n);
    text += line;
  }
  if (out != nullptr && capacity > 0) {
    const size_t n = text.size() < (size_t)capacity - 1 ? text.size() : (size_t)capacity - 1;
    memcpy(out, text.data(), n);
    out[n] = 0;
  }
  return (int64_t)text.size() + 1;
}

extern "C" int ubv_version(void) { return 100; }   // 0.1.0
extern "C" const char* ubv_last_error(void) { return ubv::g_err; }
extern "C" const char* ubv_arch(void) { return "gfx950"; }
