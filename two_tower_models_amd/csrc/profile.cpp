// Optional per-kernel timing with HIP events recorded on the launch stream (what
// bench.py's `roofline` object is computed from).  Off by default: zero overhead.
#include <hip/hip_runtime.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "common.hpp"

namespace tt {
namespace {
struct Slot {
  std::string name;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};
bool g_on = false;
bool g_paused = false;  // tt_profile_pause: launches the library makes for its OWN measurements are not the caller's
std::string g_only;  // "" = every bracketed kernel; else ",name,name," (tt_profile_filter)
std::vector<Slot> g_slots;
Slot& slot_for(const char* name) {
  for (auto& s : g_slots)
    if (s.name == name) return s;
  g_slots.push_back(Slot{name, {}});
  return g_slots.back();
}
void clear_all() {
  for (auto& s : g_slots)
    for (auto& p : s.ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  g_slots.clear();
}
}  // namespace

ProfScope::ProfScope(const char* name, hipStream_t st) : st_(st), slot_(-1), idx_(-1) {
  if (!g_on || g_paused) return;
  if (!g_only.empty() && g_only.find("," + std::string(name) + ",") == std::string::npos) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
  Slot& s = slot_for(name);
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  (void)hipEventRecord(a, st);
  s.ev.emplace_back(a, b);
  slot_ = (int)(&s - &g_slots[0]);
  idx_ = (int)s.ev.size() - 1;
}
ProfScope::~ProfScope() {
  if (slot_ < 0) return;
  (void)hipEventRecord(g_slots[slot_].ev[idx_].second, st_);
}
}  // namespace tt

extern "C" int tt_profile_enable(int on) {
  tt::clear_all();
  tt::g_on = on != 0;
  return 0;
}

extern "C" int tt_profile_pause(int paused) {
  tt::g_paused = paused != 0;
  return 0;
}

extern "C" int tt_profile_filter(const char* kernels) {
  tt::g_only = (kernels && *kernels) ? "," + std::string(kernels) + "," : std::string();
  return 0;
}

extern "C" int tt_profile_read(const char* kernel, double* total_ms, int64_t* launches) {
  if (!kernel || !total_ms || !launches) return tt::fail_arg("tt_profile_read: null pointer");
  *total_ms = 0.0;
  *launches = 0;
  for (auto& s : tt::g_slots) {
    if (s.name != kernel) continue;
    for (auto& p : s.ev) {
      hipError_t e = hipEventSynchronize(p.second);
      float ms = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, p.first, p.second);
      if (e != hipSuccess) { tt::set_error("tt_profile_read: %s", hipGetErrorString(e)); return (int)e; }
      *total_ms += ms;
      *launches += 1;
    }
  }
  return 0;
}
