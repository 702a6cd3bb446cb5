// The switch table's one reader (sf_switches.h): every environment variable of the library is looked up here and nowhere else.
#include "sf_switches.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {
const char* const kNames[SW_COUNT] = {
#define SF_SW_NAME(id, name, what) name,
  SF_SWITCH_TABLE(SF_SW_NAME)
#undef SF_SW_NAME
};
char* g_val[SW_COUNT];
std::atomic<bool> g_loaded{false};
std::mutex g_mu;
void load_locked() {
  for (int i = 0; i < SW_COUNT; ++i) {
    const char* v = getenv(kNames[i]);
    // values are leaked on reload by design: a caller may still hold the previous pointer (a few bytes per reload, tests only)
    g_val[i] = v ? strdup(v) : nullptr;
  }
  g_loaded.store(true, std::memory_order_release);
}
}  // namespace

const char* sf_sw(SfSw k) {
  if (!g_loaded.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_loaded.load(std::memory_order_relaxed)) load_locked();
  }
  return g_val[k];
}

extern "C" void sf_reload_switches(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  load_locked();
}

// name / description of switch i (i >= count: nullptr): `python -m streamformer_amd.switches` prints the table
extern "C" const char* sf_switch_info(int i, int what) {
  static const char* const kWhat[SW_COUNT] = {
#define SF_SW_WHAT(id, name, what) what,
    SF_SWITCH_TABLE(SF_SW_WHAT)
#undef SF_SW_WHAT
  };
  if (i < 0 || i >= SW_COUNT) return nullptr;
  return what ? kWhat[i] : kNames[i];
}
