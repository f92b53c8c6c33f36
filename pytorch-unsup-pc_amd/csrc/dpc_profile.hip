// Opt-in per-kernel event timing (see dpc_profile.h).  The only mutable global state in the library; guarded by
// an explicit enable call, never touched otherwise.
#include "dpc_profile.h"

#include <mutex>
#include <vector>

#include "../../include/dpc_render.h"

namespace {
struct Slot {
  const char* name;
  hipEvent_t a, b;
};
std::mutex g_mu;  // launches may come from torch's autograd thread while the caller reads results
bool g_on = false;
std::vector<Slot> g_slots;
size_t g_used = 0;
bool g_open = false;
}  // namespace

void dpc_prof_before(const char* name, hipStream_t st) {
  if (!g_on) return;  // the common case takes no lock
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_on || g_used >= g_slots.size()) return;
  g_slots[g_used].name = name;
  (void)hipEventRecord(g_slots[g_used].a, st);
  g_open = true;
}

void dpc_prof_after(hipStream_t st) {
  if (!g_on) return;
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_on || !g_open) return;
  (void)hipEventRecord(g_slots[g_used].b, st);
  ++g_used;
  g_open = false;
}

extern "C" {

int dpc_profile_enable(int capacity) {
  if (capacity < 0) return DPC_ERR_SHAPE;
  std::lock_guard<std::mutex> lock(g_mu);
  while ((int)g_slots.size() < capacity) {
    Slot s{nullptr, nullptr, nullptr};
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return DPC_ERR_LAUNCH;
    g_slots.push_back(s);
  }
  g_used = 0;
  g_open = false;
  g_on = true;
  return DPC_OK;
}

int dpc_profile_disable(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_on = false;
  return DPC_OK;
}

int dpc_profile_count(void) { return (int)g_used; }

// Milliseconds an EMPTY begin/end event pair reads on `stream`: the floor every bracketed launch carries.
int dpc_profile_pair_overhead(void* stream, int pairs, float* ms) {
  if (!ms || pairs < 1) return DPC_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return DPC_ERR_LAUNCH;
  float tot = 0.f;
  int ok = 0;
  for (int i = 0; i < pairs; ++i) {
    (void)hipEventRecord(a, st);
    (void)hipEventRecord(b, st);
    if (hipEventSynchronize(b) != hipSuccess) break;
    float t = 0.f;
    if (hipEventElapsedTime(&t, a, b) == hipSuccess) { tot += t; ++ok; }
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  if (!ok) return DPC_ERR_LAUNCH;
  *ms = tot / (float)ok;
  return DPC_OK;
}

int dpc_profile_get(int i, const char** name, float* ms) {
  if (i < 0 || (size_t)i >= g_used || !name || !ms) return DPC_ERR_SHAPE;
  *name = g_slots[i].name;
  return hipEventElapsedTime(ms, g_slots[i].a, g_slots[i].b) == hipSuccess ? DPC_OK : DPC_ERR_LAUNCH;
}

}  // extern "C"
