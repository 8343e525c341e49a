// Version / error reporting for libm355seg.
#include "common.hpp"
#include <map>
#include <mutex>
#include <utility>

namespace m355 {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  return atoi(v);
}
static Tuning read_tuning() {
  Tuning t;
  t.conv_ntw = env_int("M355_CONV_NTW", 0);
  t.conv_ksplit = env_int("M355_CONV_KSPLIT", 0);
  t.conv_slots = env_int("M355_CONV_SLOTS", 0);
  t.conv_persistent = env_int("M355_CONV_PERSISTENT", 1);
  t.no_small = env_int("M355_NO_SMALL", 0);
  t.smallcout_valu = env_int("M355_SMALLCOUT_VALU", 1);
  t.bww_nsplit = env_int("M355_BWW_NSPLIT", 0);
  t.bww_gen = env_int("M355_BWW_GEN", 2);
  t.bww_queue = env_int("M355_BWW_QUEUE", 1);
  t.h16_persistent = env_int("M355_H16_PERSISTENT", 1);
  t.tile16 = env_int("M355_TILE16", 1);
  t.convt_h16 = env_int("M355_CONVT_H16", 1);
  t.h16_w8 = env_int("M355_H16_W8", 1);
  t.h16_oneshot = env_int("M355_H16_ONESHOT", 1);
  t.h16_xcd = env_int("M355_H16_XCD", 1);
  t.conv_cube = env_int("M355_CONV_CUBE", 3);
  t.h16_order = env_int("M355_H16_ORDER", 3);
  t.fuse_softmax = env_int("M355_FUSE_SOFTMAX", 1);
  t.convt_wgs = env_int("M355_CONVT_WGS", 0);
  t.h16_stagger = env_int("M355_H16_STAGGER", 2);
  t.h16r = env_int("M355_H16R", 0);
  return t;
}
static Tuning g_tuning = read_tuning();
const Tuning& tuning() { return g_tuning; }

int num_cus() {
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    (void)hipGetLastError();  // no device (build host): not an error for a host-side query
    return 256;
  }
  if (cache[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      n = 256;
    }
    cache[dev] = n;
  }
  return cache[dev];
}
// ---- work-queue state of the queue-driven conv kernels ----
// One 64-byte slot (eight per-XCD ticket counters + an exit counter, all zero between launches: the last workgroup of
// a launch resets them) per (device, stream), in a small pool the library owns.  Launches on one stream are ordered,
// so they can share a slot; launches on DIFFERENT streams -- validation overlapping training on the same model, two
// replicas of a predictor -- get different slots and may run concurrently.  (Round 2 kept this state in the tail of
// the packed-weight buffer: two concurrent launches over one model corrupted each other's queues.)
// The pool is allocated on first use (hipMalloc + hipMemset, outside any stream capture: torch's capture warm-up runs
// every kernel eagerly first); kernels captured into a hipGraph keep the slot of their capture stream.
int* queue_state(hipStream_t st) {
  constexpr int SLOTS = 4096;
  static std::mutex mu;
  static int* pool[64] = {nullptr};
  static std::map<std::pair<int, hipStream_t>, int> slot_of;
  static int next_slot[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    (void)hipGetLastError();
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(mu);
  if (!pool[dev]) {
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)SLOTS * 64) != hipSuccess || hipMemset(p, 0, (size_t)SLOTS * 64) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    pool[dev] = (int*)p;
  }
  auto key = std::make_pair(dev, st);
  auto it = slot_of.find(key);
  if (it == slot_of.end()) it = slot_of.emplace(key, next_slot[dev]++ % SLOTS).first;   // (> 4096 live streams share slots)
  return pool[dev] + (size_t)it->second * 16;
}
}  // namespace m355

extern "C" void m355_reload_tuning(void) { m355::g_tuning = m355::read_tuning(); }
extern "C" int m355_version(void) { return M355_ABI_VERSION; }
extern "C" const char* m355_last_error(void) { return m355::g_err; }
