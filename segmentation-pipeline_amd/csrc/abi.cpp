// Version / error reporting for libm355seg.
#include "common.hpp"
#include <map>
#include <mutex>
#include <tuple>
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
  t.f32x3 = env_int("M355_F32X3", 1);
  t.f32x3_bww = env_int("M355_F32X3_BWW", 1);
  t.f32x3_edge = env_int("M355_F32X3_EDGE", 0);
  t.f32x3_convt = env_int("M355_F32X3_CONVT", 1);
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
// a launch resets them) per (device, stream, capture), in a 256 KB pool per device.  Launches on one stream are ordered,
// so they can share a slot; launches on DIFFERENT streams -- validation overlapping training on the same model, two
// replicas of a predictor -- get different slots and may run concurrently.  (Round 2 kept this state in the tail of
// the packed-weight buffer: two concurrent launches over one model corrupted each other's queues.)
// Launches recorded into a hipGraph bake the slot pointer into the graph, and torch captures every graph on the same
// side stream: keyed by stream alone, a captured validation forward and a captured train step replayed concurrently on
// two streams would share one slot (round-3 review).  A capturing stream therefore keys its slot by the CAPTURE id as
// well (hipStreamGetCaptureInfo): every captured launch sequence owns a slot no other graph and no eager stream uses.
// Slots are handed out round-robin and never freed (64 bytes each); beyond 4096 live keys they are shared again.
// The pool memory: the caller's (m355_queue_pool_set: a zero-filled device buffer of m355_queue_pool_bytes(), handed
// over once per device before the first queue-driven launch -- the Python host does this from torch's allocator, and
// the library then makes NO device allocation at all), else allocated here on first use (hipMalloc + hipMemset,
// outside any stream capture: a capture's warm-up runs every kernel eagerly first).
namespace {
constexpr int QUEUE_SLOTS = 4096;
std::mutex g_queue_mu;
int* g_queue_pool[64] = {nullptr};
bool g_queue_pool_owned[64] = {false};
}  // namespace

static int* g_overflow_flag[64] = {nullptr};
int* overflow_flag() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    (void)hipGetLastError();
    return nullptr;
  }
  return g_overflow_flag[dev];
}

int queue_pool_set(void* buf, size_t bytes, int dev) {
  if (dev < 0 || dev >= 64 || !buf || bytes < (size_t)QUEUE_SLOTS * 64 || ((uintptr_t)buf & 63)) return M355_EINVALID_ARG;
  std::lock_guard<std::mutex> lock(g_queue_mu);
  if (g_queue_pool[dev] && g_queue_pool[dev] != (int*)buf) return M355_EUNSUPPORTED;   // already in use by launches
  g_queue_pool[dev] = (int*)buf;
  return M355_OK;
}

int* queue_state(hipStream_t st) {
  static std::map<std::tuple<int, hipStream_t, unsigned long long>, int> slot_of;
  static int next_slot[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    (void)hipGetLastError();
    return nullptr;
  }
  unsigned long long capture = 0;
  {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(st, &cs, &id) == hipSuccess) {
      if (cs == hipStreamCaptureStatusActive) capture = id + 1;   // (+1: id 0 stays "not capturing")
    } else {
      (void)hipGetLastError();
    }
  }
  std::lock_guard<std::mutex> lock(g_queue_mu);
  if (!g_queue_pool[dev]) {
    if (capture) return nullptr;   // never allocate inside a capture (does not happen: warm-up runs eagerly first)
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)QUEUE_SLOTS * 64) != hipSuccess || hipMemset(p, 0, (size_t)QUEUE_SLOTS * 64) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    g_queue_pool[dev] = (int*)p;
    g_queue_pool_owned[dev] = true;
  }
  auto key = std::make_tuple(dev, st, capture);
  auto it = slot_of.find(key);
  if (it == slot_of.end()) it = slot_of.emplace(key, next_slot[dev]++ % QUEUE_SLOTS).first;
  return g_queue_pool[dev] + (size_t)it->second * 16;
}
}  // namespace m355

extern "C" void m355_reload_tuning(void) { m355::g_tuning = m355::read_tuning(); }
extern "C" size_t m355_queue_pool_bytes(void) { return (size_t)4096 * 64; }
extern "C" int m355_queue_pool_set(void* zeroed_device_buffer, size_t bytes, int32_t device) {
  const int rc = m355::queue_pool_set(zeroed_device_buffer, bytes, device);
  if (rc == M355_EINVALID_ARG)
    m355::set_error("queue_pool_set: need a 64-byte aligned, zero-filled device buffer of >= %zu bytes and a device index in [0, 64)",
                    (size_t)4096 * 64);
  else if (rc != M355_OK)
    m355::set_error("queue_pool_set: device %d already has a work-queue pool in use", (int)device);
  return rc;
}
extern "C" int m355_overflow_flag_set(void* device_word, int32_t device) {
  if (device < 0 || device >= 64 || ((uintptr_t)device_word & 3)) {
    m355::set_error("overflow_flag_set: need a 4-byte aligned device word (or NULL) and a device index in [0, 64)");
    return M355_EINVALID_ARG;
  }
  m355::g_overflow_flag[device] = (int*)device_word;
  return M355_OK;
}
extern "C" int m355_version(void) { return M355_ABI_VERSION; }
extern "C" const char* m355_last_error(void) { return m355::g_err; }
