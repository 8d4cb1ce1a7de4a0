// libmem.so for MI355X — allocator / stream / copy library behind include/ares_memory.h.
//
// Replaces the reference's three libmem flavours (cgoutils/memory/malloc.c, cuda_malloc.cu,
// rmm_alloc.cu) with one HIP implementation designed for a 288 GB HBM3E device:
//   * host memory is pinned + portable (hipHostMalloc) and zero-filled, so AsyncCopyHostToDevice
//     is a true asynchronous DMA (reference cuda_malloc.cu:44-52);
//   * device memory comes from a per-device cache of hipMalloc'ed blocks in size bins; a freed block
//     is fenced with events on every stream the host uses and only reused once the fence has
//     passed, so the per-batch, per-column alloc/free pattern of the Go host
//     (query/aql_processor.go:726-804, :1345-1431) never reaches the driver after warm-up and never
//     synchronises the device; the block is zeroed before DeviceAllocate returns (reference
//     cuda_malloc.cu:97-104 contract);
//   * ARES_MEM_POOL=0 switches to plain hipMalloc/hipFree (debugging aid, same semantics).
// Every Go-facing entry point selects the device itself; the lower-case ones assume the caller
// (libalgorithm.so) already did (reference cgoutils/memory.h:88-98).
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "ares_memory.h"
#include "ares_extensions.h"

#pragma clang diagnostic ignored "-Wdeprecated-declarations"  // hipProfilerStart/Stop: the ABI asks for them

namespace {

constexpr int kMaxErrorLen = 160;
constexpr int kMaxDevices = 64;

char *format_error(const char *what, hipError_t err) {
  char *buf = static_cast<char *>(malloc(kMaxErrorLen));
  snprintf(buf, kMaxErrorLen, "ERROR when calling HIP functions: %s: %s\n", what,
           hipGetErrorString(err));
  return buf;
}

char *format_message(const char *fn, const char *msg) {
  char *buf = static_cast<char *>(malloc(kMaxErrorLen));
  snprintf(buf, kMaxErrorLen, "ERROR when making C function %s: %s\n", fn, msg);
  return buf;
}

inline CGoCallResHandle ok(void *res = nullptr) { return CGoCallResHandle{res, nullptr}; }
inline CGoCallResHandle fail(const char *what, hipError_t err) {
  (void)hipGetLastError();  // clear the sticky error like cudaGetLastError does
  return CGoCallResHandle{nullptr, format_error(what, err)};
}

// ARES_RTC_TRACE=<file> (shared with libalgorithm.so): entry points that kept the calling thread longer than 5 ms
struct MemSlow {
  const char *what;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit MemSlow(const char *w) : what(w) {}
  ~MemSlow() {
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms <= 5.0) return;
    static const char *path = getenv("ARES_RTC_TRACE");
    if (!path || !path[0]) return;
    if (FILE *o = fopen(path, "a")) {
      const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
      fprintf(o, "%.3f slow libmem:%s %.3f ms\n", now, what, ms);
      fclose(o);
    }
  }
};

// one line per large DeviceAllocate in the same file: where the block came from (cold-start diagnostics)
inline bool mem_trace_enabled() {
  static const char *path = getenv("ARES_RTC_TRACE");
  return path && path[0];
}
inline void mem_trace_alloc(size_t bytes, size_t rounded, const char *source, size_t parkedInBin, size_t parkedBytes) {
  static const char *path = getenv("ARES_RTC_TRACE");
  if (!path || !path[0] || rounded < (size_t(64) << 20)) return;
  if (FILE *o = fopen(path, "a")) {
    const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    fprintf(o, "%.3f libmem:alloc %.1f MB (bin %.1f MB) from %s; %zu other blocks parked in the bin, %.1f MB parked in all\n", now,
            bytes / 1e6, rounded / 1e6, source, parkedInBin, parkedBytes / 1e6);
    fclose(o);
  }
}

#define MEM_TRY(expr, what)                        \
  do {                                             \
    hipError_t e_ = (expr);                        \
    if (e_ != hipSuccess) return fail(what, e_);   \
  } while (0)

// libalgorithm.so may hold transforms it has accepted but not launched yet (cross-call fusion,
// aresdb_amd/csrc/algo/transform.hip); it registers a hook here, and every entry point through
// which the host could observe, free or overwrite device memory runs it first.
typedef void (*FlushHook)(int device);
std::atomic<FlushHook> g_flushHook{nullptr};
// finer-grained notifications (include/ares_extensions.h: AresDeferralHooks); null = use the flush hook
std::atomic<const AresDeferralHooks *> g_hooks{nullptr};
inline void flush_pending(int device) {
  if (FlushHook h = g_flushHook.load(std::memory_order_acquire)) h(device);
}
inline void flush_pending_current() {
  int device = 0;
  if (hipGetDevice(&device) == hipSuccess) flush_pending(device);
}
// the host waits for a stream: pending work MAY stay pending when nothing but later library calls
// can observe its results
inline void notify_wait(int device, void *stream) {
  const AresDeferralHooks *h = g_hooks.load(std::memory_order_acquire);
  if (h && h->on_wait) h->on_wait(device, stream); else flush_pending(device);
}
// a copy is about to read or write [ptr, ptr + bytes) (and about to be submitted: fences recorded before do not cover it)
extern std::atomic<uint64_t> g_activity;
inline void notify_access(int device, const void *ptr, size_t bytes) {
  g_activity.fetch_add(1, std::memory_order_acq_rel);
  const AresDeferralHooks *h = g_hooks.load(std::memory_order_acquire);
  if (h && h->on_access) h->on_access(device, ptr, bytes); else flush_pending(device);
}
inline void notify_access_current(const void *ptr, size_t bytes) {
  int device = 0;
  if (hipGetDevice(&device) == hipSuccess) notify_access(device, ptr, bytes);
}
// the host frees a block: true = work that has not been launched yet still reads it, keep it aside
inline uintptr_t notify_free(int device, void *ptr, size_t bytes) {
  const AresDeferralHooks *h = g_hooks.load(std::memory_order_acquire);
  if (h && h->on_free) return h->on_free(device, ptr, bytes);
  flush_pending(device);
  return 0;
}

std::atomic<const AresMemAuxHooks *> g_aux{nullptr};
void note_write(int device, const void *ptr, size_t bytes);  // below: the written ranges of live blocks
inline void notify_write(int device, const void *ptr, size_t bytes) {
  note_write(device, ptr, bytes);
  const AresMemAuxHooks *h = g_aux.load(std::memory_order_acquire);
  if (h && h->size >= offsetof(AresMemAuxHooks, on_write) + sizeof(h->on_write) && h->on_write) h->on_write(device, ptr, bytes);
}
inline void notify_write_current(const void *ptr, size_t bytes) {
  int device = 0;
  if (hipGetDevice(&device) == hipSuccess) notify_write(device, ptr, bytes);
}
inline void notify_stream_destroy(int device, void *stream) {
  const AresMemAuxHooks *h = g_aux.load(std::memory_order_acquire);
  if (h && h->size >= offsetof(AresMemAuxHooks, on_stream_destroy) + sizeof(h->on_stream_destroy) && h->on_stream_destroy)
    h->on_stream_destroy(device, stream);
}
inline void notify_trim(int device) {
  const AresMemAuxHooks *h = g_aux.load(std::memory_order_acquire);
  if (h && h->size >= offsetof(AresMemAuxHooks, trim) + sizeof(h->trim) && h->trim) h->trim(device);
}

bool use_pool() {
  static const bool v = [] {
    const char *e = getenv("ARES_MEM_POOL");
    return !(e && e[0] == '0');
  }();
  return v;
}

// ---- device memory cache -----------------------------------------------------------------------------
// The Go host allocates and frees per column, per scratch frame, per batch
// (query/aql_processor.go:726-804, :1345-1431) and — because cudaFree synchronises the device — may
// free a buffer right after enqueueing the kernel that reads it (shrinkStackFrame,
// query/time_series_aggregate.go:756-769).  A drop-in allocator therefore has to (a) make frees
// cheap and (b) never hand a block out again while work enqueued before its free may still touch it.
//
// Design: blocks come from hipMalloc and are cached per device in size bins (1/8-octave steps).
// DeviceFree records one event on every stream the host uses on that device (the streams made by
// CreateCudaStream plus the null stream) and parks the block with that fence; DeviceAllocate reuses
// a parked block of the bin only once its whole fence has completed.  The zero fill DeviceAllocate
// promises happens when the block is PARKED: the private stream waits for the fence, clears the block
// and records one more event that joins the fence — so an allocation from the cache costs no fill, no
// launch and no host wait (only a block that comes fresh from hipMalloc is cleared synchronously).
// No driver allocation after warm-up, no device-wide
// synchronisation ever, and no reliance on cross-stream reuse inside HIP's own stream-ordered pool
// (which libalgorithm.so uses for its stream-local temporaries only).
// One event of a block's fence and the stream it was recorded on.  An event must not outlive its stream: the HIP runtime
// of ROCm 7.x reaches into the stream object when such an event is queried and its work is not yet marked complete in
// the runtime's books — with the stream destroyed that is a use after free (observed: std::bad_variant_access thrown out
// of hipEventQuery, segmentation faults, hangs, single words of unrelated heap memory changed by one; profiles/
// r4_race_hunt.md).  DestroyCudaStream therefore retires every fence event of the stream while the stream still exists.
struct FenceEvent {
  hipEvent_t event;
  hipStream_t stream;
};
struct ParkedBlock {
  void *ptr;
  std::vector<FenceEvent> fence;
  bool zeroed = false;  // the block was cleared behind its fence (the last event of the fence covers the fill)
};

constexpr size_t kWaitForParkedFrom = size_t(16) << 20;  // bytes; see pool_alloc

// Write tracking.  DeviceAllocate hands out cleared memory, and the clearing happens when a block is parked;
// a block (or the part of one) that nothing wrote to since it was last cleared is still clear.  Every writer
// reports what it writes — libmem's own copies and fills here, libalgorithm.so its kernels' outputs through
// AresMemNoteWrite (it switches tracking on with AresMemEnableWriteTracking once it does; without that call a
// freed block counts as written all over) — so parking a block clears only the ranges written: the Go host's
// per-batch index vector that a fused batch never materialises, and the unused tails of a query's result
// vectors (sized for "every row a new group"), are most of the 9 GB per 1 B-row step that were cleared before.
constexpr size_t kMaxDirtyRanges = 24;
struct LiveBlock {
  size_t rounded = 0;
  bool allDirty = false;
  std::vector<std::pair<size_t, size_t>> dirty;  // [lo, hi) byte ranges, disjoint, sorted
};
std::atomic<bool> g_writeTracking{false};
// Fence sharing (below: DeviceState::lastFence) is only sound when everything that submits work to the device's streams
// says so: libalgorithm.so switches it on (AresMemEnableActivityTracking) once it reports its entry points and launches.
std::atomic<bool> g_activityTracking{false};
// Counts everything that may have submitted work to any stream (process-wide: a finer count would only share more).
std::atomic<uint64_t> g_activity{1};
inline void note_activity() { g_activity.fetch_add(1, std::memory_order_acq_rel); }

struct DeviceState {
  std::once_flag once;
  hipStream_t allocStream = nullptr;  // fills of parked blocks, each behind the block's fence
  hipStream_t freshStream = nullptr;  // the synchronous fill of a block that comes straight from hipMalloc: never queued
                                      // behind other queries' fences (head-of-line blocking on allocStream)
  hipError_t initError = hipSuccess;
  size_t cacheCapBytes = 0;           // parked bytes above which the cache is trimmed (a quarter of the device, read once)
  size_t driverAllocs = 0, driverFrees = 0, trims = 0;  // driver calls the cache could not avoid (AresMemDriverCalls)
  std::mutex mu;
  std::vector<hipStream_t> streams;                      // streams created through CreateCudaStream
  std::map<size_t, std::vector<ParkedBlock>> bins;       // rounded size -> parked blocks
  std::map<uintptr_t, LiveBlock> live;                   // allocation -> rounded size + what was written to it
  std::vector<FenceEvent> freeEvents;  // recycled (with the stream of their last record: they go when that stream goes)
  size_t parkedBytes = 0;
  // freed by the host while deferred work of one stream (the tag) still reads them (AresMemReleaseHeld)
  std::vector<std::pair<void *, uintptr_t>> held;
  // allocations that took a parked block out of `bins` and are waiting for its fence outside `mu` (pool_alloc): the
  // events they wait on are in nobody's books meanwhile, so DestroyCudaStream lets them finish before it retires a
  // stream's events
  // allocations waiting (outside the lock) for a parked block's fence, per stream their fence events belong to: only those
  // hold events of a stream outside every list — DestroyCudaStream lets exactly them through, not every waiter of the device
  std::map<hipStream_t, int> waiters;
  std::condition_variable waitersCv;
  // ---- shared fences.  A fence is one event per stream, recorded at a free.  The Go host frees in bursts (a batch's columns,
  // index and predicate vector: query/aql_processor.go:695-714) with nothing submitted in between: the events of the first
  // free of a burst fence the others just as well — recording them again cost 1.6 us per event and free, and the first
  // hipEventQuery of every new event 3.5 us at the allocation that meets it (tools/ubench_libmem.cpp, ubench_events.py).
  // g_activity counts everything that may have submitted work since: entry points and kernel launches of libalgorithm.so
  // (AresMemNoteActivity), this library's own copies and fills, stream creation; a free that finds it unchanged since
  // `lastFence` was recorded shares those events (reference counts in `events`; an event seen complete is not queried again).
  uint64_t lastFenceActivity = 0;
  std::vector<FenceEvent> lastFence;  // holds one reference on each of its events
  struct EventInfo {
    int refs = 0;
    bool done = false;
  };
  std::map<hipEvent_t, EventInfo> events;
};
DeviceState g_devices[kMaxDevices];

hipError_t device_state(int device, DeviceState **out) {
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  DeviceState &st = g_devices[device];
  std::call_once(st.once, [&] {
    // ARES_MEM_FILL_PRIORITY=low puts the fills of parked blocks on a lowest-priority stream.  Measured:
    // the scan kernel gains 0-3 %, but the first process on a box now and then takes twice as long per
    // step with it (3 of 5 fresh boxes) — so the default is an ordinary stream.
    int priority = 0;
    const char *e = getenv("ARES_MEM_FILL_PRIORITY");
    if (e && strcmp(e, "low") == 0) {
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) priority = least;
      else (void)hipGetLastError();
    }
    st.initError = hipStreamCreateWithPriority(&st.allocStream, hipStreamNonBlocking, priority);
    if (st.initError == hipSuccess) st.initError = hipStreamCreateWithFlags(&st.freshStream, hipStreamNonBlocking);
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) st.cacheCapBytes = totalB / 4;
    else (void)hipGetLastError();
    if (const char *cap = getenv("ARES_MEM_CACHE_MB")) st.cacheCapBytes = static_cast<size_t>(atoll(cap)) << 20;
  });
  *out = &st;
  return st.initError;
}

hipError_t current_device_state(DeviceState **out) {
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  return device_state(device, out);
}

size_t bin_size(size_t bytes) {
  if (bytes < 256) return 256;
  const int top = 63 - __builtin_clzll(static_cast<unsigned long long>(bytes));
  const size_t step = static_cast<size_t>(1) << (top > 3 ? top - 3 : 0);
  return (bytes + step - 1) / step * step;
}

// caller holds st->mu
bool fence_done(DeviceState *st, const ParkedBlock &b) {
  for (const FenceEvent &f : b.fence) {
    DeviceState::EventInfo &info = st->events[f.event];
    if (info.done) continue;  // (seen complete through another block that shares it)
    if (hipEventQuery(f.event) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    info.done = true;
  }
  return true;
}

// caller holds st->mu: one reference on `f` goes; the last one recycles the event (while the stream of its last record
// exists: see FenceEvent) or destroys it
void release_event(DeviceState *st, const FenceEvent &f) {
  auto it = st->events.find(f.event);
  if (it != st->events.end() && --it->second.refs > 0) return;
  if (it != st->events.end()) st->events.erase(it);
  bool alive = f.stream == nullptr || f.stream == st->allocStream || f.stream == st->freshStream;
  for (size_t i = 0; !alive && i < st->streams.size(); i++) alive = st->streams[i] == f.stream;
  if (alive) st->freeEvents.push_back(f);
  else (void)hipEventDestroy(f.event);
}

// caller holds st->mu.  An event is only kept for reuse while the stream of its last record exists (see FenceEvent):
// one whose stream has gone meanwhile is destroyed here, never recorded again.
void recycle_events(DeviceState *st, ParkedBlock &b) {
  for (const FenceEvent &f : b.fence) release_event(st, f);
  b.fence.clear();
}
void drop_last_fence(DeviceState *st) {
  for (const FenceEvent &f : st->lastFence) release_event(st, f);
  st->lastFence.clear();
  st->lastFenceActivity = 0;
}

// Releases parked blocks (oldest bins first) until `need` more bytes fit under the cap or nothing
// is left; used when hipMalloc fails or the cache grows past half of the device.
// ARES_MEM_ABORT=1 (race hunting): a host-side misuse that the library can survive (a block freed twice) kills the
// process on the spot instead of coming back as an error through CGoCallResHandle
bool mem_abort_on_invariant() {
  static const bool on = [] {
    const char *e = getenv("ARES_MEM_ABORT");
    return e && e[0] == '1';
  }();
  return on;
}

bool mem_debug() {
  static const bool on = [] {
    const char *e = getenv("ARES_MEM_DEBUG");
    return e && e[0] == '1';
  }();
  return on;
}

void trim(DeviceState *st, size_t keepBytes) {
  st->trims++;
  if (mem_debug()) fprintf(stderr, "libmem: trimming the block cache from %.1f MB to %.1f MB\n", st->parkedBytes / 1e6, keepBytes / 1e6);
  for (auto it = st->bins.begin(); it != st->bins.end() && st->parkedBytes > keepBytes;) {
    auto &vec = it->second;
    while (!vec.empty() && st->parkedBytes > keepBytes) {
      ParkedBlock b = vec.back();
      vec.pop_back();
      for (const FenceEvent &f : b.fence) (void)hipEventSynchronize(f.event);
      recycle_events(st, b);
      (void)hipFree(b.ptr);
      st->driverFrees++;
      st->parkedBytes -= it->first;
    }
    it = vec.empty() ? st->bins.erase(it) : std::next(it);
  }
}

// ARES_MEM_DEBUG=1: driver allocations the cache could not avoid, reported at exit (diagnostics)
void count_driver_allocation(size_t bytes) {
  if (!mem_debug()) return;
  static std::atomic<long> calls{0}, total{0};
  static const int registered = atexit([] {
    fprintf(stderr, "libmem: %ld hipMalloc calls, %.1f MB in total\n", calls.load(), static_cast<double>(total.load()) / 1e6);
  });
  (void)registered;
  calls++;
  total += static_cast<long>(bytes);
}

// ARES_MEM_VERIFY_CLEAN=1 (tests): every block taken from the cache as "cleared" is checked on the device —
// a writer that did not report what it wrote shows up as an abort here instead of as stale data in a
// DeviceAllocate result.
__global__ void verify_clear_kernel(const uint32_t *p, size_t words, unsigned long long *firstBad) {
  unsigned long long first = ~0ull;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < words; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    if (p[i] != 0u && i < first) first = i;
  if (first != ~0ull) atomicMin(firstBad, first);
}
void verify_clear(DeviceState *st, void *ptr, size_t rounded, size_t requested) {
  static const bool on = [] {
    const char *e = getenv("ARES_MEM_VERIFY_CLEAN");
    return e && e[0] == '1';
  }();
  if (!on) return;
  static unsigned long long *flag = nullptr;
  if (!flag && hipHostMalloc(reinterpret_cast<void **>(&flag), sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) return;
  *flag = ~0ull;
  hipLaunchKernelGGL(verify_clear_kernel, dim3(1024), dim3(256), 0, st->freshStream, static_cast<const uint32_t *>(ptr), rounded / 4, flag);
  if (hipStreamSynchronize(st->freshStream) != hipSuccess || *flag != ~0ull) {
    uint32_t word = 0;
    (void)hipMemcpy(&word, static_cast<const uint32_t *>(ptr) + *flag, 4, hipMemcpyDeviceToHost);
    fprintf(stderr, "libmem: a block handed out as cleared holds data (a writer did not report its writes): bin %zu bytes, "
                    "request %zu bytes, first non-zero word at byte offset %llu = 0x%08x\n", rounded, requested, *flag * 4ull, word);
    abort();
  }
}

hipError_t pool_alloc(DeviceState *st, void **p, size_t bytes, bool zero) {
  if (bytes == 0) bytes = 1;
  if (!use_pool()) {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return e;
    return zero ? hipMemset(*p, 0, bytes) : hipSuccess;
  }
  size_t rounded = bin_size(bytes);  // (becomes the size of the block that is handed out: a parked block of a larger bin may serve)
  void *ptr = nullptr;
  bool cleared = false;
  ParkedBlock waitFor;
  waitFor.ptr = nullptr;
  {
    std::lock_guard<std::mutex> lock(st->mu);
    auto it = st->bins.find(rounded);
    if ((it == st->bins.end() || it->second.empty()) && rounded >= kWaitForParkedFrom) {
      // Nothing parked in the request's own bin: a parked block of a bin up to 1.5 x larger serves as well — it is idle
      // memory either way, and the driver charges 28 ms per GB of fresh memory at best.  (A query's result vectors are
      // sized by the batch's survivor count: the same query with another constant asks for another bin.)
      for (auto jt = st->bins.upper_bound(rounded); jt != st->bins.end() && jt->first <= rounded + rounded / 2; ++jt)
        if (!jt->second.empty()) {
          it = jt;
          rounded = jt->first;
          break;
        }
    }
    if (it != st->bins.end()) {
      auto &vec = it->second;
      for (size_t i = 0; i < vec.size(); i++) {
        if (fence_done(st, vec[i])) {
          ptr = vec[i].ptr;
          cleared = vec[i].zeroed;
          recycle_events(st, vec[i]);
          vec[i] = vec.back();
          vec.pop_back();
          st->parkedBytes -= rounded;
          break;
        }
      }
      // Nothing ready, but blocks of this size are parked: for a large block, waiting for the oldest one's
      // fence and fill (bounded by the work enqueued before its free, typically well under a millisecond)
      // beats asking the driver for another multi-hundred-megabyte allocation — which costs milliseconds
      // (tens of them in a fresh process) and grows the footprint.
      if (!ptr && !vec.empty() && rounded >= kWaitForParkedFrom) {
        ParkedBlock b = vec.front();
        vec.erase(vec.begin());
        st->parkedBytes -= rounded;
        waitFor = b;
        for (const FenceEvent &f : waitFor.fence) st->waiters[f.stream]++;
      }
    }
    if (mem_trace_enabled()) {  // (diagnostics only: nothing is counted when the trace is off)
      const char *source = ptr ? "the cache" : waitFor.ptr ? "the cache after a wait" : "the driver";
      auto bt = st->bins.find(rounded);
      mem_trace_alloc(bytes, rounded, source, bt == st->bins.end() ? 0 : bt->second.size(), st->parkedBytes);
    }
  }
  if (waitFor.ptr) {  // (outside the lock: frees and allocations of other threads go on)
    for (const FenceEvent &f : waitFor.fence) (void)hipEventSynchronize(f.event);
    ptr = waitFor.ptr;
    cleared = waitFor.zeroed;
    {
      std::lock_guard<std::mutex> lock(st->mu);
      for (const FenceEvent &f : waitFor.fence) {
        auto w = st->waiters.find(f.stream);
        if (w != st->waiters.end() && --w->second <= 0) st->waiters.erase(w);
      }
      recycle_events(st, waitFor);
    }
    st->waitersCv.notify_all();
  }
  if (!ptr) {
    count_driver_allocation(rounded);
    {
      std::lock_guard<std::mutex> lock(st->mu);
      st->driverAllocs++;
    }
    // ARES_MEM_POOL_ALLOC=1 (experiment for the next round): fresh blocks from the stream-ordered pool (see
    // libalgorithm's ARES_TEMP_POOL_ALLOC); nothing is ever freed into that pool
    static const bool poolAlloc = [] {
      const char *v = getenv("ARES_MEM_POOL_ALLOC");
      return v && v[0] == '1';
    }();
    hipError_t e = poolAlloc ? hipMallocAsync(&ptr, rounded, st->freshStream) : hipMalloc(&ptr, rounded);
    if (e == hipSuccess && poolAlloc) e = hipStreamSynchronize(st->freshStream);
    if (e != hipSuccess) {  // out of memory: give both libraries' caches back and retry once
      (void)hipGetLastError();
      {
        std::lock_guard<std::mutex> lock(st->mu);
        trim(st, 0);
      }
      int device = 0;
      if (hipGetDevice(&device) == hipSuccess) notify_trim(device);
      e = hipMalloc(&ptr, rounded);
      if (e != hipSuccess) return e;
    }
  }
  {
    std::lock_guard<std::mutex> lock(st->mu);
    if (st->live.count(reinterpret_cast<uintptr_t>(ptr))) {  // invariant: a block has one owner
      fprintf(stderr, "libmem: block %p (%zu bytes) was about to be handed out while it is still live\n", ptr, rounded);
      abort();
    }
    LiveBlock &b = st->live[reinterpret_cast<uintptr_t>(ptr)];
    b.rounded = rounded;
    b.dirty.clear();
    b.allDirty = !zero;  // deviceMalloc: libalgorithm.so's own result buffers (HyperLogLog) — their writers do not report
  }
  *p = ptr;
  if (zero && cleared) verify_clear(st, ptr, rounded, bytes);
  if (zero && !cleared) {  // (the whole block: what lies behind `bytes` is handed out by a later, larger request)
    hipError_t e = hipMemsetAsync(ptr, 0, rounded, st->freshStream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st->freshStream);
  }
  return hipSuccess;
}

// caller holds st->mu.  Records [ptr, ptr + bytes) as written in the live block that contains ptr.
void note_write_locked(DeviceState *st, const void *ptr, size_t bytes) {
  if (bytes == 0 || st->live.empty()) return;
  const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
  auto it = st->live.upper_bound(a);
  if (it == st->live.begin()) return;
  --it;
  LiveBlock &b = it->second;
  if (a >= it->first + b.rounded || b.allDirty) return;
  size_t lo = a - it->first, hi = lo + bytes < b.rounded ? lo + bytes : b.rounded;
  std::vector<std::pair<size_t, size_t>> &d = b.dirty;
  size_t i = 0;
  while (i < d.size() && d[i].second < lo) i++;
  size_t j = i;
  while (j < d.size() && d[j].first <= hi) {
    lo = d[j].first < lo ? d[j].first : lo;
    hi = d[j].second > hi ? d[j].second : hi;
    j++;
  }
  d.erase(d.begin() + i, d.begin() + j);
  d.insert(d.begin() + i, {lo, hi});
  if (d.size() > kMaxDirtyRanges) {
    b.allDirty = true;
    d.clear();
  }
}

void note_write(int device, const void *ptr, size_t bytes) {
  if (device < 0 || device >= kMaxDevices || !use_pool()) return;
  DeviceState *st = &g_devices[device];
  std::lock_guard<std::mutex> lock(st->mu);
  note_write_locked(st, ptr, bytes);
}

hipError_t pool_free(DeviceState *st, void *p) {
  if (p == nullptr) return hipSuccess;
  if (!use_pool()) return hipFree(p);
  std::lock_guard<std::mutex> lock(st->mu);
  auto it = st->live.find(reinterpret_cast<uintptr_t>(p));
  if (it == st->live.end()) {  // not ours (allocated before the pool was switched on) — or freed twice
    for (auto &bin : st->bins)
      for (const ParkedBlock &pb : bin.second)
        if (pb.ptr == p) {  // the host's bug, reported the way the reference's DeviceFree reports a failing cudaFree
          fprintf(stderr, "libmem: block %p freed twice (it is parked in the cache)\n", p);
          if (mem_abort_on_invariant()) abort();
          return hipErrorInvalidDevicePointer;
        }
    return hipFree(p);
  }
  const size_t rounded = it->second.rounded;
  std::vector<std::pair<size_t, size_t>> written;
  if (it->second.allDirty || !g_writeTracking.load(std::memory_order_acquire)) written.emplace_back(0, rounded);
  else written.swap(it->second.dirty);
  st->live.erase(it);
  ParkedBlock b;
  b.ptr = p;
  auto fence_on = [&](hipStream_t s) -> hipError_t {
    hipEvent_t e;
    if (!st->freeEvents.empty()) {
      e = st->freeEvents.back().event;
      st->freeEvents.pop_back();
    } else {
      hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
      if (err != hipSuccess) return err;
    }
    hipError_t err = hipEventRecord(e, s);
    b.fence.push_back(FenceEvent{e, s});
    st->events[e] = DeviceState::EventInfo{1, false};
    return err;
  };
  hipError_t err = hipSuccess;
  const uint64_t activityNow = g_activity.load(std::memory_order_acquire);
  if (g_activityTracking.load(std::memory_order_acquire) && !st->lastFence.empty() && st->lastFenceActivity == activityNow) {
    // nothing was submitted since the last fence was recorded: it fences this block as well
    for (const FenceEvent &f : st->lastFence) {
      st->events[f.event].refs++;
      b.fence.push_back(f);
    }
  } else {
    err = fence_on(nullptr);
    for (hipStream_t s : st->streams)
      if (err == hipSuccess) err = fence_on(s);
    if (err == hipSuccess && g_activityTracking.load(std::memory_order_acquire)) {
      drop_last_fence(st);
      for (const FenceEvent &f : b.fence) {
        st->events[f.event].refs++;
        st->lastFence.push_back(f);
      }
      st->lastFenceActivity = activityNow;
    }
  }
  if (err != hipSuccess) {  // cannot fence: fall back to the synchronising free of the reference
    (void)hipGetLastError();
    recycle_events(st, b);
    return hipFree(p);
  }
  if (written.empty()) {
    b.zeroed = true;  // nothing wrote to it since it was cleared
  } else {  // clear what was written behind the block's fence, off every query's critical path
    bool okFill = true;
    for (const FenceEvent &f : b.fence) okFill = okFill && hipStreamWaitEvent(st->allocStream, f.event, 0) == hipSuccess;
    for (const auto &r : written)
      okFill = okFill && hipMemsetAsync(static_cast<uint8_t *>(p) + r.first, 0, r.second - r.first, st->allocStream) == hipSuccess;
    if (okFill && fence_on(st->allocStream) == hipSuccess) {
      b.zeroed = true;
    } else {  // a fill may be in flight without a fence behind it: the block must not be handed out again
      (void)hipGetLastError();
      recycle_events(st, b);
      st->driverFrees++;
      return hipFree(p);  // (synchronises the device: whatever was enqueued has run)
    }
  }
  static const bool syncFree = [] {  // diagnostics: a parked block is quiescent (fence and fill complete) before the free returns
    const char *e = getenv("ARES_MEM_SYNC_FREE");
    return e && e[0] == '1';
  }();
  if (syncFree)
    for (const FenceEvent &f : b.fence) (void)hipEventSynchronize(f.event);
  st->bins[rounded].push_back(b);
  st->parkedBytes += rounded;
  // the cache stays below a quarter of the device (the figure is read once per device, not per free)
  if (st->cacheCapBytes && st->parkedBytes > st->cacheCapBytes) trim(st, st->cacheCapBytes / 2);
  return hipSuccess;
}

}  // namespace

// ARES_BACKTRACE=1 (diagnostics): a C++ exception that nobody catches (std::terminate) and a segmentation fault print the
// native call stack of the faulting thread before the process dies — which entry point, which runtime call.
#include <execinfo.h>
#include <csignal>
#include <exception>
#include <unistd.h>
namespace {
void print_native_stack(const char *what) {
  void *frames[64];
  const int n = backtrace(frames, 64);
  (void)!write(2, what, strlen(what));
  backtrace_symbols_fd(frames, n, 2);
}
void on_terminate() {
  print_native_stack("libmem: std::terminate — native stack:\n");
  abort();
}
void on_fatal_signal(int sig) {
  print_native_stack(sig == SIGSEGV ? "libmem: SIGSEGV — native stack:\n" : "libmem: fatal signal — native stack:\n");
  signal(sig, SIG_DFL);
  raise(sig);
}
struct BacktraceInstaller {
  BacktraceInstaller() {
    const char *e = getenv("ARES_BACKTRACE");
    if (!(e && e[0] == '1')) return;
    std::set_terminate(&on_terminate);
    signal(SIGSEGV, &on_fatal_signal);
    signal(SIGBUS, &on_fatal_signal);
  }
} g_backtraceInstaller;
}  // namespace

extern "C" {

// extension (include/ares_extensions.h): called by the sibling libalgorithm.so, never by the host
void AresMemSetFlushHook(void (*hook)(int device)) { g_flushHook.store(hook, std::memory_order_release); }
void AresMemSetDeferralHooks(const AresDeferralHooks *hooks) {
  if (hooks && hooks->flush) g_flushHook.store(hooks->flush, std::memory_order_release);
  g_hooks.store(hooks, std::memory_order_release);
}

void AresMemSetAuxHooks(const AresMemAuxHooks *hooks) { g_aux.store(hooks, std::memory_order_release); }

// Write tracking (see LiveBlock): libalgorithm.so reports what its kernels write; anything else that writes
// device memory obtained from this library without going through its copy / fill entry points (a collective
// library receiving into it, say) reports it the same way.
void AresMemNoteWrite(int device, const void *ptr, size_t bytes) { note_write(device, ptr, bytes); }
void AresMemEnableWriteTracking(void) { g_writeTracking.store(true, std::memory_order_release); }
// Something may have been submitted to a stream of `device` (an entry point of libalgorithm.so was called, a kernel was
// launched from one of its hooks, RCCL was handed the query's stream): fences recorded before do not cover it.
void AresMemNoteActivity(void) { note_activity(); }
void AresMemEnableActivityTracking(void) { g_activityTracking.store(true, std::memory_order_release); }

// What the host currently owns on `device` (the Go host keeps the same books itself and asserts they
// return to zero after every query, query/aql_processor_test.go:230-231): bytes of live DeviceAllocate /
// deviceMalloc blocks (rounded to their size bins), blocks kept aside for deferred work, bytes parked
// in the cache.
void AresMemStats(int device, size_t *liveBytes, size_t *liveBlocks, size_t *heldBlocks, size_t *parkedBytes) {
  size_t lb = 0, ln = 0, hn = 0, pb = 0;
  if (device >= 0 && device < kMaxDevices) {
    DeviceState *st = &g_devices[device];
    std::lock_guard<std::mutex> lock(st->mu);
    for (auto &kv : st->live) lb += kv.second.rounded;
    ln = st->live.size();
    hn = st->held.size();
    for (auto &h : st->held) {  // freed by the host, kept aside: not the host's any more
      auto it = st->live.find(reinterpret_cast<uintptr_t>(h.first));
      if (it != st->live.end()) {
        lb -= it->second.rounded;
        ln--;
      }
    }
    pb = st->parkedBytes;
  }
  if (liveBytes) *liveBytes = lb;
  if (liveBlocks) *liveBlocks = ln;
  if (heldBlocks) *heldBlocks = hn;
  if (parkedBytes) *parkedBytes = pb;
}

// Driver calls the block cache could not avoid on `device` since the process started: hipMalloc calls,
// hipFree calls, cache trims (diagnostics: after warm-up a steady workload adds none).
void AresMemDriverCalls(int device, size_t *mallocs, size_t *frees, size_t *trims) {
  size_t m = 0, f = 0, t = 0;
  if (device >= 0 && device < kMaxDevices) {
    DeviceState *st = &g_devices[device];
    std::lock_guard<std::mutex> lock(st->mu);
    m = st->driverAllocs;
    f = st->driverFrees;
    t = st->trims;
  }
  if (mallocs) *mallocs = m;
  if (frees) *frees = f;
  if (trims) *trims = t;
}

size_t AresMemStreamEvents(int device, void *stream) {
  if (device < 0 || device >= kMaxDevices) return 0;
  DeviceState *st = &g_devices[device];
  std::lock_guard<std::mutex> lock(st->mu);
  std::map<hipEvent_t, int> seen;  // (blocks may share an event)
  for (auto &bin : st->bins)
    for (const ParkedBlock &b : bin.second)
      for (const FenceEvent &f : b.fence)
        if (f.stream == reinterpret_cast<hipStream_t>(stream)) seen[f.event] = 1;
  for (const FenceEvent &f : st->lastFence)
    if (f.stream == reinterpret_cast<hipStream_t>(stream)) seen[f.event] = 1;
  for (const FenceEvent &f : st->freeEvents)
    if (f.stream == reinterpret_cast<hipStream_t>(stream)) seen[f.event] = 1;
  return seen.size();
}

void AresMemTrimCache(int device) {
  if (device < 0 || device >= kMaxDevices) return;
  DeviceState *st = &g_devices[device];
  int current = 0;
  const bool switched = hipGetDevice(&current) == hipSuccess && current != device && hipSetDevice(device) == hipSuccess;
  {
    std::lock_guard<std::mutex> lock(st->mu);
    trim(st, 0);
  }
  if (switched) (void)hipSetDevice(current);
}

// The work that was reading the held blocks has been launched (or dropped): fence them like any
// other free — the fence is recorded now, behind that work.
void AresMemReleaseHeld(int device, uintptr_t tag) {
  if (device < 0 || device >= kMaxDevices) return;
  DeviceState *st = &g_devices[device];
  std::vector<void *> blocks;
  {
    std::lock_guard<std::mutex> lock(st->mu);
    for (size_t i = 0; i < st->held.size();) {
      if (tag == 0 || st->held[i].second == tag) {
        blocks.push_back(st->held[i].first);
        st->held[i] = st->held.back();
        st->held.pop_back();
      } else {
        i++;
      }
    }
  }
  if (blocks.empty()) return;
  int current = 0;
  const bool switched = hipGetDevice(&current) == hipSuccess && current != device && hipSetDevice(device) == hipSuccess;
  for (void *p : blocks) (void)pool_free(st, p);
  if (switched) (void)hipSetDevice(current);
}

DeviceMemoryFlags GetFlags(void) {
  // reference cuda_malloc.cu:36-42 / rmm_alloc.cu:84-91
  DeviceMemoryFlags f = DEVICE_MEMORY_IMPLEMENTATION_FLAG | HASH_REDUCTION_SUPPORT;
  if (use_pool()) f |= POOLED_MEMORY_FLAG;
  return f;
}

CGoCallResHandle HostAlloc(size_t bytes) {
  MemSlow slow_("HostAlloc");
  void *p = nullptr;
  MEM_TRY(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable), "Allocate");
  memset(p, 0, bytes);
  return ok(p);
}

CGoCallResHandle HostFree(void *p) {
  if (p) MEM_TRY(hipHostFree(p), "Free");
  return ok();
}

CGoCallResHandle HostMemCpy(void *dst, const void *src, size_t bytes) {
  if (memcpy(dst, src, bytes) != dst)
    return CGoCallResHandle{nullptr,
                            format_message("HostMemCpy", "Returned pointer does not match destination")};
  return ok();
}

CGoCallResHandle CreateCudaStream(int device) {
  MemSlow slow_("CreateCudaStream");
  MEM_TRY(hipSetDevice(device), "CreateCudaStream");
  hipStream_t s = nullptr;
  MEM_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "CreateCudaStream");
  DeviceState *st;
  MEM_TRY(device_state(device, &st), "CreateCudaStream");
  {
    std::lock_guard<std::mutex> lock(st->mu);
    st->streams.push_back(s);  // frees are fenced against every stream the host works on
    drop_last_fence(st);       // (a fence shared from now on names this stream too)
  }
  return ok(reinterpret_cast<void *>(s));
}

CGoCallResHandle WaitForCudaStream(void *s, int device) {
  MemSlow slow_("WaitForCudaStream");
  MEM_TRY(hipSetDevice(device), "WaitForCudaStream");
  notify_wait(device, s);
  MEM_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(s)), "WaitForCudaStream");
  return ok();
}

CGoCallResHandle DestroyCudaStream(void *s, int device) {
  MemSlow slow_("DestroyCudaStream");
  MEM_TRY(hipSetDevice(device), "DestroyCudaStream");
  flush_pending(device);
  if (s) {
    DeviceState *st;
    MEM_TRY(device_state(device, &st), "DestroyCudaStream");
    {
      std::lock_guard<std::mutex> lock(st->mu);
      for (size_t i = 0; i < st->streams.size(); i++)
        if (st->streams[i] == reinterpret_cast<hipStream_t>(s)) {
          st->streams[i] = st->streams.back();
          st->streams.pop_back();
          break;
        }
    }
    MEM_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(s)), "DestroyCudaStream");
    notify_stream_destroy(device, s);
    {
      // The stream is idle: whatever its fence events stand for has happened.  They are retired NOW, while the stream
      // exists (see FenceEvent) — destroyed, not recycled: nothing of the runtime's bookkeeping for this stream is kept.
      // An allocation that is waiting for a parked block's fence may hold events of THIS stream outside every list: those
      // waiters (counted per stream: a waiter parked on other, still busy streams' events is none of this call's business) are
      // let through first — this stream is idle, its events have fired; no new fence can name it, it left `streams` above —
      // and hand their events to freeEvents, where the sweep below finds them.
      std::unique_lock<std::mutex> lock(st->mu);
      st->waitersCv.wait(lock, [&] { return st->waiters.find(reinterpret_cast<hipStream_t>(s)) == st->waiters.end(); });
      drop_last_fence(st);  // (its events of this stream go to freeEvents unless a parked block shares them: swept either way)
      for (auto &bin : st->bins)
        for (ParkedBlock &b : bin.second)
          for (size_t i = 0; i < b.fence.size();)
            if (b.fence[i].stream == reinterpret_cast<hipStream_t>(s)) {
              auto info = st->events.find(b.fence[i].event);
              if (info == st->events.end() || --info->second.refs <= 0) {  // the last block that names it
                if (info != st->events.end()) st->events.erase(info);
                (void)hipEventDestroy(b.fence[i].event);
              }
              b.fence[i] = b.fence.back();
              b.fence.pop_back();
            } else {
              i++;
            }
      for (size_t i = 0; i < st->freeEvents.size();)
        if (st->freeEvents[i].stream == reinterpret_cast<hipStream_t>(s)) {
          (void)hipEventDestroy(st->freeEvents[i].event);
          st->freeEvents[i] = st->freeEvents.back();
          st->freeEvents.pop_back();
        } else {
          i++;
        }
    }
    MEM_TRY(hipStreamDestroy(reinterpret_cast<hipStream_t>(s)), "DestroyCudaStream");
  }
  return ok();
}

CGoCallResHandle DeviceAllocate(size_t bytes, int device) {
  MemSlow slow_("DeviceAllocate");
  MEM_TRY(hipSetDevice(device), "DeviceAllocate");
  DeviceState *st;
  MEM_TRY(device_state(device, &st), "DeviceAllocate");
  void *p = nullptr;
  MEM_TRY(pool_alloc(st, &p, bytes, /*zero=*/true), "DeviceAllocate");
  return ok(p);
}

// size of a live allocation (0 = unknown)
static size_t allocation_size(DeviceState *st, void *p) {
  if (use_pool()) {
    std::lock_guard<std::mutex> lock(st->mu);
    auto it = st->live.find(reinterpret_cast<uintptr_t>(p));
    if (it != st->live.end()) return it->second.rounded;
  }
  size_t bytes = 0;
  if (hipMemPtrGetInfo(p, &bytes) != hipSuccess) {
    (void)hipGetLastError();
    bytes = 0;
  }
  return bytes;
}

static hipError_t free_or_hold(DeviceState *st, void *p, int device) {
  if (p == nullptr) return hipSuccess;
  if (const uintptr_t tag = notify_free(device, p, allocation_size(st, p))) {
    std::lock_guard<std::mutex> lock(st->mu);
    st->held.emplace_back(p, tag);
    return hipSuccess;
  }
  return pool_free(st, p);
}

CGoCallResHandle DeviceFree(void *p, int device) {
  MemSlow slow_("DeviceFree");
  MEM_TRY(hipSetDevice(device), "DeviceFree");
  DeviceState *st;
  MEM_TRY(device_state(device, &st), "DeviceFree");
  MEM_TRY(free_or_hold(st, p, device), "DeviceFree");
  return ok();
}

CGoCallResHandle AsyncCopyHostToDevice(void *dst, void *src, size_t bytes, void *stream, int device) {
  MemSlow slow_("AsyncCopyHostToDevice");
  MEM_TRY(hipSetDevice(device), "AsyncCopyHostToDevice");
  notify_access(device, dst, bytes);
  notify_write(device, dst, bytes);
  if (bytes)
    MEM_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, reinterpret_cast<hipStream_t>(stream)),
            "AsyncCopyHostToDevice");
  return ok();
}

CGoCallResHandle AsyncCopyDeviceToDevice(void *dst, void *src, size_t bytes, void *stream, int device) {
  MEM_TRY(hipSetDevice(device), "AsyncCopyDeviceToDevice");
  notify_access(device, dst, bytes);
  notify_access(device, src, bytes);
  notify_write(device, dst, bytes);
  if (bytes)
    MEM_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)),
            "AsyncCopyDeviceToDevice");
  return ok();
}

CGoCallResHandle AsyncCopyDeviceToHost(void *dst, void *src, size_t bytes, void *stream, int device) {
  MemSlow slow_("AsyncCopyDeviceToHost");
  MEM_TRY(hipSetDevice(device), "AsyncCopyDeviceToHost");
  notify_access(device, src, bytes);
  if (bytes)
    MEM_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream)),
            "AsyncCopyDeviceToHost");
  return ok();
}

CGoCallResHandle GetDeviceCount(void) {
  int n = 0;
  MEM_TRY(hipGetDeviceCount(&n), "GetDeviceCount");
  return ok(reinterpret_cast<void *>(static_cast<intptr_t>(n)));
}

CGoCallResHandle GetDeviceGlobalMemoryInMB(int device) {
  hipDeviceProp_t prop;
  MEM_TRY(hipGetDeviceProperties(&prop, device), "GetDeviceGlobalMemoryInMB");
  return ok(reinterpret_cast<void *>(static_cast<intptr_t>(prop.totalGlobalMem / (1024 * 1024))));
}

// The Go host brackets one stage of one batch with these when `profiling=<stage>` is requested
// (query/aql_processor.go:808-822); under rocprofv3 they delimit the collection window.
CGoCallResHandle CudaProfilerStart(void) {
  (void)hipProfilerStart();
  (void)hipGetLastError();
  return ok();
}

CGoCallResHandle CudaProfilerStop(void) {
  flush_pending_current();
  MEM_TRY(hipDeviceSynchronize(), "cudaProfilerStop");
  (void)hipProfilerStop();
  (void)hipGetLastError();
  return ok();
}

CGoCallResHandle GetDeviceMemoryInfo(size_t *freeSize, size_t *totalSize, int device) {
  // pooled flavour of the reference (rmm_alloc.cu:239-246): real numbers instead of "Not supported"
  MEM_TRY(hipSetDevice(device), "GetDeviceMemoryInfo");
  MEM_TRY(hipMemGetInfo(freeSize, totalSize), "GetDeviceMemoryInfo");
  return ok();
}

CGoCallResHandle deviceMalloc(void **devPtr, size_t size) {
  DeviceState *st;
  MEM_TRY(current_device_state(&st), "deviceMalloc");
  MEM_TRY(pool_alloc(st, devPtr, size, /*zero=*/false), "deviceMalloc");
  return ok();
}

CGoCallResHandle deviceFree(void *devPtr) {
  int device = 0;
  MEM_TRY(hipGetDevice(&device), "deviceFree");
  DeviceState *st;
  MEM_TRY(device_state(device, &st), "deviceFree");
  MEM_TRY(free_or_hold(st, devPtr, device), "deviceFree");
  return ok();
}

CGoCallResHandle deviceMemset(void *devPtr, int value, size_t count) {
  notify_access_current(devPtr, count);
  notify_write_current(devPtr, count);
  MEM_TRY(hipMemset(devPtr, value, count), "deviceMemset");
  return ok();
}

CGoCallResHandle asyncCopyHostToDevice(void *dst, const void *src, size_t count, void *stream) {
  notify_access_current(dst, count);
  notify_write_current(dst, count);
  if (count)
    MEM_TRY(hipMemcpyAsync(dst, src, count, hipMemcpyHostToDevice, reinterpret_cast<hipStream_t>(stream)),
            "asyncCopyHostToDevice");
  return ok();
}

CGoCallResHandle asyncCopyDeviceToHost(void *dst, const void *src, size_t count, void *stream) {
  notify_access_current(src, count);
  if (count)
    MEM_TRY(hipMemcpyAsync(dst, src, count, hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream)),
            "asyncCopyDeviceToHost");
  return ok();
}

CGoCallResHandle waitForCudaStream(void *stream) {
  {
    int device = 0;
    if (hipGetDevice(&device) == hipSuccess) notify_wait(device, stream);
  }
  MEM_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)), "waitForCudaStream");
  return ok();
}

}  // extern "C"
