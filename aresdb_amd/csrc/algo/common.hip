// Host helpers shared by the entry points (see common.hpp).
#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <vector>

namespace ares {

std::atomic<uint32_t> g_envGeneration{1};
thread_local CallStream t_callStream{nullptr, false};

void slow_trace(const char *what, double ms) {
  static const char *path = getenv("ARES_RTC_TRACE");
  if (!path || !path[0]) return;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (FILE *o = fopen(path, "a")) {
    const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    fprintf(o, "%.3f slow %s %.3f ms\n", now, what, ms);
    fclose(o);
  }
}

namespace {
struct PinnedSlot {
  uint64_t *ptr = nullptr;
  uint32_t seq = 0;  // sequence number of the slot's last published read-back (read_back_u32)
  PinnedSlot() {
    // (+ 16 words: the sequence word a published read-back ends with sits behind the result words)
    if (hipHostMalloc(reinterpret_cast<void **>(&ptr), (kPinnedWords + 16) * sizeof(uint32_t), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
      (void)hipGetLastError();
      ptr = nullptr;
    } else {
      reinterpret_cast<volatile uint32_t *>(ptr)[kPinnedWords] = 0u;
    }
  }
  ~PinnedSlot() {
    if (ptr) (void)hipHostFree(ptr);
  }
};
PinnedSlot &pinned_slot() {
  thread_local PinnedSlot slot;
  if (!slot.ptr) throw AlgorithmError("ERROR: cannot allocate pinned result words");
  return slot;
}

// the words land in the calling thread's mapped pinned slot, then — system-scope release — the sequence word behind them
__global__ __launch_bounds__(256) void publish_words_kernel(const uint32_t *dev, uint32_t *pinned, int count, uint32_t *seqWord, uint32_t seq) {
  for (int i = threadIdx.x; i < count; i += 256) pinned[i] = dev[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(seqWord, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

uint64_t *pinned_words() { return pinned_slot().ptr; }

// A count-returning entry point's last step.  Measured on MI355X (tools/ubench_sync.hip, 2 us kernel): launch + 16-byte
// hipMemcpyAsync D2H + hipStreamSynchronize 25.5 us per round trip; launch + hipStreamSynchronize with the result written to
// mapped pinned memory by a kernel 14.1 us; the host POLLING a word in that memory 8.9 us — the copy command and the
// runtime's wait are most of what a small query's call costs.  So the words are published by a one-workgroup kernel behind
// the producer and the host spins on the sequence word; a stream that does not answer within a few milliseconds (a long
// kernel ahead of the read-back, or a failed one) is waited for the ordinary way.  ARES_READBACK=copy: the copy command.
void read_back_u32(const uint32_t *dev, uint32_t *host, int count, hipStream_t stream) {
  if (count > kPinnedWords) throw AlgorithmError("ERROR: read_back_u32: more words than the pinned slot holds");
  PinnedSlot &slot = pinned_slot();
  uint32_t *pinned = reinterpret_cast<uint32_t *>(slot.ptr);
  static EnvSwitch<bool> publish("ARES_READBACK", [](const char *e) { return !(e && strcmp(e, "copy") == 0); });
  if (!publish.get()) {
    hip_check(hipMemcpyAsync(pinned, dev, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, stream), "read back result");
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    for (int i = 0; i < count; i++) host[i] = pinned[i];
    return;
  }
  uint32_t *seqWord = pinned + kPinnedWords;
  const uint32_t seq = ++slot.seq ? slot.seq : ++slot.seq;  // (never 0: what a fresh slot holds)
  hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(256), 0, stream, dev, pinned, count, seqWord, seq);
  check_launch("read back result");
  const auto t0 = std::chrono::steady_clock::now();
  bool seen = false;
  for (uint32_t spins = 0;; spins++) {
    if (__atomic_load_n(seqWord, __ATOMIC_ACQUIRE) == seq) {
      seen = true;
      break;
    }
    __builtin_ia32_pause();
    if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(3)) break;
  }
  if (!seen) {  // a long (or failed) kernel is ahead: the runtime's wait — it sleeps, and it reports the stream's error
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (__atomic_load_n(seqWord, __ATOMIC_ACQUIRE) != seq) throw AlgorithmError("ERROR: read back result: the stream finished without publishing it");
  }
  for (int i = 0; i < count; i++) host[i] = pinned[i];
}

// ---- stream-local block cache ---------------------------------------------------------------------
namespace {
struct CachedBlock {
  void *ptr;
  size_t size;
};
struct StreamCache {
  std::map<size_t, std::vector<void *>> bins;  // rounded size -> free blocks
  size_t bytes = 0;
};
std::mutex g_cacheMutex;
std::map<std::pair<int, hipStream_t>, StreamCache> g_caches;
// blocks of destroyed streams (DestroyCudaStream synchronised the stream first: nothing enqueued touches them any
// more), per device: any stream of the device may take them.  The Go host makes and destroys two streams per
// query — its temporaries carry over to the next query instead of going through hipFree / hipMalloc.
std::map<int, StreamCache> g_orphans;
std::map<void *, size_t> g_blockSize;  // every block handed out or cached -> rounded size
std::map<void *, bool> g_handedOut;    // invariant check: a block is either in a cache or with exactly one user
size_t g_cachedBytes = 0;
std::map<void *, uint64_t> g_releasedAt;  // cached block -> the release count at which it was cached (oldest first out)
uint64_t g_releases = 0;

void check_take(void *p) {  // caller holds g_cacheMutex
  bool &out = g_handedOut[p];
  if (out) {
    fprintf(stderr, "libalgorithm: temporary block %p handed out twice\n", p);
    abort();
  }
  out = true;
}
void check_give(void *p) {  // caller holds g_cacheMutex
  auto it = g_handedOut.find(p);
  if (it == g_handedOut.end() || !it->second) {
    fprintf(stderr, "libalgorithm: temporary block %p released twice (or never taken)\n", p);
    abort();
  }
  it->second = false;
}

size_t cache_bin(size_t bytes) {
  if (bytes < 256) return 256;
  const int top = 63 - __builtin_clzll(static_cast<unsigned long long>(bytes));
  const size_t step = static_cast<size_t>(1) << (top > 3 ? top - 3 : 0);
  return (bytes + step - 1) / step * step;
}

// frees every cached block (all streams of all devices); the caller holds g_cacheMutex.  hipFree
// waits for outstanding work, so blocks still referenced by enqueued kernels stay valid until then.
void drop_all_cached() {
  auto drop = [](StreamCache &c) {
    for (auto &bin : c.bins)
      for (void *p : bin.second) {
        (void)hipFree(p);
        g_blockSize.erase(p);
        g_handedOut.erase(p);
      }
    c.bins.clear();
    c.bytes = 0;
  };
  for (auto &kv : g_caches) drop(kv.second);
  for (auto &kv : g_orphans) drop(kv.second);
  g_cachedBytes = 0;
}
}  // namespace

// The cache has outgrown its bound: the blocks cached LONGEST go back to the driver until three quarters of the bound are
// left — what a query in steady state releases and takes again batch after batch stays (dropping everything made a
// working set near the bound thrash: gigabytes of hipMalloc per batch, 19 -> 275 ms per 1 B rows of archive batches in one
// run out of four), the leftovers of earlier sizes (a workspace that has grown since) go.  The caller holds g_cacheMutex.
void evict_oldest_cached(size_t bound) {
  struct Victim {
    uint64_t at;
    void *ptr;
    size_t size;
    StreamCache *cache;
  };
  std::vector<Victim> all;
  auto collect = [&](StreamCache &c) {
    for (auto &bin : c.bins)
      for (void *p : bin.second) {
        auto it = g_releasedAt.find(p);
        all.push_back(Victim{it == g_releasedAt.end() ? 0 : it->second, p, bin.first, &c});
      }
  };
  for (auto &kv : g_caches) collect(kv.second);
  for (auto &kv : g_orphans) collect(kv.second);
  std::sort(all.begin(), all.end(), [](const Victim &a, const Victim &b) { return a.at < b.at; });
  {
    size_t real = 0;
    for (const Victim &v : all) real += v.size;
    char what[160];
    snprintf(what, sizeof(what), "temporary: cache over its bound: %.1f MB counted, %.1f MB in %zu blocks, bound %.1f MB", g_cachedBytes / 1e6, real / 1e6,
             all.size(), bound / 1e6);
    slow_trace(what, 0.0);
  }
  for (const Victim &v : all) {
    if (g_cachedBytes <= bound / 4 * 3) break;
    std::vector<void *> &bin = v.cache->bins[v.size];
    auto it = std::find(bin.begin(), bin.end(), v.ptr);
    if (it == bin.end()) continue;
    bin.erase(it);
    v.cache->bytes -= v.size;
    g_cachedBytes -= v.size;
    (void)hipFree(v.ptr);
    g_blockSize.erase(v.ptr);
    g_handedOut.erase(v.ptr);
    g_releasedAt.erase(v.ptr);
  }
}

void (*g_memNoteActivity)() = nullptr;  // AresMemNoteActivity of the sibling libmem.so (transform.hip resolves it)
void (*g_memNoteWrite)(int, const void *, size_t) = nullptr;  // AresMemNoteWrite of the sibling libmem.so (transform.hip resolves it)
int current_device() {
  int device = 0;
  (void)hipGetDevice(&device);
  return device;
}
void mem_note_dim_rows(int device, const DimensionVector &v, size_t firstRow, size_t rows) {
  if (!g_memNoteWrite || !v.DimValues || rows == 0 || v.VectorCapacity <= 0) return;
  const size_t cap = static_cast<size_t>(v.VectorCapacity);
  size_t off = 0;
  int nd = 0;
  for (int w = 0; w < NUM_DIM_WIDTH; w++) {
    const size_t width = static_cast<size_t>(1) << (NUM_DIM_WIDTH - 1 - w);
    for (int k = 0; k < v.NumDimsPerDimWidth[w]; k++) {
      g_memNoteWrite(device, v.DimValues + off + width * firstRow, width * rows);
      off += width * cap;
      nd++;
    }
  }
  for (int d = 0; d < nd; d++) g_memNoteWrite(device, v.DimValues + off + static_cast<size_t>(d) * cap + firstRow, rows);
}
void mem_note_vector_all(int device, const DimensionVector &v) {
  if (!g_memNoteWrite || v.VectorCapacity <= 0) return;
  const size_t cap = static_cast<size_t>(v.VectorCapacity);
  mem_note_dim_rows(device, v, 0, cap);
  if (v.HashValues) g_memNoteWrite(device, v.HashValues, 8 * cap);
  if (v.IndexVector) g_memNoteWrite(device, v.IndexVector, 4 * cap);
}
void (*g_memTrimCache)(int) = nullptr;  // AresMemTrimCache of the sibling libmem.so (transform.hip resolves it)

// A stream is being destroyed (the host has synchronised it): its cached temporaries become the device's —
// the Go host creates and destroys two streams per query, blocks cached under dead handles would only pile up,
// and giving them back to the driver costs a device-synchronising hipFree each plus a hipMalloc in the next query.
void stream_cache_purge(int device, hipStream_t stream) {
  // The blocks stay with the device for its other streams (ARES_TEMP_ORPHANS=0: give them back to the driver with
  // hipFree, which synchronises the device, as rounds 2 and 3 did).  Round 3 saw rare off-by-one results and aborts with
  // the blocks kept and blamed the reuse; round 4 traced them to the HIP runtime being handed events of destroyed
  // streams (libmem's fences, the lazy compactions' error words) — a use after free inside the runtime that the
  // device-wide synchronisation of hipFree merely made unlikely.  Those events now die with their stream
  // (mem/memory.hip FenceEvent, transform.hip hook_on_stream_destroy; profiles/r4_race_hunt.md).
  static const bool keep = [] {
    const char *e = getenv("ARES_TEMP_ORPHANS");
    return !(e && e[0] == '0');
  }();
  if (!keep) {
    std::vector<void *> blocks;
    {
      std::lock_guard<std::mutex> lock(g_cacheMutex);
      auto it = g_caches.find({device, stream});
      if (it == g_caches.end()) return;
      for (auto &bin : it->second.bins)
        for (void *p : bin.second) {
          blocks.push_back(p);
          g_blockSize.erase(p);
          g_handedOut.erase(p);
        }
      g_cachedBytes -= it->second.bytes;
      g_caches.erase(it);
    }
    for (void *p : blocks) (void)hipFree(p);
    return;
  }
  static const bool destroySync = [] {  // diagnostics: is it the device-wide synchronisation that the old path had?
    const char *e = getenv("ARES_DESTROY_SYNC");
    return e && e[0] == '1';
  }();
  if (destroySync) (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lock(g_cacheMutex);
  auto it = g_caches.find({device, stream});
  if (it == g_caches.end()) return;
  StreamCache &o = g_orphans[device];
  for (auto &bin : it->second.bins) {
    std::vector<void *> &dst = o.bins[bin.first];
    dst.insert(dst.end(), bin.second.begin(), bin.second.end());
  }
  o.bytes += it->second.bytes;
  g_caches.erase(it);
}

// gives every cached block of the device back to the driver (the sibling library ran out of memory)
void stream_cache_trim(int device) {
  std::vector<void *> blocks;
  {
    std::lock_guard<std::mutex> lock(g_cacheMutex);
    for (auto &kv : g_caches) {
      if (kv.first.first != device) continue;
      for (auto &bin : kv.second.bins)
        for (void *p : bin.second) {
          blocks.push_back(p);
          g_blockSize.erase(p);
          g_handedOut.erase(p);
        }
      kv.second.bins.clear();
      g_cachedBytes -= kv.second.bytes;
      kv.second.bytes = 0;
    }
    auto o = g_orphans.find(device);
    if (o != g_orphans.end()) {
      for (auto &bin : o->second.bins)
        for (void *p : bin.second) {
          blocks.push_back(p);
          g_blockSize.erase(p);
          g_handedOut.erase(p);
        }
      g_cachedBytes -= o->second.bytes;
      g_orphans.erase(o);
    }
  }
  for (void *p : blocks) (void)hipFree(p);
}

namespace {
// A cached block for a request of `rounded` bytes; the caller holds g_cacheMutex.  The exact bin first.  A large request
// (a HashReduce workspace: hundreds of megabytes to gigabytes) also takes the smallest LARGER cached block: a cached block
// is idle memory whatever its size, and asking the driver for fresh memory instead costs 28 ms per GB at best
// (profiles/r4_experiments.md "cold start").
constexpr size_t kFitAnyFrom = static_cast<size_t>(64) << 20;
void *take_cached(StreamCache &c, size_t rounded, bool larger) {
  auto it = c.bins.find(rounded);
  if (it == c.bins.end() || it->second.empty()) {
    if (!larger || rounded < kFitAnyFrom) return nullptr;
    it = c.bins.upper_bound(rounded);
    while (it != c.bins.end() && it->second.empty()) ++it;
    // ... up to twice the request: a 260 MB temporary (a decoded run-length column) that sits on a 4 GB workspace block for
    // a whole batch makes the workspace's next user go to the driver (archive batches: 4.3 + 1.5 GB of hipMalloc per batch)
    if (it == c.bins.end() || it->first > 2 * rounded) return nullptr;
  }
  void *p = it->second.back();
  it->second.pop_back();
  c.bytes -= it->first;
  g_cachedBytes -= it->first;
  g_releasedAt.erase(p);
  check_take(p);
  return p;
}
}  // namespace

void *stream_alloc(size_t bytes, hipStream_t stream) {
  const size_t rounded = cache_bin(bytes);
  int device = 0;
  hip_check(hipGetDevice(&device), "hipGetDevice");
  {
    std::lock_guard<std::mutex> lock(g_cacheMutex);
    StreamCache &c = g_caches[{device, stream}];
    auto o = g_orphans.find(device);
    for (int larger = 0; larger < 2; larger++) {  // exact bins of the stream and of the device first, then anything that fits
      if (void *p = take_cached(c, rounded, larger != 0)) return p;
      if (o != g_orphans.end())
        if (void *p = take_cached(o->second, rounded, larger != 0)) return p;
    }
  }
  void *p = nullptr;
  thread_local char what[64];
  snprintf(what, sizeof(what), "temporary: hipMalloc of %.1f MB", static_cast<double>(rounded) / 1e6);
  SlowScope slow(what);
  // ARES_TEMP_POOL_ALLOC=1 (experiment for the next round, profiles/r4_experiments.md "cold start"): fresh blocks from the
  // stream-ordered pool — hipMallocAsync of 2 GB took 10 ms where hipMalloc took 61-318.  Blocks are never freed INTO the
  // pool (no reuse inside it); a block that leaves the cache goes through hipFree like the others.
  static const bool poolAlloc = [] {
    const char *v = getenv("ARES_TEMP_POOL_ALLOC");
    return v && v[0] == '1';
  }();
  hipError_t e = poolAlloc ? hipMallocAsync(&p, rounded, stream) : hipMalloc(&p, rounded);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    {
      std::lock_guard<std::mutex> lock(g_cacheMutex);
      drop_all_cached();
    }
    if (g_memTrimCache) g_memTrimCache(device);  // the sibling allocator's parked blocks too
    hip_check(hipMalloc(&p, rounded), "hipMalloc");
  }
  std::lock_guard<std::mutex> lock(g_cacheMutex);
  g_blockSize[p] = rounded;
  check_take(p);
  return p;
}

namespace {
// idle = the block's last use has been waited for on the host: it goes to the device's shared bins, any stream may take it
void release_block(void *ptr, hipStream_t stream, bool idle) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  SlowScope slow("temporary: release");
  std::lock_guard<std::mutex> lock(g_cacheMutex);
  auto it = g_blockSize.find(ptr);
  if (it == g_blockSize.end()) return;
  const size_t rounded = it->second;
  check_give(ptr);
  StreamCache &c = idle ? g_orphans[device] : g_caches[{device, stream}];
  c.bins[rounded].push_back(ptr);
  c.bytes += rounded;
  g_cachedBytes += rounded;
  g_releasedAt[ptr] = ++g_releases;
  // keep the cache below an eighth of the device: the longest-cached blocks go when it outgrows that
  static const size_t cap = [] {
    if (const char *e = getenv("ARES_TEMP_CACHE_MB"))  // (tests: a bound small enough for evictions to happen all the time)
      if (atol(e) > 0) return static_cast<size_t>(atol(e)) << 20;
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) {
      (void)hipGetLastError();
      return static_cast<size_t>(32) << 30;
    }
    return totalB / 8;
  }();
  if (g_cachedBytes > cap) evict_oldest_cached(cap);
}
}  // namespace

void temp_stats(size_t *handedOut, size_t *cached) {
  std::lock_guard<std::mutex> lock(g_cacheMutex);
  size_t out = 0;
  for (auto &kv : g_handedOut)
    if (kv.second) {
      auto it = g_blockSize.find(kv.first);
      if (it != g_blockSize.end()) out += it->second;
    }
  if (handedOut) *handedOut = out;
  if (cached) *cached = g_cachedBytes;
}

void stream_release(void *ptr, hipStream_t stream) { release_block(ptr, stream, false); }
void stream_release_idle(void *ptr) { release_block(ptr, nullptr, true); }

// ---- kernel timing -------------------------------------------------------------------------------
namespace {
struct TimedLaunch {
  const char *name;
  hipEvent_t start, stop;
  hipStream_t stream;
  bool resolved;  // ms holds the duration, the events are gone (their stream was destroyed)
  float ms;
};
std::mutex g_profMutex;
std::atomic<int> g_profEnabled{0};
std::vector<TimedLaunch> g_launches;
std::vector<hipEvent_t> g_freeEvents;

hipEvent_t take_event() {
  if (!g_freeEvents.empty()) {
    hipEvent_t e = g_freeEvents.back();
    g_freeEvents.pop_back();
    return e;
  }
  hipEvent_t e;
  hip_check(hipEventCreate(&e), "hipEventCreate");
  return e;
}
}  // namespace

KernelTimer::KernelTimer(const char *name, hipStream_t stream) : slot_(-1), stream_(stream), slow_(name) {
  if (!g_profEnabled.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lock(g_profMutex);
  TimedLaunch t{name, take_event(), take_event(), stream, false, 0.0f};
  hip_check(hipEventRecord(t.start, stream), "hipEventRecord");
  g_launches.push_back(t);
  slot_ = static_cast<int>(g_launches.size()) - 1;
}

void profiler_stream_gone(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_profMutex);
  for (auto &t : g_launches) {
    if (t.resolved || t.stream != stream) continue;
    float ms = 0;
    if (hipEventSynchronize(t.stop) != hipSuccess || hipEventElapsedTime(&ms, t.start, t.stop) != hipSuccess) {
      (void)hipGetLastError();
      ms = -1.0f;  // (not counted)
    }
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
    t.resolved = true;
    t.ms = ms;
  }
}

size_t profiler_stream_events(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_profMutex);
  size_t n = 0;
  for (auto &t : g_launches) n += (!t.resolved && t.stream == stream) ? 2 : 0;
  return n;
}

KernelTimer::~KernelTimer() {
  if (slot_ < 0) return;
  std::lock_guard<std::mutex> lock(g_profMutex);
  if (slot_ < static_cast<int>(g_launches.size()) && !g_launches[slot_].resolved) (void)hipEventRecord(g_launches[slot_].stop, stream_);
}

}  // namespace ares

// Exported (not part of the reference ABI; declared in include/ares_extensions.h).
extern "C" void AresReloadEnv(void) { ares::g_envGeneration.fetch_add(1, std::memory_order_acq_rel); }

extern "C" void AresProfilerEnable(int on) {
  std::lock_guard<std::mutex> lock(ares::g_profMutex);
  ares::g_profEnabled.store(on ? 1 : 0);
  if (on) {
    // (events are destroyed, not recycled: a recycled event still belongs to the stream of its last record)
    for (auto &t : ares::g_launches)
      if (!t.resolved) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
      }
    ares::g_launches.clear();
  }
}

// Writes "name launches total_ms\n" lines for everything recorded since the last enable; returns
// the number of bytes needed.  The caller must have synchronised the streams it used.
extern "C" size_t AresProfilerReport(char *buf, size_t len) {
  std::lock_guard<std::mutex> lock(ares::g_profMutex);
  std::map<std::string, std::pair<long, double>> agg;
  for (auto &t : ares::g_launches) {
    float ms = 0;
    if (t.resolved) {
      if (t.ms >= 0) {
        auto &a = agg[t.name];
        a.first++;
        a.second += t.ms;
      }
    } else if (hipEventSynchronize(t.stop) == hipSuccess && hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
      auto &a = agg[t.name];
      a.first++;
      a.second += ms;
    } else {
      (void)hipGetLastError();
    }
  }
  std::string out;
  char line[256];
  for (auto &kv : agg) {
    snprintf(line, sizeof(line), "%s %ld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    out += line;
  }
  if (buf && len) {
    const size_t n = out.size() < len - 1 ? out.size() : len - 1;
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}
