// Host-side plumbing shared by every entry point of libalgorithm.so (MI355X / gfx950).
//
// Mirrors the reference's ABI wrapper pattern (query/filter.cu:141-165, query/utils.cu:44-59):
// select the device, run, convert any C++ exception into a strdup()'ed message — an exception
// never crosses the C boundary.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "ares_algorithm.h"

namespace ares {

class AlgorithmError : public std::runtime_error {
 public:
  explicit AlgorithmError(const std::string &m) : std::runtime_error(m) {}
};

// ARES_RTC_TRACE=<file>: host time of anything that took longer than 5 ms on the calling thread (cold-start diagnostics)
void slow_trace(const char *what, double ms);
class SlowScope {
 public:
  explicit SlowScope(const char *what) : what_(what), t0_(std::chrono::steady_clock::now()) {}
  ~SlowScope() {
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count();
    if (ms > 5.0) slow_trace(what_, ms);
  }
  SlowScope(const SlowScope &) = delete;
  SlowScope &operator=(const SlowScope &) = delete;

 private:
  const char *what_;
  std::chrono::steady_clock::time_point t0_;
};
inline void hip_check(hipError_t e, const char *what) {
  if (e != hipSuccess) {
    (void)hipGetLastError();
    throw AlgorithmError(std::string("ERROR: ") + what + ": " + hipGetErrorString(e));
  }
}

// every checked runtime call is timed on the way (two clock reads): ARES_RTC_TRACE names the ones that blocked the host
template <class F>
inline void hip_check_timed(F &&call, const char *what) {
  SlowScope slow(what);
  hip_check(call(), what);
}
#define hip_check(expr, what) hip_check_timed([&]() -> hipError_t { return (expr); }, what)

// The sibling libmem.so shares the fence events of a free with the frees that follow it while nothing was submitted to the
// device in between (mem/memory.hip "shared fences"): every entry point and every kernel launch of this library — launches
// happen inside libmem's hooks too — says so (AresMemNoteActivity; null: no such libmem, nothing is shared).
extern void (*g_memNoteActivity)();
inline void mem_note_activity() {
  if (g_memNoteActivity) g_memNoteActivity();
}
// Checks the launch that was just enqueued (reference CheckCUDAError, utils.cu:44-59).
inline void check_launch(const char *what) {
  mem_note_activity();
  hip_check(hipGetLastError(), what);
}

// launches the transforms held back for cross-call fusion on `device` (transform.hip)
void flush_deferred(int device);
// second-stage fusion (transform.hip, include/ares_extensions.h)
bool fuse_pending_into_hash_reduce(int device, hipStream_t stream, const DimensionVector &in, const uint8_t *inValues,
                                   const DimensionVector &out, uint8_t *outValues, int valueBytes, int length, int aggFunc,
                                   int *groups);
void invalidate_filter_journal(int device, const uint32_t *indexVector);
// ARES_HASH_REDUCE=global: every HashReduce takes the global-table path (hash_reduce.hip)
bool global_table_forced();
// true when the sibling libmem.so reports waits, frees and copies to this library (transform.hip)
bool deferral_hooks_active();
// something writes (or frees) [ptr, ptr + bytes): partition-grouped results that overlap are no longer
// trusted (hash_reduce_lds.hip)
void grouped_note_write(int device, const void *ptr, size_t bytes);
void grouped_note_write(int device, const DimensionVector &v);  // the whole vector
// [ptr, ptr + bytes) is about to be read: measure rows that an image-mode HashReduce left unwritten (they are defined by
// the query's table image) are written first (hash_reduce_lds.hip)
void grouped_materialize_for_read(int device, const void *ptr, size_t bytes);
// out of memory elsewhere: the idle range buffers and table images of the device go back to the driver
void grouped_trim(int device);
void flush_deferred_for_inputs(int device, const void *a, size_t aBytes, const void *b, size_t bBytes);
// the same for an entry point that reads a whole dimension vector and a value vector
void flush_deferred_for_vector(int device, const DimensionVector &v, const void *values, size_t valueBytes);
// skipped work whose outputs lie in ranges that are about to be overwritten is dropped, never launched
void drop_skipped_outputs(int device, const void *a, size_t aBytes, const void *b, size_t bBytes);

// ---- lazy fills (transform.hip): buffers defined as a repeated 4- or 8-byte pattern that nobody has written yet ----
// false = deferral is off (no sibling libmem.so hooks): the caller writes the bytes itself
bool defer_fill(int device, hipStream_t stream, void *dst, size_t bytes, uint64_t pattern, int unit);
bool pending_fill_exact(int device, const void *dst, size_t bytes, uint64_t *pattern, int *unit);
// a lazy fill covering rows [*prev, length) of a vector of `width`-byte values
bool pending_fill_tail(int device, const void *base, int width, int length, int *prev, uint64_t *pattern);
void retire_fills_for_write(int device, const void *ptr, size_t bytes);   // a kernel of the caller overwrites the range
void materialize_fills_for_read(int device, const void *ptr, size_t bytes);  // a kernel of the caller reads the range
// queued (or skipped) transforms whose outputs overlap [ptr, ptr + bytes) are launched now: for an entry point that reads
// the range with a kernel but does not flush
void launch_pending_writers(int device, const void *ptr, size_t bytes);
// `indexVector` is defined as iota(0 .. n) and not written yet (InitIndexVector is lazy)
bool virtual_iota_peek(int device, const uint32_t *indexVector, int n, bool consume = false);
// the caller was handed `indexVector` and reads it with a kernel: a lazy iota a consumer left behind is written now
void materialize_index_vector(int device, const uint32_t *indexVector);
// ... and the same for every buffer of a dimension vector (dimension rows, hash vector, index vector), lazy fills included
void settle_dimension_vector(int device, const DimensionVector &v, bool rowsOnly = false);
// bytes of stream temporaries handed out / cached (AresTempStats)
void temp_stats(size_t *handedOut, size_t *cached);

// The stream of the entry point the calling thread is executing.  Work that was DEFINED on another stream (a lazy
// fill, a lazy iota, a lazy compaction) and is written on that stream at one of this call's flush points is waited for
// before the call's own kernels — which run on this stream — read it (transform.hip: order_before_caller).  Calls
// that arrive from libmem.so (copies, frees) have no stream here: they wait for whatever they launch.
struct CallStream {
  hipStream_t stream;
  bool known;
};
extern thread_local CallStream t_callStream;
class CallStreamScope {
 public:
  CallStreamScope(void *stream, const char *entry) : saved_(t_callStream), slow_(entry) {
    t_callStream = CallStream{reinterpret_cast<hipStream_t>(stream), true};
  }
  ~CallStreamScope() { t_callStream = saved_; }
  CallStreamScope(const CallStreamScope &) = delete;
  CallStreamScope &operator=(const CallStreamScope &) = delete;

 private:
  CallStream saved_;
  SlowScope slow_;
};

// NOFLUSH: only for the entry points which decide themselves whether to queue or flush.  Every entry point names its
// stream parameter `cudaStream` (the reference's spelling).
#define ARES_ABI_BEGIN_NOFLUSH(device)                 \
  CGoCallResHandle resHandle = {nullptr, nullptr};     \
  try {                                                \
    ares::hip_check(hipSetDevice(device), "hipSetDevice");  \
    ares::CallStreamScope callStreamScope_(cudaStream, __func__);  \
    ares::mem_note_activity();                                      \
    (void)ares::deferral_hooks_active(); /* write tracking is on before this entry point's first kernel */

#define ARES_ABI_BEGIN(device)     \
  ARES_ABI_BEGIN_NOFLUSH(device)   \
  ares::flush_deferred(device);

#define ARES_ABI_END(name)                                            \
  }                                                                   \
  catch (std::exception & e) {                                        \
    fprintf(stderr, "Exception happened when doing %s: %s\n", name, e.what()); \
    resHandle.pStrErr = strdup(e.what());                             \
  }                                                                   \
  return resHandle;

// Behaviour switches read from the environment (ARES_HASH_REDUCE, ARES_GROUPED, ...).  A switch latches the
// value it parsed and re-reads it only after AresReloadEnv() (include/ares_extensions.h) was called: a
// test that flips a variable inside one process says so, a server never pays for getenv on the query path.
extern std::atomic<uint32_t> g_envGeneration;
template <typename T>
class EnvSwitch {
 public:
  EnvSwitch(const char *name, T (*parse)(const char *)) : name_(name), parse_(parse) {}
  T get() {
    const uint32_t g = g_envGeneration.load(std::memory_order_acquire);
    if (gen_.load(std::memory_order_acquire) != g) {
      value_.store(parse_(getenv(name_)), std::memory_order_relaxed);
      gen_.store(g, std::memory_order_release);
    }
    return value_.load(std::memory_order_relaxed);
  }

 private:
  const char *name_;
  T (*parse_)(const char *);
  std::atomic<uint32_t> gen_{0};  // g_envGeneration starts at 1
  std::atomic<T> value_{};
};

inline void *int_result(int64_t v) { return reinterpret_cast<void *>(static_cast<intptr_t>(v)); }

// Stream-local temporary device memory (tile descriptors, hash-partition regions, radix-sort
// ping-pong buffers).  Blocks are cached per (device, stream) in 1/8-octave size bins: a block
// released on stream S is only ever handed to a later request on the SAME stream, so plain stream
// order makes the reuse safe — no events, no driver call after warm-up, and no reliance on
// cross-stream reuse inside the runtime's own pool.
void *stream_alloc(size_t bytes, hipStream_t stream);
void stream_release(void *ptr, hipStream_t stream);
void stream_release_idle(void *ptr);  // the last use has been waited for: the block goes to the device's shared bins
void stream_cache_purge(int device, hipStream_t stream);  // the stream is being destroyed
void stream_cache_trim(int device);                       // out of memory elsewhere: give everything back
// Write reports towards the sibling libmem.so (AresMemNoteWrite, include/ares_extensions.h): it clears a freed
// block only where something wrote.  Every entry point reports the outputs of its kernels; the three things
// that may never be written — a deferred InitIndexVector, a lazily compacted index vector, transforms that
// HashReduce consumed — report when (if) they are materialised.
extern void (*g_memNoteWrite)(int device, const void *ptr, size_t bytes);
inline void mem_note_write(int device, const void *ptr, size_t bytes) {
  if (g_memNoteWrite && ptr && bytes) g_memNoteWrite(device, ptr, bytes);
}
int current_device();
// rows [firstRow, firstRow + rows) of every dimension (values + validity byte) of a dimension vector
void mem_note_dim_rows(int device, const DimensionVector &v, size_t firstRow, size_t rows);
// everything a dimension vector owns, to its capacity: dimension values + validity, hash vector, index vector
void mem_note_vector_all(int device, const DimensionVector &v);
extern void (*g_memTrimCache)(int device);                // sibling libmem.so's AresMemTrimCache, when present

class StreamBuffer {
 public:
  StreamBuffer(size_t bytes, hipStream_t stream) : stream_(stream) { ptr_ = stream_alloc(bytes ? bytes : 16, stream); }
  ~StreamBuffer() {
    if (ptr_ && idle_) stream_release_idle(ptr_);
    else if (ptr_) stream_release(ptr_, stream_);
  }
  // The host has WAITED for the stream after the last kernel that touches the buffer: on release it goes to the device's
  // shared bins instead of the stream's own.  (The Go host alternates two streams per query; HashReduce returns a count,
  // so it has synchronised its stream when it lets go of its workspace — one workspace serves both streams.)
  void mark_idle() { idle_ = true; }
  StreamBuffer(const StreamBuffer &) = delete;
  StreamBuffer &operator=(const StreamBuffer &) = delete;
  template <typename T>
  T *as() const { return static_cast<T *>(ptr_); }
  void *get() const { return ptr_; }

 private:
  void *ptr_ = nullptr;
  hipStream_t stream_;
  bool idle_ = false;
};

// A few pinned host words per calling thread: where count-returning entry points receive their
// result (D2H of 4-16 bytes + stream sync).  ABI calls arrive on arbitrary goroutine threads.
uint64_t *pinned_words();

// Reads `count` 32-bit words written by a kernel at `dev` back to the host; synchronises stream.
constexpr int kPinnedWords = 4096 + 16;  // the thread's pinned result slot: a filter reads one partial count per workgroup back
void read_back_u32(const uint32_t *dev, uint32_t *host, int count, hipStream_t stream);

// the stream is being destroyed (it is idle): timing events recorded on it are resolved and destroyed now — an event
// must not be touched once its stream is gone (mem/memory.hip: FenceEvent)
void profiler_stream_gone(hipStream_t stream);
size_t profiler_stream_events(hipStream_t stream);  // unresolved timing events recorded on `stream`
// Optional per-kernel timing with HIP events on the launch stream (off by default).  bench.py
// switches it on through the exported AresProfilerEnable / AresProfilerReport pair to obtain the
// average duration of every kernel inside the timed region (the roofline figures).
class KernelTimer {
 public:
  KernelTimer(const char *name, hipStream_t stream);
  ~KernelTimer();

 private:
  int slot_;
  hipStream_t stream_;
  SlowScope slow_;  // a launch call that blocks the host (first use of a code object, scratch growth, a full queue)
};

// Launch + error check (+ timing when profiling is enabled).
#define ARES_LAUNCH(name, kernel, grid, block, stream, ...)                              \
  do {                                                                                   \
    ares::KernelTimer timer_(name, stream);                                              \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__);        \
    ares::check_launch(name);                                                            \
  } while (0)

// Grid size helper: enough blocks to cover `tiles`, capped so that huge inputs are processed by
// a few waves of blocks per CU (256 CUs x 8 blocks) with a grid-stride / ticket loop.
inline int capped_grid(int64_t tiles, int cap = 256 * 8) {
  if (tiles < 1) tiles = 1;
  // ARES_GRID_PER_CU=n (experiments): every grid-stride kernel runs with at most 256 n workgroups
  static const int perCU = [] {
    const char *e = getenv("ARES_GRID_PER_CU");
    return e ? atoi(e) : 0;
  }();
  if (perCU > 0 && cap >= 256 && 256 * perCU < cap) cap = 256 * perCU;  // (lowers a caller's cap, never raises it: callers size buffers by it)
  return static_cast<int>(tiles < cap ? tiles : cap);
}

}  // namespace ares
