// InitIndexVector, Unary/BinaryTransform, Unary/BinaryFilter for MI355X (gfx950): the HOST side — operand binding, the
// deferral state machine, the hooks libmem.so calls, the ABI entry points.  Every kernel lives in transform_kernels.hpp.
//
// ---- the deferral state machine: what is held back, and the invariants that keep it unobservable ---------------------
// One DeferState per device, one mutex (DeferLock); nothing below synchronises a stream while it holds the lock (late
// launches that the host believes done are synchronised after the unlock: t_syncAfterUnlock).  The things a call may
// DEFINE instead of doing:
//   * pending[(device, stream)]   root transforms of the hot shape, one queue per stream, launched together
//                                 (transform_multi_kernel) or consumed by HashReduce's fused scan;
//   * limbo[(device, stream)]     a queue HashReduce consumed: its outputs were never written, it stays launchable until
//                                 the stream's next InitIndexVector (begin_batch) or until its outputs are freed;
//   * journals[index vector]      the hot-shape filters applied to the vector since InitIndexVector: what the fused scan
//                                 replays instead of reading the vector;
//   * compactions[index vector]   filters that were counted (row space: `todo`, survivor bits) or evaluated (predicate
//                                 bytes) but whose compaction of the vector has not run;
//   * iotas[index vector]         InitIndexVector recorded, not written ("virtual iota");
//   * fills[first byte]           a buffer defined as a repeated 4- / 8-byte pattern (constant measures, the hash vector
//                                 of a query without dimensions);
//   * (hash_reduce_lds.hip)       measure rows defined by a table image ("lazy values");
//   * expansions[(device, stream)] run-length encoded (mode-3) columns of an archive batch, decoded ONCE per batch into a
//                                 stream temporary laid out like a mode-2 column; the hot-shape machinery below (row-space
//                                 filters, queued transforms, the fused scans) then reads the copy.  Every definition that
//                                 reads a copy keeps it alive (`keep`); the per-batch cache dies at the stream's next
//                                 InitIndexVector and whenever the source column is written or freed;
//   * sorts[index vector]         Sort over rows whose transforms are still pending: the hash vector (a marker entry of
//                                 `fills`) and the index vector (its `iotas` entry, marked `sorted`) are defined as "what
//                                 Sort would leave"; Reduce consumes the definition with the pending transforms
//                                 (sort_reduce_fused.hip) and leaves its own (`reduced`: the input's hash and index vector
//                                 and the output's index vector are what a replay — skipped transforms, Sort, Reduce —
//                                 would write: materialize_sort).
// Invariants (each one is what a reader of device memory relies on; the sequence fuzzer reads every buffer at random
// points and the soak test does so from four threads):
//   I1  Before ANY byte of device memory is read by something other than the consumer a definition was made for — a copy
//       (hook_on_access), an entry point that is handed the buffer (flush_deferred_for_inputs / _for_vector,
//       settle_dimension_vector, launch_pending_writers) — every definition that covers the byte is written, in call order,
//       on the stream it was made on, and that stream is synchronised before the reader proceeds if the reader runs
//       elsewhere (order_before_caller).
//   I2  Before a byte is overwritten or freed, definitions inside the range are retired (retire_fills, drop_skipped_outputs,
//       hook_on_free); work that READS the range and is still pending keeps the block from being reused (libmem holds it
//       under the stream's tag until AresMemReleaseHeld) or is launched first (the free is fenced behind it).
//   I3  A definition never outlives what it refers to: stream destruction (hook_on_stream_destroy) drops the stream's
//       queues, journals, compactions and iotas, hands its fills to "whoever asks", settles its error words; events are
//       destroyed while their stream exists (ROCm 7: querying an event of a destroyed stream is a use after free).
//   I4  State keyed by a pointer (journals, compactions, iotas, fills) dies with the allocation: hook_on_free erases it, so
//       a recycled address never meets a stale entry.  State keyed by a stream dies with the stream (I3).
//   I5  Whatever a call decides under the lock it acts on under the SAME acquisition; a decision taken under an earlier
//       acquisition is re-validated (run_filter_rows re-checks the row-space eligibility its caller established: another
//       thread's flush may have applied the vector's pending filters in between — the round-5 soak finding).
//   I6  A flush from an unrelated call (another stream, another query) launches what is pending but leaves alone what
//       only its own consumer will ever read: limbo, consumed iotas, lazy fills, lazily defined measure rows (I1 covers
//       their readers), and compactions whose only reader is limbo work ("dormant").
// ARES_DEFER=0 switches all of it off (every call launches its own kernel); ARES_FUSE=0 the HashReduce stage.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ares_extensions.h"

#include "binding.hpp"
#include "common.hpp"
#include "device_model.hpp"
#include "dim_layout.hpp"
#include "fast_eval.hpp"
#include "hash_reduce_lds.hpp"
#include "lookback.hpp"
#include "sort_reduce_fused.hpp"

#include "transform_kernels.hpp"

namespace ares {

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// cross-call fusion: root transforms are queued per (device, stream) and launched together
// ---------------------------------------------------------------------------------------------
// Contract and flush points: include/ares_extensions.h.  A queue holds jobs that share the index
// vector and length; a job whose buffers overlap a queued job's (read-after-write or
// write-after-anything) forces the queue out first, so the fused launch never reorders dependent
// work.
// lazily defined sorts (below, after the fills): run for real / forgotten.  Caller holds the device's DeferLock.
static void materialize_sort(const uint32_t *indexVector);
static void drop_sort(const uint32_t *indexVector);
namespace {
struct ByteRange {
  const uint8_t *lo, *hi;
  bool overlaps(const ByteRange &o) const { return lo < o.hi && o.lo < hi; }
};
struct PendingQueue {
  MultiJobs jobs;
  uint32_t colRows[kMaxMultiJobs];  // rows of each job's source column
  const uint32_t *idx = nullptr;
  int n = 0;
  std::vector<ByteRange> reads, writes;
  // the host has waited for the stream since these jobs were queued (second-stage fusion,
  // include/ares_extensions.h): a late launch from outside the stream's own call order must
  // restore "the stream is idle" before anybody looks
  bool overWait = false;
  std::vector<std::shared_ptr<StreamBuffer>> keep;  // decoded copies of run-length columns the jobs read
  // (limbo only) the queue was consumed by a fused Sort + Reduce: whoever launches it replays the whole sequence
  const uint32_t *sortIdx = nullptr;
};
struct DeferState;
DeferState &state_of(int device);
// The deferral state is per device: queries on different GPUs of one process never contend, and the
// work done under a device's mutex is bookkeeping and kernel launches only (synchronisations run after
// the unlock, see below).  DeferLock selects the state its holder works on; the helpers marked
// "caller holds the device's DeferLock" reach it through t_state.
thread_local DeferState *t_state = nullptr;
// Stream synchronisations that work done under a DeferLock asks for (a queue the host has already
// waited for was launched late; a copy is about to read what a late launch writes): they run after
// the mutex is released, so one query's late launch never stalls the other streams of the device.
thread_local std::vector<hipStream_t> t_syncAfterUnlock;
struct DeferLock {
  std::unique_lock<std::mutex> lock;
  explicit DeferLock(int device);
  void unlock() {
    if (lock.owns_lock()) {
      t_state = nullptr;
      lock.unlock();
    }
    drain();
  }
  static void drain() {
    std::vector<hipStream_t> todo;
    todo.swap(t_syncAfterUnlock);
    for (size_t i = 0; i < todo.size(); i++) {
      bool seen = false;
      for (size_t j = 0; j < i; j++) seen = seen || todo[j] == todo[i];
      if (!seen) hip_check(hipStreamSynchronize(todo[i]), "hipStreamSynchronize");
    }
  }
  ~DeferLock() noexcept(false) {
    if (lock.owns_lock()) {
      t_state = nullptr;
      lock.unlock();
    }
    if (std::uncaught_exceptions() == 0) drain();
    else t_syncAfterUnlock.clear();
  }
};

// caller holds a DeferLock: something the calling entry point may read next was just launched on `producer`.  Kernels of
// the call run on the call's own stream: when that is another stream (or unknown: the call came from libmem.so), the
// producer is synchronised once the lock is released — before the caller launches anything.
void order_before_caller(hipStream_t producer) {
  if (!t_callStream.known || t_callStream.stream != producer) t_syncAfterUnlock.push_back(producer);
}

// Filters of the hot shape that have compacted an index vector since its InitIndexVector: with them
// HashReduce can re-derive the survivors from the source columns instead of reading index, dimension
// and measure vectors.  Any other writer of the index vector invalidates the entry.
struct FilterJournal {
  int device;
  hipStream_t stream;
  uint32_t start;
  int n0;
  bool valid;
  std::vector<FastOperands> filters;
  std::vector<uint32_t> colRows;
  std::vector<std::shared_ptr<StreamBuffer>> keep;  // decoded copies of run-length columns the filters read
  int lastCount = -1;  // survivors of the last journalled filter (-1: not known), what the next one must be called with
};

// Queues that a HashReduce consumed on the fly: their outputs were never written.  They stay
// launchable (their inputs are held by libmem.so) until the stream starts its next batch or the
// outputs are freed; a copy that touches an output launches them first.
void (*g_releaseHeld)(int, uintptr_t) = nullptr;  // AresMemReleaseHeld of the sibling libmem.so
bool g_fuseEnabled = true;
// Blocks are held on behalf of ONE stream's pending work: the tag names that stream, so that one
// query's progress never releases what another query's not-yet-launched kernels still read.
uintptr_t hold_tag(hipStream_t stream) { return reinterpret_cast<uintptr_t>(stream) + 1; }
struct ReleaseSet {
  std::vector<uintptr_t> tags;
  void add(hipStream_t s) { tags.push_back(hold_tag(s)); }
  void run(int device) const {
    if (g_releaseHeld)
      for (uintptr_t t : tags) g_releaseHeld(device, t);
  }
};

// A fast filter has written the predicate vector and returned the survivor count, but the compaction
// of its index vector has not run: when HashReduce re-derives the survivors from the filter journal
// nobody ever reads the compacted vector.  Whatever does read it (a second filter, transforms that
// are launched after all, a copy, any other entry point) runs the compaction first.
// One filter that has been counted in row space (filter_rows_kernel) and not applied to the index vector yet.
struct LazyFilter {
  FastOperands f;     // idx = nullptr, pad = 0: the replay fills them in
  uint32_t colRows;   // rows of the column
  uint8_t *pred;      // the predicate vector the host passed with the call
  int rowsBefore;     // length of the index vector before this filter (= the previous filter's count)
  std::shared_ptr<StreamBuffer> keep;  // the decoded copy of a run-length column the filter reads (or null)
};
// A run-length encoded column decoded for the stream's current batch (expand_runs_kernel): what it was made from, and the copy
struct RunExpansion {
  const uint8_t *base;
  uint32_t nullsOff, valuesOff, length, bitOff, step, startCount;
  int rows;
  std::shared_ptr<StreamBuffer> copy;
  size_t valuesAt;  // byte offset of the values behind the validity bitmap
};
// What the next filter call of the stream is expected to be (the previous batch of the stream had the same filter on the
// same column right behind this one): evaluated in the same pass, handed out without a kernel when the call arrives.
struct PredictedFilter {
  bool valid = false;
  FastOperands g;     // on the column of the filter it was evaluated with
  int count = 0;
  std::shared_ptr<StreamBuffer> bits;
};
struct PendingCompact {
  int device;
  hipStream_t stream;
  uint32_t *idx;
  const uint8_t *pred;
  int n, pad, tiles;
  bool virtualIdx;
  std::shared_ptr<StreamBuffer> ws;  // [total, error, ticket ...][tile counts][tile offsets + 1][loaded]
  unsigned int *ticket;
  uint32_t *error, *tileOffsets, *loaded;
  uint32_t *tileCounts = nullptr, *total = nullptr;  // not null: the scan of the tile counts has not run yet
  // row-space form (todo not empty: the fields from `pred` to `total` are unused).  The index vector holds
  // iota(0 .. n0) — virtual or written — with the first `applied` filters of the journal applied; `todo` are the filters
  // counted since, in call order; `bits` the survivors after the last of them, in row space (one 16-bit word per lane and tile).
  std::vector<LazyFilter> todo;
  int n0 = 0, applied = 0, lastCount = 0;
  std::shared_ptr<StreamBuffer> bits;
  PredictedFilter predicted;
  uint64_t generation = 0;  // of the filter that was booked last (DeferState::filterGeneration)
};

// Filters of the previous and of the current batch of a stream, by shape: what filter_rows_kernel's prediction goes by.
struct FilterShape {
  int akind, functor, I, bkind;
  uint32_t bbits, bok;
  bool sameColumnAsPrevious;
  bool same_as(const FilterShape &o) const {
    return akind == o.akind && functor == o.functor && I == o.I && bkind == o.bkind && bbits == o.bbits && bok == o.bok &&
           sameColumnAsPrevious == o.sameColumnAsPrevious;
  }
};
struct FilterHistory {
  std::vector<FilterShape> previous, current;
};

// Index vectors that InitIndexVector has defined but not written yet ("virtual iota"): the fast
// filter and transform kernels that consume them compute rows = position instead of loading 4 bytes
// per row; any other use (every flush point) materialises them first.
struct PendingIota {
  int device;
  hipStream_t stream;
  uint32_t start;
  int n;
  // A consumer has already used the vector as what it is defined to be (Reduce over zero dimensions): only somebody
  // who is handed THIS vector, copies it or frees it still cares — unrelated flush points and stream waits leave it.
  bool consumed = false;
  // Sort has been DEFINED over this vector (DeferState::sorts): what it is defined to hold is the sorted order, not the iota
  bool sorted = false;
};

// Buffers that a call has defined as "`unit`-byte pattern, repeated" but that nobody has written yet ("lazy fill"):
// the measure vector a constant transform fills (COUNT(*) is SUM over a literal 1, query/aql_compiler.go:1191-1197),
// the hash vector Sort gives a query without dimensions (every row hashes alike).  A consumer that knows what the
// bytes would be (Reduce over zero dimensions) never makes anybody write them; every other reader, copy or flush
// point writes them first; a free or an overwrite retires them.
struct PendingFill {
  int device;
  hipStream_t stream;
  size_t bytes;
  uint64_t pattern;
  int unit;  // 4 or 8
  bool streamGone = false;  // the defining stream was destroyed: written on the stream of whoever asks for it
  // not a pattern at all: a buffer a lazily defined Sort (+ Reduce) would write — the hash vector, the output's index vector.
  // Reading it runs the sort (materialize_sort), a free or a whole overwrite drops the definition (drop_sort)
  const uint32_t *sortIdx = nullptr;
};

// Sort over a dimension vector whose batch rows are still pending transforms (define_lazy_sort), and — once `reduced` —
// the Reduce that consumed it together with them (fuse_pending_into_sort_reduce)
struct PendingSort {
  int device;
  hipStream_t stream;
  DimensionVector keys;
  int length;
  bool reduced = false;
  bool fromVectors = false;  // every row of `keys` exists (define_lazy_sort_vectors): no queue, no limbo — a replay is Sort [+ Reduce]
  // what a replay of the reduced state needs
  DimensionVector outKeys;
  uint8_t *inValues = nullptr, *outValues = nullptr;
  int valueBytes = 0, aggFunc = 0, groups = 0;
  bool constMeasure = false;  // the batch's measure rows were a lazy fill the Reduce consumed: `fill` at `fillAt`
  uint8_t *fillAt = nullptr;
  PendingFill fill;
};

void hook_on_wait(int device, void *stream);
uintptr_t hook_on_free(int device, void *ptr, size_t bytes);
void hook_on_access(int device, const void *ptr, size_t bytes);
void hook_on_write(int device, const void *ptr, size_t bytes);
void hook_on_stream_destroy(int device, void *stream);
void hook_trim(int device);

// registers the flush hook with the sibling libmem.so; false = deferral is off for this process
bool defer_available() {
  static const bool ok = [] {
    const char *e = getenv("ARES_DEFER");
    if (e && e[0] == '0') return false;
    Dl_info info;
    if (!dladdr(reinterpret_cast<void *>(&AresFlushDeferred), &info) || !info.dli_fname) return false;
    std::string path(info.dli_fname);
    const size_t slash = path.rfind('/');
    path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/libmem.so";
    void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_NOLOAD);  // only if the host already uses it
    if (!h) return false;
    auto set = reinterpret_cast<void (*)(void (*)(int))>(dlsym(h, "AresMemSetFlushHook"));
    if (!set) return false;
    set(&AresFlushDeferred);
    // second stage (optional: an older libmem.so only knows the flush hook)
    const char *fuse = getenv("ARES_FUSE");
    auto setHooks = reinterpret_cast<void (*)(const AresDeferralHooks *)>(dlsym(h, "AresMemSetDeferralHooks"));
    auto release = reinterpret_cast<void (*)(int, uintptr_t)>(dlsym(h, "AresMemReleaseHeld"));
    g_fuseEnabled = !(fuse && fuse[0] == '0');
    if (setHooks && release) {  // the hooks also tell HashReduce whether its previous results are untouched
      static const AresDeferralHooks hooks = {&AresFlushDeferred, &hook_on_wait, &hook_on_free, &hook_on_access};
      g_releaseHeld = release;
      setHooks(&hooks);
    }
    auto setAux = reinterpret_cast<void (*)(const AresMemAuxHooks *)>(dlsym(h, "AresMemSetAuxHooks"));
    if (setAux) {
      static const AresMemAuxHooks aux = {sizeof(AresMemAuxHooks), &hook_on_write, &hook_on_stream_destroy, &hook_trim};
      setAux(&aux);
    }
    g_memTrimCache = reinterpret_cast<void (*)(int)>(dlsym(h, "AresMemTrimCache"));
    g_memNoteWrite = reinterpret_cast<void (*)(int, const void *, size_t)>(dlsym(h, "AresMemNoteWrite"));
    auto track = reinterpret_cast<void (*)()>(dlsym(h, "AresMemEnableWriteTracking"));
    if (g_memNoteWrite && track) track();  // from here on every kernel output is reported
    else g_memNoteWrite = nullptr;
    g_memNoteActivity = reinterpret_cast<void (*)()>(dlsym(h, "AresMemNoteActivity"));
    auto share = reinterpret_cast<void (*)()>(dlsym(h, "AresMemEnableActivityTracking"));
    if (g_memNoteActivity && share) share();  // from here on every entry point and every launch is reported: frees may share fences
    else g_memNoteActivity = nullptr;
    return true;
  }();
  return ok;
}
// second-stage fusion (HashReduce consumes pending transforms, lazy compaction): ARES_FUSE=0 switches it
// off; the hooks stay in place
bool fuse_available() { return defer_available() && g_releaseHeld != nullptr && g_fuseEnabled; }
}  // namespace
bool deferral_hooks_active() { return defer_available() && g_releaseHeld != nullptr; }
namespace {

// The error word of a lazily launched compaction (a bounded wait inside filter_compact_kernel timed
// out) cannot be read back where the launch happens — that may be inside a free or a copy.  It is
// copied to pinned memory behind the kernel and looked at by the following entry points of the
// device: the first one that finds it set reports the failure to the host.
// (The event lives no longer than its stream: hook_on_stream_destroy settles the checks of a stream that is going away.
// An event queried after its stream was destroyed is a use after free inside the ROCm 7.x runtime — see FenceEvent in
// mem/memory.hip.)
struct ErrorCheck {
  int device;
  hipStream_t stream;
  hipEvent_t done;
  uint32_t *pinned;
  std::shared_ptr<StreamBuffer> ws;  // keeps the error word alive until the copy has run
};

struct DeferState {
  std::mutex mutex;
  std::map<std::pair<int, hipStream_t>, PendingQueue> pending;  // root transforms queued per (device, stream)
  std::map<std::pair<int, hipStream_t>, PendingQueue> limbo;    // queues a HashReduce consumed on the fly
  std::map<const uint32_t *, FilterJournal> journals;
  std::map<const uint32_t *, PendingCompact> compactions;
  std::map<std::pair<int, hipStream_t>, FilterHistory> filterHistory;
  uint64_t filterGeneration = 0;
  std::map<uint32_t *, PendingIota> iotas;
  std::map<uint8_t *, PendingFill> fills;  // by first byte; ranges never overlap
  std::map<const uint32_t *, PendingSort> sorts;  // by index vector
  std::map<std::pair<int, hipStream_t>, std::vector<RunExpansion>> expansions;  // decoded run-length columns of the stream's batch
  std::vector<ErrorCheck> errorChecks;
  std::vector<uint32_t *> errorSlots;  // recycled pinned words
  bool errorSeen = false;              // a check that was settled outside an entry point failed: the next poll reports it
};
constexpr int kMaxDevices = 64;
DeferState &state_of(int device) {
  static DeferState states[kMaxDevices];
  if (device < 0 || device >= kMaxDevices) throw std::invalid_argument("device index out of range");
  return states[device];
}
DeferLock::DeferLock(int device) : lock(state_of(device).mutex, std::defer_lock) {
  {
    SlowScope slow("wait for the device's deferral lock");
    lock.lock();
  }
  t_state = &state_of(device);
}

// caller holds the device's DeferLock
void watch_error_word(int device, hipStream_t stream, const uint32_t *errorDev, std::shared_ptr<StreamBuffer> ws) {
  ErrorCheck c;
  c.device = device;
  c.stream = stream;
  c.ws = std::move(ws);
  if (!t_state->errorSlots.empty()) {
    c.pinned = t_state->errorSlots.back();
    t_state->errorSlots.pop_back();
  } else {
    hip_check(hipHostMalloc(reinterpret_cast<void **>(&c.pinned), sizeof(uint32_t), hipHostMallocPortable), "hipHostMalloc");
  }
  hip_check(hipEventCreateWithFlags(&c.done, hipEventDisableTiming), "hipEventCreate");
  *c.pinned = 0;
  hip_check(hipMemcpyAsync(c.pinned, errorDev, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
  hip_check(hipEventRecord(c.done, stream), "hipEventRecord");
  t_state->errorChecks.push_back(std::move(c));
}

// caller holds the device's DeferLock; throws when a finished lazy compaction reported a failure
void poll_error_words(int device) {
  bool failed = t_state->errorSeen;
  t_state->errorSeen = false;
  for (size_t i = 0; i < t_state->errorChecks.size();) {
    ErrorCheck &c = t_state->errorChecks[i];
    if (c.device == device && hipEventQuery(c.done) == hipSuccess) {
      failed = failed || *c.pinned != 0;
      (void)hipEventDestroy(c.done);
      t_state->errorSlots.push_back(c.pinned);
      t_state->errorChecks[i] = std::move(t_state->errorChecks.back());
      t_state->errorChecks.pop_back();
    } else {
      (void)hipGetLastError();
      i++;
    }
  }
  if (failed) throw AlgorithmError("ERROR: filter: compaction wait timed out (reported by a deferred compaction)");
}

// caller holds the device's DeferLock and has selected the device: runs the pending compaction of `idx` (if
// there is one) on the stream its filter ran on
void replay_lazy_filters(const PendingCompact &c);
void run_compaction(const uint32_t *idx) {
  auto it = t_state->compactions.find(idx);
  if (it == t_state->compactions.end()) return;
  const PendingCompact c = it->second;
  t_state->compactions.erase(it);
  if (!c.todo.empty()) {
    replay_lazy_filters(c);
    return;
  }
  mem_note_write(c.device, c.idx, 4ull * static_cast<size_t>(c.n));
  CompactWorkspace cw;
  cw.ticket = c.ticket;
  cw.error = c.error;
  cw.tileOffsets = c.tileOffsets;
  cw.loaded = c.loaded;
  const int cgrid = capped_grid((c.tiles + kTilesPerTicket - 1) / kTilesPerTicket, 256 * 8);
  if (c.tileCounts)
    ARES_LAUNCH("filter_scan_kernel", filter_scan_kernel, 1, 1024, c.stream, c.tileCounts, c.tileOffsets, c.tiles, c.total);
  if (c.virtualIdx)
    ARES_LAUNCH("filter_compact_kernel<iota>", (filter_compact_kernel<uint32_t, true>), cgrid, kBlock, c.stream, c.pred, c.idx, 0u,
                c.pad, cw, c.n, c.tiles);
  else
    ARES_LAUNCH("filter_compact_kernel", (filter_compact_kernel<uint32_t, false>), cgrid, kBlock, c.stream, c.pred, c.idx, 0u,
                c.pad, cw, c.n, c.tiles);
  watch_error_word(c.device, c.stream, c.error, c.ws);
  order_before_caller(c.stream);
  // (c.ws is released to the stream's cache when the last copy of the shared_ptr goes: behind the launch
  // and the copy of the error word)
}
// Filters that were only counted (filter_rows_kernel) are applied to the index vector after all: each one as the call
// would have run eagerly — predicate bytes over the vector as the previous filter left it, tile counts, offsets,
// in-place compaction — so that predicate and index vectors hold exactly what the reference leaves in them.
// caller holds the device's DeferLock and has selected the device
void replay_lazy_filters(const PendingCompact &c) {
  bool virtualIdx = c.virtualIdx;
  mem_note_write(c.device, c.idx, 4ull * static_cast<size_t>(c.n0));
  for (const LazyFilter &L : c.todo) {
    const int n = L.rowsBefore;
    if (n <= 0) break;
    mem_note_write(c.device, L.pred, static_cast<size_t>(n));
    FastOperands f = L.f;
    f.idx = virtualIdx ? nullptr : c.idx;
    f.pad = static_cast<int>(reinterpret_cast<uintptr_t>(L.pred) & 3);
    const int64_t numQuads = (static_cast<int64_t>(n) + f.pad + 3) / 4;
    const int tiles = static_cast<int>((numQuads + kBlock * kPQ - 1) / (kBlock * kPQ));
    const size_t head = 64;
    const size_t words = static_cast<size_t>(tiles) * 3 + 1;  // [tile counts][tile offsets + 1][loaded]
    auto wsBuf = std::make_shared<StreamBuffer>(head + 4 * words, c.stream);
    uint32_t *w = wsBuf->as<uint32_t>();
    uint32_t *total = w, *error = w + 1;
    uint32_t *tileCounts = w + 16, *tileOffsets = tileCounts + tiles, *loaded = tileOffsets + tiles + 1;
    hip_check(hipMemsetAsync(w, 0, head + 4 * words, c.stream), "hipMemsetAsync");
    ARES_LAUNCH("filter_pred_kernel", filter_pred_kernel, capped_grid(tiles, 256 * 16), kBlock, c.stream, f, L.pred, tileCounts, n, tiles,
                static_cast<uint32_t *>(nullptr));
    ARES_LAUNCH("filter_scan_kernel", filter_scan_kernel, 1, 1024, c.stream, tileCounts, tileOffsets, tiles, total);
    CompactWorkspace cw;
    cw.ticket = w + 2;
    cw.error = error;
    cw.tileOffsets = tileOffsets;
    cw.loaded = loaded;
    const int cgrid = capped_grid((tiles + kTilesPerTicket - 1) / kTilesPerTicket, 256 * 8);
    if (virtualIdx)
      ARES_LAUNCH("filter_compact_kernel<iota>", (filter_compact_kernel<uint32_t, true>), cgrid, kBlock, c.stream, L.pred, c.idx, 0u, f.pad,
                  cw, n, tiles);
    else
      ARES_LAUNCH("filter_compact_kernel", (filter_compact_kernel<uint32_t, false>), cgrid, kBlock, c.stream, L.pred, c.idx, 0u, f.pad, cw,
                  n, tiles);
    watch_error_word(c.device, c.stream, error, wsBuf);
    virtualIdx = false;
  }
  order_before_caller(c.stream);
}

bool compaction_touches(const PendingCompact &c, const ByteRange &r) {
  if (!c.todo.empty()) {  // index vector, every predicate vector and every column a replay would read
    const ByteRange ri{reinterpret_cast<const uint8_t *>(c.idx), reinterpret_cast<const uint8_t *>(c.idx) + 4ull * c.n0};
    bool hit = ri.overlaps(r);
    for (const LazyFilter &L : c.todo) {
      const ByteRange rp{L.pred, L.pred + L.rowsBefore};
      const ByteRange rv{reinterpret_cast<const uint8_t *>(L.f.vals), reinterpret_cast<const uint8_t *>(L.f.vals) + fast_value_bytes(L.f, L.colRows)};
      hit = hit || rp.overlaps(r) || rv.overlaps(r);
      if (L.f.nulls) {
        const ByteRange rn{L.f.nulls, L.f.nulls + (static_cast<uint64_t>(L.colRows) + L.f.bitOff + 7) / 8 + 2};
        hit = hit || rn.overlaps(r);
      }
    }
    return hit;
  }
  const ByteRange ri{reinterpret_cast<const uint8_t *>(c.idx), reinterpret_cast<const uint8_t *>(c.idx) + 4ull * c.n};
  const ByteRange rp{c.pred, c.pred + c.n};
  return ri.overlaps(r) || rp.overlaps(r);
}

// ARES_FILTER_CHECK=<log file> (race hunting, see FilterCheck below): the bare-column jobs of a queue that was just
// launched are checked against a copy of their column taken right behind the kernel; a mismatch is looked at again a
// millisecond later — did the kernel read something that is no longer there (a writer racing with it) or is the column
// itself not what the host uploaded?  Synchronises the queue's stream only.
void transform_self_check(hipStream_t stream, const PendingQueue &q) {
  static const char *path = getenv("ARES_FILTER_CHECK");
  if (!path || !path[0] || q.n <= 0 || q.n > (1 << 20)) return;
  const size_t n = static_cast<size_t>(q.n);
  std::vector<uint32_t> idx(q.idx ? n : 0), out(n), vals, vals2;
  std::vector<uint8_t> outOk(n), nulls, nulls2;
  if (hipStreamSynchronize(stream) != hipSuccess) return;
  if (q.idx && hipMemcpyAsync(idx.data(), q.idx, 4 * n, hipMemcpyDeviceToHost, stream) != hipSuccess) return;
  for (int j = 0; j < q.jobs.count; j++) {
    const FastOperands &f = q.jobs.f[j];
    const SinkD &sk = q.jobs.s[j];
    if (f.arity != 1 || sk.type == SINK_MEASURE || sk.width != 4 || f.step != 4 || !sk.nulls || f.akind == K_F32 || f.I == K_F32) continue;
    const size_t rows = q.colRows[j], nb = f.nulls ? (rows + f.bitOff + 7) / 8 : 0;
    vals.resize(rows); vals2.resize(rows); nulls.resize(nb); nulls2.resize(nb);
    (void)hipMemcpyAsync(vals.data(), f.vals, 4 * rows, hipMemcpyDeviceToHost, stream);
    if (nb) (void)hipMemcpyAsync(nulls.data(), f.nulls, nb, hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(out.data(), sk.values, 4 * n, hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(outOk.data(), sk.nulls, n, hipMemcpyDeviceToHost, stream);
    if (hipStreamSynchronize(stream) != hipSuccess) return;
    long bad = 0, firstBad = -1;
    for (size_t i = 0; i < n; i++) {
      const uint32_t row = q.idx ? idx[i] : static_cast<uint32_t>(i);
      if (row >= rows) continue;
      const uint32_t ok = nb ? (nulls[(row + f.bitOff) >> 3] >> ((row + f.bitOff) & 7)) & 1u : 1u;
      if (out[i] != vals[row] || (outOk[i] != 0) != (ok != 0)) {
        if (firstBad < 0) firstBad = static_cast<long>(i);
        bad++;
      }
    }
    if (!bad) continue;
    usleep(1000);
    (void)hipMemcpyAsync(vals2.data(), f.vals, 4 * rows, hipMemcpyDeviceToHost, stream);
    if (nb) (void)hipMemcpyAsync(nulls2.data(), f.nulls, nb, hipMemcpyDeviceToHost, stream);
    (void)hipStreamSynchronize(stream);
    long colChanged = 0;
    for (size_t r = 0; r < rows; r++) colChanged += vals[r] != vals2[r];
    for (size_t b = 0; b < nb; b++) colChanged += nulls[b] != nulls2[b];
    static std::mutex logMutex;
    std::lock_guard<std::mutex> lock(logMutex);
    if (FILE *o = fopen(path, "a")) {
      const uint32_t row = q.idx ? idx[firstBad] : static_cast<uint32_t>(firstBad);
      fprintf(o, "TRANSFORMCHECK MISMATCH job %d/%d n %d colRows %zu idx %p vals %p nulls %p out %p stream %p: %ld positions differ, first %ld "
                 "(row %u: out %u ok %u, column %u then %u); column bytes that changed within 1 ms: %ld\n",
              j, q.jobs.count, q.n, rows, (const void *)q.idx, (const void *)f.vals, (const void *)f.nulls, (const void *)sk.values, (void *)stream, bad,
              firstBad, row, out[firstBad], outOk[firstBad], vals[row], vals2[row], colChanged);
      fclose(o);
    }
  }
}

// caller holds the device's DeferLock and has selected the device.  inOrder: the launch is part of the
// stream's own call sequence (a later call on the same stream follows); otherwise a queue the host
// has already waited for is synchronised after its late launch.
void launch_queue(hipStream_t stream, PendingQueue &q, bool inOrder = false) {
  if (q.jobs.count == 0) return;
  const bool syncAfter = q.overWait && !inOrder;
  q.overWait = false;
  if (q.idx) run_compaction(q.idx);  // the transforms read the compacted index vector
  const int64_t numQuads = (static_cast<int64_t>(q.n) + 3) / 4;
  const int64_t tiles = (numQuads + kBlock * kTQ - 1) / (kBlock * kTQ);
  if (q.jobs.count == 1) {
    FastOperands f = q.jobs.f[0];
    f.idx = q.idx;
    f.pad = q.jobs.s[0].type == SINK_MEASURE ? 0 : static_cast<int>(reinterpret_cast<uintptr_t>(q.jobs.s[0].nulls) & 3);
    const int64_t nq = (static_cast<int64_t>(q.n) + f.pad + 3) / 4;
    const int64_t t1 = (nq + kBlock * kTQ - 1) / (kBlock * kTQ);
    ARES_LAUNCH("transform_fast_kernel", transform_fast_kernel, capped_grid(t1, 256 * 16), kBlock, stream, f, q.jobs.s[0], q.n, nq);
  } else {
    ARES_LAUNCH("transform_multi_kernel", transform_multi_kernel, capped_grid(tiles, 256 * 16), kBlock, stream, q.jobs, q.idx,
                q.n, numQuads);
  }
  if (g_memNoteWrite) {
    const int device = current_device();
    for (const ByteRange &w : q.writes) mem_note_write(device, w.lo, static_cast<size_t>(w.hi - w.lo));
  }
  transform_self_check(stream, q);
  q.jobs.count = 0;
  q.reads.clear();
  q.writes.clear();
  q.keep.clear();  // (released in stream order, behind the kernel that read them)
  if (syncAfter) t_syncAfterUnlock.push_back(stream);  // (DeferLock: after the mutex is released)
}

// caller holds the device's DeferLock.  Launches the skipped transforms of every limbo entry of the device
// (all = true) or of those whose outputs overlap `range`, and forgets the entries.
bool materialize_limbo(int device, const ByteRange *range, ReleaseSet *released) {
  bool any = false;
  for (auto it = t_state->limbo.begin(); it != t_state->limbo.end();) {
    bool hit = it->first.first == device;
    if (hit && range) {
      hit = false;
      for (const ByteRange &w : it->second.writes) hit = hit || w.overlaps(*range);
    }
    if (hit && it->second.sortIdx) {  // consumed by a fused Sort + Reduce: the whole sequence is replayed (the entry goes with it)
      materialize_sort(it->second.sortIdx);
      it = t_state->limbo.begin();
      any = true;
    } else if (hit) {
      it->second.overWait = true;  // the host believes this work is long done
      launch_queue(it->first.second, it->second);
      if (released) released->add(it->first.second);
      it = t_state->limbo.erase(it);
      any = true;
    } else {
      ++it;
    }
  }
  return any;
}
}  // namespace

// caller holds the device's DeferLock and has selected the device
static void launch_init_index(uint32_t *indexVector, uint32_t start, int n, hipStream_t stream) {
  if (n > 0) mem_note_write(current_device(), indexVector, 4ull * static_cast<size_t>(n));
  if (n <= 0) return;
  const int64_t quads = (static_cast<int64_t>(n) + 3) / 4;
  const int grid = capped_grid((quads + kBlock - 1) / kBlock, 256 * 16);
  ARES_LAUNCH("init_index_kernel", init_index_kernel, grid, kBlock, stream, indexVector, start, n);
  order_before_caller(stream);  // (a lazy iota written at another stream's flush point)
}

// ---- lazy fills -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void fill_pattern_kernel(uint8_t *dst, size_t units, uint64_t pattern, int unit) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < units; i += static_cast<size_t>(gridDim.x) * kBlock) {
    if (unit == 8) reinterpret_cast<uint64_t *>(dst)[i] = pattern;
    else reinterpret_cast<uint32_t *>(dst)[i] = static_cast<uint32_t>(pattern);
  }
}

// caller holds the device's DeferLock and has selected the device
static void launch_fill(uint8_t *dst, const PendingFill &f) {
  if (f.bytes == 0 || f.sortIdx) return;  // (a sort's marker is not a pattern: materialize_sort)
  mem_note_write(f.device, dst, f.bytes);
  const size_t units = f.bytes / static_cast<size_t>(f.unit);
  // a fill whose defining stream is gone is written on the caller's stream (or, from a libmem.so hook, on the null stream)
  const hipStream_t stream = f.streamGone ? (t_callStream.known ? t_callStream.stream : nullptr) : f.stream;
  ARES_LAUNCH("fill_pattern_kernel", fill_pattern_kernel, capped_grid(static_cast<int64_t>((units + kBlock - 1) / kBlock), 256 * 8), kBlock,
              stream, dst, units, f.pattern, f.unit);
  order_before_caller(stream);  // whoever made us write it reads (or overwrites) it next, maybe on another stream
}

// caller holds the device's DeferLock: every lazy fill of the device (r == nullptr) or those that overlap r are
// written now
static void materialize_fills(int device, const ByteRange *r, std::vector<hipStream_t> *touched = nullptr) {
  for (auto it = t_state->fills.begin(); it != t_state->fills.end();) {
    const ByteRange v{it->first, it->first + it->second.bytes};
    if (it->second.device == device && it->second.sortIdx) {
      // a buffer a lazily defined Sort (+ Reduce) would write: only for somebody who looks at these very bytes
      if (r && v.overlaps(*r)) {
        if (touched) touched->push_back(it->second.stream);
        materialize_sort(it->second.sortIdx);  // (erases the marker)
        it = t_state->fills.begin();
      } else {
        ++it;
      }
    } else if (it->second.device == device && (!r || v.overlaps(*r))) {
      launch_fill(it->first, it->second);
      if (touched) touched->push_back(it->second.stream);
      it = t_state->fills.erase(it);
    } else {
      ++it;
    }
  }
}

// caller holds the device's DeferLock: [r.lo, r.hi) is freed (`gone`) or about to be overwritten.  Lazy fills inside
// it die; one that sticks out at an end is shortened (on a pattern boundary); one that is cut in the middle, or off
// a pattern boundary, is written first.
static void retire_fills(int device, const ByteRange &r, bool gone) {
  for (auto it = t_state->fills.begin(); it != t_state->fills.end();) {
    uint8_t *lo = it->first;
    PendingFill f = it->second;
    const uint8_t *hi = lo + f.bytes;
    if (f.device != device || !(lo < r.hi && r.lo < hi)) {
      ++it;
      continue;
    }
    if (f.sortIdx) {  // freed, or overwritten whole: the definition dies; cut: what the sort would leave is written first
      if (gone || (r.lo <= lo && hi <= r.hi)) drop_sort(f.sortIdx);
      else materialize_sort(f.sortIdx);
      it = t_state->fills.begin();
      continue;
    }
    it = t_state->fills.erase(it);
    if (gone || (r.lo <= lo && hi <= r.hi)) continue;
    const bool headCut = r.lo <= lo, tailCut = hi <= r.hi;  // (not both: handled above)
    if (headCut && (static_cast<size_t>(r.hi - lo) % static_cast<size_t>(f.unit)) == 0) {
      f.bytes = static_cast<size_t>(hi - r.hi);
      t_state->fills[const_cast<uint8_t *>(r.hi)] = f;
      it = t_state->fills.upper_bound(const_cast<uint8_t *>(r.hi));
    } else if (tailCut && (static_cast<size_t>(r.lo - lo) % static_cast<size_t>(f.unit)) == 0) {
      f.bytes = static_cast<size_t>(r.lo - lo);
      t_state->fills[lo] = f;
      it = t_state->fills.upper_bound(lo);
    } else {
      launch_fill(lo, f);
      it = t_state->fills.upper_bound(lo);
    }
  }
}

// ---- lazily defined sorts ---------------------------------------------------------------------------------------------
// every buffer a lazily defined Sort (+ Reduce) reads or would write
static bool sort_touches(const PendingSort &s, const ByteRange &r) {
  auto hit = [&](const void *p, size_t bytes) {
    const uint8_t *lo = static_cast<const uint8_t *>(p);
    return p && bytes && lo < r.hi && r.lo < lo + bytes;
  };
  auto vector_bytes = [](const DimensionVector &v) {
    size_t rowBytes = 0;
    for (int w = 0; w < NUM_DIM_WIDTH; w++) rowBytes += static_cast<size_t>(v.NumDimsPerDimWidth[w]) * ((1u << (NUM_DIM_WIDTH - 1 - w)) + 1);
    return rowBytes * static_cast<size_t>(v.VectorCapacity > 0 ? v.VectorCapacity : 0);
  };
  const size_t n = static_cast<size_t>(s.length > 0 ? s.length : 0);
  bool any = hit(s.keys.DimValues, vector_bytes(s.keys)) || hit(s.keys.HashValues, 8 * n) || hit(s.keys.IndexVector, 4 * n);
  if (s.reduced)
    any = any || hit(s.outKeys.DimValues, vector_bytes(s.outKeys)) || hit(s.outKeys.IndexVector, 4 * n) ||
          hit(s.inValues, static_cast<size_t>(s.valueBytes) * n) || hit(s.outValues, static_cast<size_t>(s.valueBytes) * n);
  return any;
}

// caller holds the device's DeferLock.  The definition is forgotten: its buffers are freed, redefined or overwritten whole,
// or the work a replay needs is gone.  The marker entries go with it; the index vector's iota entry too (nobody may take
// the vector for an iota any more).
static void drop_sort(const uint32_t *indexVector) {
  auto it = t_state->sorts.find(indexVector);
  if (it == t_state->sorts.end()) return;
  t_state->sorts.erase(it);
  auto io = t_state->iotas.find(const_cast<uint32_t *>(indexVector));
  if (io != t_state->iotas.end() && io->second.sorted) t_state->iotas.erase(io);
  for (auto f = t_state->fills.begin(); f != t_state->fills.end();) f = (f->second.sortIdx == indexVector) ? t_state->fills.erase(f) : std::next(f);
  for (auto &kv : t_state->limbo)
    if (kv.second.sortIdx == indexVector) kv.second.sortIdx = nullptr;  // (plain skipped work from now on)
}

// caller holds the device's DeferLock and has selected the device.  Somebody reads (or partly overwrites) what a lazily
// defined Sort — and, once `reduced`, the Reduce that consumed it — would have written: the sequence runs for real, on the
// stream it was defined on: the batch's transforms (pending, or in limbo), the constant measure rows the Reduce consumed,
// InitIndexVector, Sort, Reduce.  Integer aggregates only are ever consumed, so the replayed Reduce rewrites the output's
// dimension and measure rows with the bytes they already hold.  (Rare: the Go host never looks at these vectors.  The sort
// synchronises its stream while the lock is held.)
static void materialize_sort(const uint32_t *indexVector) {
  auto it = t_state->sorts.find(indexVector);
  if (it == t_state->sorts.end()) return;
  const PendingSort s = it->second;
  drop_sort(indexVector);  // (the entries go first: nothing below finds them again)
  ReleaseSet released;
  if (s.reduced && s.fromVectors) {
    // (the rows exist: nothing to launch first)
  } else if (s.reduced) {
    auto lim = t_state->limbo.find({s.device, s.stream});
    if (lim != t_state->limbo.end()) {
      lim->second.overWait = true;  // the host believes this work is long done
      launch_queue(s.stream, lim->second);
      released.add(s.stream);
      t_state->limbo.erase(lim);
    }
    if (s.constMeasure) {
      PendingFill f = s.fill;
      f.sortIdx = nullptr;
      launch_fill(s.fillAt, f);
    }
  } else {
    auto pq = t_state->pending.find({s.device, s.stream});
    if (pq != t_state->pending.end() && pq->second.jobs.count) {
      if (pq->second.overWait) released.add(s.stream);
      launch_queue(s.stream, pq->second);
    }
  }
  launch_init_index(const_cast<uint32_t *>(indexVector), 0, s.length, s.stream);
  sort_keys_now(s.keys, s.length, s.stream);
  if (s.reduced) (void)reduce_now(s.keys, s.inValues, s.outKeys, s.outValues, s.valueBytes, s.length, s.aggFunc, s.stream);
  order_before_caller(s.stream);
  released.run(s.device);
}

// [dst, dst + bytes) is defined as `pattern` (unit = 4 or 8 bytes) repeated; false = deferral is off, the caller writes
bool defer_fill(int device, hipStream_t stream, void *dst, size_t bytes, uint64_t pattern, int unit) {
  if (!defer_available() || bytes == 0 || (unit != 4 && unit != 8) || bytes % static_cast<size_t>(unit) ||
      reinterpret_cast<uintptr_t>(dst) % static_cast<uintptr_t>(unit))
    return false;
  drop_skipped_outputs(device, dst, bytes, nullptr, 0);  // what an earlier HashReduce skipped and would write there is dead
  DeferLock lock(device);
  uint8_t *p = static_cast<uint8_t *>(dst);
  const ByteRange r{p, p + bytes};
  for (auto &kv : t_state->pending) {  // queued transforms that write into the range come first (call order)
    bool hit = false;
    if (kv.first.first == device && kv.second.jobs.count)
      for (const ByteRange &w : kv.second.writes) hit = hit || w.overlaps(r);
    if (hit) launch_queue(kv.first.second, kv.second);
  }
  retire_fills(device, r, false);
  t_state->fills[p] = PendingFill{device, stream, bytes, pattern, unit, false};
  return true;
}

// a lazy fill that covers exactly [dst, dst + bytes)
bool pending_fill_exact(int device, const void *dst, size_t bytes, uint64_t *pattern, int *unit) {
  if (!defer_available()) return false;
  DeferLock lock(device);
  auto it = t_state->fills.find(const_cast<uint8_t *>(static_cast<const uint8_t *>(dst)));
  if (it == t_state->fills.end() || it->second.device != device || it->second.bytes != bytes || it->second.sortIdx) return false;
  if (pattern) *pattern = it->second.pattern;
  if (unit) *unit = it->second.unit;
  return true;
}

// a lazy fill that covers rows [prev, length) of a vector of `width`-byte values for some prev: *prev and *pattern
bool pending_fill_tail(int device, const void *base, int width, int length, int *prev, uint64_t *pattern) {
  if (!defer_available() || length <= 0) return false;
  DeferLock lock(device);
  const uint8_t *b = static_cast<const uint8_t *>(base), *end = b + static_cast<size_t>(width) * length;
  auto it = t_state->fills.lower_bound(const_cast<uint8_t *>(b));
  if (it == t_state->fills.end() || it->second.device != device || it->second.unit != width || it->second.sortIdx) return false;
  if (it->first >= end || it->first + it->second.bytes != end || static_cast<size_t>(it->first - b) % static_cast<size_t>(width)) return false;
  *prev = static_cast<int>(static_cast<size_t>(it->first - b) / static_cast<size_t>(width));
  *pattern = it->second.pattern;
  return true;
}

// [ptr, ptr + bytes) is about to be overwritten by a kernel of the calling entry point
void retire_fills_for_write(int device, const void *ptr, size_t bytes) {
  if (!ptr || bytes == 0 || !defer_available()) return;
  DeferLock lock(device);
  if (t_state->fills.empty()) return;
  const uint8_t *p = static_cast<const uint8_t *>(ptr);
  retire_fills(device, ByteRange{p, p + bytes}, false);
}

// [ptr, ptr + bytes) is about to be read by a kernel of the calling entry point
void materialize_fills_for_read(int device, const void *ptr, size_t bytes) {
  if (!ptr || bytes == 0 || !defer_available()) return;
  DeferLock lock(device);
  if (t_state->fills.empty()) return;
  const uint8_t *p = static_cast<const uint8_t *>(ptr);
  const ByteRange r{p, p + bytes};
  materialize_fills(device, &r);
}

// true when `indexVector` is defined as iota(0 .. n) and not written yet (InitIndexVector is lazy); consume: the
// caller has used it as such — from now on only whoever is handed this very vector makes anybody write it
bool virtual_iota_peek(int device, const uint32_t *indexVector, int n, bool consume) {
  if (!defer_available()) return false;
  DeferLock lock(device);
  auto it = t_state->iotas.find(const_cast<uint32_t *>(indexVector));
  if (it == t_state->iotas.end() || it->second.device != device || it->second.start != 0 || it->second.n != n || it->second.sorted) return false;
  if (consume) it->second.consumed = true;
  return true;
}

// an entry point reads (or rewrites) the buffers of a dimension vector with kernels: lazy fills inside them are written
// first, and so is a lazy iota that a consumer has left behind
void settle_dimension_vector(int device, const DimensionVector &v, bool rowsOnly) {
  if (!defer_available()) return;
  const size_t cap = v.VectorCapacity > 0 ? static_cast<size_t>(v.VectorCapacity) : 0;
  size_t rowBytes = 0;
  for (int w = 0; w < NUM_DIM_WIDTH; w++) rowBytes += static_cast<size_t>(v.NumDimsPerDimWidth[w]) * ((1u << (NUM_DIM_WIDTH - 1 - w)) + 1);
  materialize_fills_for_read(device, v.DimValues, rowBytes * cap);
  if (rowsOnly) return;  // (the caller is about to DEFINE the hash and index vector: define_lazy_sort_vectors)
  materialize_fills_for_read(device, v.HashValues, 8 * cap);
  materialize_fills_for_read(device, v.IndexVector, 4 * cap);
  materialize_index_vector(device, v.IndexVector);
}

// an entry point is handed `indexVector` and will read it with a kernel: a lazy iota that a consumer has left behind
// is written now (the unconsumed ones are written by the entry point's flush)
void materialize_index_vector(int device, const uint32_t *indexVector) {
  if (!indexVector || !defer_available()) return;
  DeferLock lock(device);
  auto it = t_state->iotas.find(const_cast<uint32_t *>(indexVector));
  if (it == t_state->iotas.end() || it->second.device != device || !it->second.consumed) return;
  if (it->second.sorted) {  // defined as what Sort leaves: run it
    materialize_sort(indexVector);
    return;
  }
  launch_init_index(it->first, it->second.start, it->second.n, it->second.stream);
  t_state->iotas.erase(it);
}

// limboA/limboB: when given, only the skipped work whose outputs overlap these byte ranges is
// launched (the caller reads nothing else); otherwise all of it
// exempt: the index vector whose pending filters the caller is about to extend (a filter call of the hot shape): its
// compaction stays pending
static void flush_deferred_impl(int device, const ByteRange *limboA, const ByteRange *limboB, const uint32_t *exempt = nullptr) {
  // (before the deferral lock: measure rows a table image defines are written by hash_reduce_lds.hip under its own lock)
  if (limboA) grouped_materialize_for_read(device, limboA->lo, static_cast<size_t>(limboA->hi - limboA->lo));
  if (limboB) grouped_materialize_for_read(device, limboB->lo, static_cast<size_t>(limboB->hi - limboB->lo));
  DeferLock lock(device);
  poll_error_words(device);
  // lazy fills (like work a HashReduce skipped, below) are only written for byte ranges the caller reads: they live in
  // measure / hash vectors, which only Sort, Reduce, HashReduce, Expand, HyperLogLog and copies look at
  if (limboA) materialize_fills(device, limboA);
  if (limboB) materialize_fills(device, limboB);
  for (auto it = t_state->iotas.begin(); it != t_state->iotas.end();) {
    const ByteRange v{reinterpret_cast<const uint8_t *>(it->first), reinterpret_cast<const uint8_t *>(it->first) + 4ull * it->second.n};
    const bool wanted = !it->second.consumed || (limboA && v.overlaps(*limboA)) || (limboB && v.overlaps(*limboB));
    if (it->second.device == device && wanted && it->second.sorted) {
      materialize_sort(it->first);
      it = t_state->iotas.begin();
    } else if (it->second.device == device && wanted) {
      launch_init_index(it->first, it->second.start, it->second.n, it->second.stream);
      it = t_state->iotas.erase(it);
    } else {
      ++it;
    }
  }
  // pending compactions: whoever comes next may read the index vector — except where only work that a
  // HashReduce skipped would read it (the previous batch of the query's other stream): that stays
  // dormant with its skipped work and dies with it
  auto dormant = [&](const uint32_t *idx) {
    bool skipped = false, queued = false;
    for (auto &kv : t_state->limbo) skipped = skipped || kv.second.idx == idx;
    for (auto &kv : t_state->pending) queued = queued || (kv.second.jobs.count && kv.second.idx == idx);
    return skipped && !queued;
  };
  for (;;) {
    auto c = t_state->compactions.begin();
    while (c != t_state->compactions.end() && (c->second.device != device || c->first == exempt || dormant(c->first))) ++c;
    if (c == t_state->compactions.end()) break;
    run_compaction(c->first);
  }
  ReleaseSet released;
  for (auto &kv : t_state->pending)
    if (kv.first.first == device) {
      if (kv.second.overWait && kv.second.jobs.count) released.add(kv.first.second);
      launch_queue(kv.first.second, kv.second);
    }
  // Work that a HashReduce skipped (limbo) is only launched for byte ranges the caller reads: a full
  // flush from an unrelated call (the next batch's first filter on the query's other stream, say)
  // leaves it alone.
  if (limboA) materialize_limbo(device, limboA, &released);
  if (limboB) materialize_limbo(device, limboB, &released);
  released.run(device);
}

void flush_deferred(int device) { flush_deferred_impl(device, nullptr, nullptr); }

void flush_deferred_for_vector(int device, const DimensionVector &v, const void *values, size_t valueBytes) {
  size_t rowBytes = 0;
  for (int w = 0; w < NUM_DIM_WIDTH; w++) rowBytes += static_cast<size_t>(v.NumDimsPerDimWidth[w]) * ((1u << (NUM_DIM_WIDTH - 1 - w)) + 1);
  flush_deferred_for_inputs(device, v.DimValues, rowBytes * static_cast<size_t>(v.VectorCapacity > 0 ? v.VectorCapacity : 0),
                            values, valueBytes);
}

// A HashReduce (or any other writer) is about to overwrite [a, a + aBytes) and [b, b + bBytes): work
// that an earlier HashReduce skipped and whose outputs lie in there must never be launched any more —
// it would overwrite the new results (the Go host ping-pongs its result buffers and alternates two
// streams: batch k's skipped transforms target the buffer batch k+1 reduces into).
void drop_skipped_outputs(int device, const void *a, size_t aBytes, const void *b, size_t bBytes) {
  if (!fuse_available()) return;
  const ByteRange ra{static_cast<const uint8_t *>(a), static_cast<const uint8_t *>(a) + (aBytes ? aBytes : 1)};
  const ByteRange rb{static_cast<const uint8_t *>(b), static_cast<const uint8_t *>(b) + (bBytes ? bBytes : 1)};
  ReleaseSet released;
  {
    DeferLock lock(device);
    for (auto it = t_state->limbo.begin(); it != t_state->limbo.end();) {
      bool hit = false;
      if (it->first.first == device)
        for (const ByteRange &w : it->second.writes) hit = hit || w.overlaps(ra) || w.overlaps(rb);
      if (hit) {
        if (it->second.idx) t_state->compactions.erase(it->second.idx);
        released.add(it->first.second);
        const uint32_t *sortIdx = it->second.sortIdx;
        it = t_state->limbo.erase(it);
        if (sortIdx) drop_sort(sortIdx);  // (what its Reduce left defined cannot be replayed any more)
      } else {
        ++it;
      }
    }
  }
  released.run(device);
}

// for an entry point that reads exactly [a, a + aBytes) and [b, b + bBytes) of device memory
void flush_deferred_for_inputs(int device, const void *a, size_t aBytes, const void *b, size_t bBytes) {
  const ByteRange ra{static_cast<const uint8_t *>(a), static_cast<const uint8_t *>(a) + (aBytes ? aBytes : 1)};
  const ByteRange rb{static_cast<const uint8_t *>(b), static_cast<const uint8_t *>(b) + (bBytes ? bBytes : 1)};
  flush_deferred_impl(device, &ra, &rb);
}

// InitIndexVector opens a batch on the stream: what the previous batch's HashReduce skipped is dead
// now (the host has swapped its result buffers), and the new index vector starts a filter journal.
static void begin_batch(int device, hipStream_t stream, const uint32_t *indexVector, uint32_t start, int n) {
  if (!fuse_available()) return;
  bool release = false;
  {
    DeferLock lock(device);
    auto lim = t_state->limbo.find({device, stream});
    if (lim != t_state->limbo.end()) {  // the skipped work of the previous batch dies, and with it the compaction it would need
      if (lim->second.idx) t_state->compactions.erase(lim->second.idx);
      const uint32_t *sortIdx = lim->second.sortIdx;
      t_state->limbo.erase(lim);
      if (sortIdx) drop_sort(sortIdx);
      release = true;
    }
    t_state->compactions.erase(indexVector);  // the vector is redefined
    t_state->expansions.erase({device, stream});  // (what still reads a decoded column keeps it alive itself)
    {  // ARES_RTC_TRACE (diagnostics): who holds decoded run-length columns when a batch begins
      static int calls = 0;
      if ((++calls & 15) == 0) {
        size_t j = 0, jn = 0, c = 0, cn = 0, p = 0, l = 0, e = 0;
        for (auto &kv : t_state->journals) { jn++; j += kv.second.keep.size(); }
        for (auto &kv : t_state->compactions) { cn++; for (auto &t : kv.second.todo) c += t.keep ? 1 : 0; }
        for (auto &kv : t_state->pending) p += kv.second.keep.size();
        for (auto &kv : t_state->limbo) l += kv.second.keep.size();
        for (auto &kv : t_state->expansions) e += kv.second.size();
        if (j + c + p + l + e > 16) {
          char what[200];
          snprintf(what, sizeof(what), "decoded columns held at begin_batch: journals %zu (in %zu), compactions %zu (in %zu), pending %zu, limbo %zu, expansions %zu",
                   j, jn, c, cn, p, l, e);
          slow_trace(what, 0.0);
        }
      }
    }
    drop_sort(indexVector);                   // ... and so is whatever a lazily defined Sort meant it to hold
    {  // the stream's filters of the batch that just ended are what the new batch's are predicted from
      FilterHistory &h = t_state->filterHistory[{device, stream}];
      if (!h.current.empty()) {  // (the Sort path opens a second index vector per batch: no filters there, nothing to learn)
        h.previous.swap(h.current);
        h.current.clear();
      }
    }
    FilterJournal j;
    j.device = device;
    j.stream = stream;
    j.start = start;
    j.n0 = n;
    j.valid = true;
    t_state->journals[indexVector] = j;
  }
  if (release) g_releaseHeld(device, hold_tag(stream));
}

// a fast filter has compacted `indexVector` (f == nullptr: something else has — forget the journal)
static void journal_filter(int device, const uint32_t *indexVector, const FastOperands *f, uint32_t colRows, int rowsBefore,
                           std::shared_ptr<StreamBuffer> keep = nullptr) {
  if (!fuse_available()) return;
  DeferLock lock(device);
  auto it = t_state->journals.find(indexVector);
  if (it == t_state->journals.end()) return;
  FilterJournal &j = it->second;
  // (a filter that is not called with what the previous one returned works on something the journal does not describe)
  if (!f || !j.valid || j.filters.size() >= static_cast<size_t>(kFusedFilters) ||
      (j.filters.empty() && rowsBefore != j.n0) || (!j.filters.empty() && j.lastCount >= 0 && rowsBefore != j.lastCount)) {
    j.valid = false;
    return;
  }
  j.lastCount = -1;  // (set by the caller once the count is known)
  FastOperands copy = *f;
  copy.idx = nullptr;
  copy.pad = 0;
  j.filters.push_back(copy);
  j.colRows.push_back(colRows);
  if (keep) j.keep.push_back(std::move(keep));
}

void invalidate_filter_journal(int device, const uint32_t *indexVector) { journal_filter(device, indexVector, nullptr, 0, 0); }

// the filter that was journalled last kept `count` rows
static void journal_count(int device, const uint32_t *indexVector, int count) {
  if (!fuse_available()) return;
  DeferLock lock(device);
  auto it = t_state->journals.find(indexVector);
  if (it != t_state->journals.end() && it->second.valid) it->second.lastCount = count;
}

static bool journal_is_valid(int device, const uint32_t *indexVector) {
  if (!fuse_available()) return false;
  static const bool lazy = [] {
    const char *e = getenv("ARES_LAZY_COMPACT");
    return !(e && e[0] == '0');
  }();
  if (!lazy) return false;
  DeferLock lock(device);
  auto it = t_state->journals.find(indexVector);
  return it != t_state->journals.end() && it->second.valid;
}

// InitIndexVector: remember instead of writing (when the flush hook is in place)
static bool defer_iota(int device, hipStream_t stream, uint32_t *indexVector, uint32_t start, int n) {
  if (!defer_available() || n <= 0) return false;
  DeferLock lock(device);
  drop_sort(indexVector);
  t_state->iotas[indexVector] = PendingIota{device, stream, start, n};
  return true;
}

// true when `indexVector` is a virtual iota(0) of exactly n rows on this device; `take` removes it
// (the caller is about to give the vector real contents)
static bool virtual_iota(int device, uint32_t *indexVector, int n, bool take) {
  DeferLock lock(device);
  auto it = t_state->iotas.find(indexVector);
  if (it == t_state->iotas.end() || it->second.device != device || it->second.start != 0 || it->second.n != n || it->second.sorted) return false;
  if (take) t_state->iotas.erase(it);
  return true;
}

// Queues one fast-path transform; returns false when deferral is unavailable (the caller launches it).
static bool defer_transform(int device, hipStream_t stream, const FastOperands &f, const SinkD &s, int n, uint32_t colRows,
                            std::shared_ptr<StreamBuffer> keep = nullptr) {
  if (!defer_available()) return false;
  DeferLock lock(device);
  // everything pending on OTHER streams of the device is unrelated; only this stream's queue matters
  PendingQueue &q = t_state->pending[{device, stream}];
  ByteRange rv{reinterpret_cast<const uint8_t *>(f.vals), reinterpret_cast<const uint8_t *>(f.vals) + fast_value_bytes(f, colRows)};
  ByteRange rn{f.nulls, f.nulls ? f.nulls + (static_cast<uint64_t>(colRows) + f.bitOff + 7) / 8 + 2 : f.nulls};
  ByteRange ri{reinterpret_cast<const uint8_t *>(f.idx), reinterpret_cast<const uint8_t *>(f.idx) + (f.idx ? 4ull * n : 0)};
  ByteRange wv{s.values, s.values + static_cast<uint64_t>(s.width) * n};
  ByteRange wn{s.nulls, s.nulls ? s.nulls + n : s.nulls};
  bool conflict = q.jobs.count == kMaxMultiJobs || (q.jobs.count > 0 && (q.idx != f.idx || q.n != n));
  for (const ByteRange &w : q.writes) conflict = conflict || w.overlaps(rv) || w.overlaps(rn) || w.overlaps(ri) || w.overlaps(wv) || w.overlaps(wn);
  for (const ByteRange &r : q.reads) conflict = conflict || r.overlaps(wv) || r.overlaps(wn);
  if (conflict) launch_queue(stream, q);
  q.idx = f.idx;
  q.n = n;
  FastOperands fq = f;
  fq.pad = 0;
  q.jobs.f[q.jobs.count] = fq;
  q.jobs.s[q.jobs.count] = s;
  q.colRows[q.jobs.count] = colRows;
  q.jobs.count++;
  q.reads.push_back(rv);
  if (f.nulls) q.reads.push_back(rn);
  if (f.idx) q.reads.push_back(ri);
  q.writes.push_back(wv);
  if (s.nulls) q.writes.push_back(wn);
  if (keep) q.keep.push_back(std::move(keep));
  return true;
}

// ---- run-length encoded columns (archive batches) -------------------------------------------------------------------------
// rows [0, n) of a mode-3 column — [counts u32 x (runs + 1)][validity bit per run][value per run], row r lives in the run that
// holds startCount + r (query/iterator.hpp:199-278; locate() of device_model.hpp) — written as [validity bit per row][value
// per row].  A lane decodes 32 consecutive rows: one binary search, then a walk along the counts; one validity word.
__global__ __launch_bounds__(kBlock) void expand_runs_kernel(const uint32_t *counts, int numRuns, const uint8_t *nulls, uint32_t bitOff,
                                                             const uint8_t *values, int step, uint32_t startCount, int n, uint32_t *outNulls,
                                                             uint8_t *outValues) {
  // A wavefront decodes 2048 consecutive rows: lane l walks rows [32 l, 32 l + 32) of the tile (values into LDS, row r at word
  // r + r / 32: the lanes' columns fall into different banks), then the tile leaves the CU as whole lines — lane l stores rows
  // l, l + 64, ... (the first version stored each lane's 128 bytes where the lane walked: 64 scattered 4-byte stores per
  // instruction, 0.90 ms per 64 Mi rows instead of 0.1).
  __shared__ uint32_t sTile[kBlock / 64][2048 + 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t *tile = sTile[wave];
  const int64_t tiles = (static_cast<int64_t>(n) + 2047) / 2048;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave; t < tiles; t += static_cast<int64_t>(gridDim.x) * (kBlock / 64)) {
    const int64_t w = t * 64 + lane;  // this lane's validity word = its 32 rows
    const int64_t r0 = w * 32;
    uint32_t okWord = 0;
    if (r0 < n) {
      const uint32_t x0 = startCount + static_cast<uint32_t>(r0);
      uint32_t first = 0, last = static_cast<uint32_t>(numRuns);
      while (first < last) {
        const uint32_t mid = first + ((last - first) >> 1);
        if (counts[mid] > x0) last = mid; else first = mid + 1;
      }
      uint32_t run = first ? first - 1 : 0u, next = run + 1 < static_cast<uint32_t>(numRuns) ? counts[run + 1] : 0xFFFFFFFFu;
      uint32_t v = step == 4 ? reinterpret_cast<const uint32_t *>(values)[run] : step == 2 ? reinterpret_cast<const uint16_t *>(values)[run] : values[run];
      uint32_t ok = nulls ? get_bit(nulls, run + bitOff) : 1u;
      for (int j = 0; j < 32; j++) {
        const uint32_t x = x0 + static_cast<uint32_t>(j);
        if (x >= next) {
          while (run + 1 < static_cast<uint32_t>(numRuns) && counts[run + 1] <= x) run++;
          next = run + 1 < static_cast<uint32_t>(numRuns) ? counts[run + 1] : 0xFFFFFFFFu;
          v = step == 4 ? reinterpret_cast<const uint32_t *>(values)[run] : step == 2 ? reinterpret_cast<const uint16_t *>(values)[run] : values[run];
          ok = nulls ? get_bit(nulls, run + bitOff) : 1u;
        }
        okWord |= ok << j;
        tile[33 * lane + j] = v;
      }
      if (r0 + 32 > n) okWord &= (1u << static_cast<uint32_t>(n - r0)) - 1u;  // (rows past the end: no bits)
      outNulls[w] = okWord;
    }
    __builtin_amdgcn_wave_barrier();
    const int64_t base = t * 2048;
#pragma unroll 4
    for (int k = 0; k < 32; k++) {
      const int rr = k * 64 + lane;
      const int64_t r = base + rr;
      if (r < n) {
        const uint32_t v = tile[rr + (rr >> 5)];
        if (step == 4) reinterpret_cast<uint32_t *>(outValues)[r] = v;
        else if (step == 2) reinterpret_cast<uint16_t *>(outValues)[r] = static_cast<uint16_t>(v);
        else outValues[r] = static_cast<uint8_t>(v);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// A call's first operand is a run-length encoded column of a 32-bit kind and rows are raw rows (no base counts): the operand
// is re-bound to the column's decoded copy for the stream's batch — made now, on the call's stream, or found in the batch's
// cache.  Returns the copy (to be kept alive by whatever is defined over it), or null when the operand stays as it is.
// The copy is a SNAPSHOT: definitions made over it (journalled filters, queued transforms) see the column as it was when
// their call came, whatever the host writes into the column afterwards — like the call's own kernel would have.
static std::shared_ptr<StreamBuffer> decode_run_length_operand(int device, hipStream_t stream, OperandD &a, const uint32_t *indexVector,
                                                              const uint32_t *baseCounts, uint32_t startCount) {
  static EnvSwitch<bool> on("ARES_EXPAND_RLE", [](const char *e) { return !(e && e[0] == '0'); });
  if (a.type != OP_COLUMN || a.mode != 3 || baseCounts != nullptr || indexVector == nullptr || !on.get() || !fuse_available()) return nullptr;
  if (!(a.kind == K_I32 || a.kind == K_U32 || a.kind == K_F32) || a.length == 0) return nullptr;
  DeferLock lock(device);
  auto j = t_state->journals.find(indexVector);  // the batch's rows: what InitIndexVector was called with
  if (j == t_state->journals.end() || j->second.device != device || j->second.stream != stream || j->second.start != 0 || j->second.n0 <= 0)
    return nullptr;
  const int rows = j->second.n0;
  std::vector<RunExpansion> &cache = t_state->expansions[{device, stream}];
  const RunExpansion *hit = nullptr;
  for (const RunExpansion &e : cache)
    if (e.base == a.base && e.nullsOff == a.nullsOff && e.valuesOff == a.valuesOff && e.length == a.length && e.bitOff == a.bitOff &&
        e.step == a.step && e.startCount == startCount && e.rows == rows)
      hit = &e;
  RunExpansion fresh;
  if (!hit) {
    const size_t nullBytes = ((static_cast<size_t>(rows) + 31) / 32 * 4 + 63) / 64 * 64;
    fresh = RunExpansion{a.base, a.nullsOff, a.valuesOff, a.length, a.bitOff, a.step, startCount, rows,
                         std::make_shared<StreamBuffer>(nullBytes + static_cast<size_t>(a.step) * rows + 64, stream), nullBytes};
    uint8_t *out = fresh.copy->as<uint8_t>();
    const int64_t tiles = (static_cast<int64_t>(rows) + 2047) / 2048;
    ARES_LAUNCH("expand_runs_kernel", expand_runs_kernel, capped_grid((tiles + kBlock / 64 - 1) / (kBlock / 64), 256 * 8), kBlock, stream,
                reinterpret_cast<const uint32_t *>(a.base), static_cast<int>(a.length), a.base + a.nullsOff, static_cast<uint32_t>(a.bitOff),
                a.base + a.valuesOff, static_cast<int>(a.step), startCount, rows, reinterpret_cast<uint32_t *>(out), out + nullBytes);
    cache.push_back(fresh);
    hit = &cache.back();
  }
  a.base = hit->copy->as<uint8_t>();
  a.mode = 2;
  a.nullsOff = 0;
  a.valuesOff = static_cast<uint32_t>(hit->valuesAt);
  a.bitOff = 0;
  a.length = static_cast<uint32_t>(rows);
  return hit->copy;
}

// the host writes into (or frees) [r.lo, r.hi): decoded copies of run-length columns that lie in there are no longer what a
// NEW call would read.  Caller holds the device's DeferLock.
static void forget_run_expansions(int device, const ByteRange &r) {
  for (auto &kv : t_state->expansions) {
    if (kv.first.first != device) continue;
    auto &v = kv.second;
    for (size_t i = 0; i < v.size();) {
      const ByteRange src{v[i].base, v[i].base + v[i].valuesOff + static_cast<size_t>(v[i].step) * v[i].length};
      if (src.overlaps(r)) {
        v[i] = v.back();
        v.pop_back();
      } else {
        i++;
      }
    }
  }
}

static bool fast_sink(const SinkD &s) {
  const bool four = s.dtype == Int32 || s.dtype == Uint32 || s.dtype == Float32;
  if (s.type == SINK_MEASURE) return s.agg != AGGR_AVG_FLOAT && s.baseCounts == nullptr;
  // 1- / 2-byte dimension slots (city_id Uint16, status SmallEnum: query/common/dim_util.go:9-12) take integer results
  const bool narrow = s.type == SINK_DIM && (s.dtype == Int8 || s.dtype == Uint8 || s.dtype == Int16 || s.dtype == Uint16);
  return ((s.type == SINK_DIM || s.type == SINK_SCRATCH) && four) || narrow;
}

// Binds an array column and its functor (query/binder.hpp:385-426, :458-560): the second operand
// of a binary array functor is a constant whose type fits the element type; `odt` is the value type
// of the sink (Bool for a filter's predicate).  Combinations the reference resolves to its generic
// "null" functor (functor.hpp:468-513, :573-590, :700-723) run with enabled = 0.
static void bind_array(const InputVector *ins, int arity, int functor, int odt, ArrayD &a) {
  const ArrayVectorPartySlice &vp = ins[0].Vector.ArrayVP;
  memset(&a, 0, sizeof(a));
  a.kind = kind_of_datatype(vp.DataType);
  if (a.kind == K_NONE) throw std::invalid_argument("Unsupported data type for ArrayVectorPartyInput");
  a.dtype = vp.DataType;
  a.width = step_in_bytes(vp.DataType);
  a.descriptors = vp.OffsetLengthVector;
  a.values = vp.OffsetLengthVector + 8ull * vp.Length - vp.ValueOffsetAdj;
  a.functor = functor;
  const char *badOperand = "Unsupported data type when value type of first input iterator is ArrayVP Iterator";
  if (arity == 1) {
    a.enabled = functor == ArrayLength && odt == Uint32;
    return;
  }
  if (ins[1].Type != ConstantInput) throw std::invalid_argument(badOperand);
  const ConstantVector &c = ins[1].Vector.Constant;
  const int ct = c.DataType;
  const bool fits = a.kind == K_UUID ? (ct == ConstInt || ct == ConstUUID)
                    : a.kind == K_GEO ? (ct == ConstInt || ct == ConstGeoPoint || ct == ConstUUID)
                                      : (ct == ConstInt || ct == ConstFloat);
  if (!fits) throw std::invalid_argument(badOperand);
  if (functor == ArrayElementAt) {
    a.index = c.Value.IntVal;
    a.enabled = ct == ConstInt && (a.kind == K_UUID) == (odt == UUID) && (a.kind == K_GEO) == (odt == GeoPoint);
    return;
  }
  if (functor != ArrayContains || odt != Bool) return;
  if (a.kind == K_UUID) {
    a.enabled = ct == ConstUUID;
    memcpy(a.c, &c.Value.UUIDVal, 16);
    return;
  }
  if (a.kind == K_GEO) {
    // the reference carries this constant in the upper half of a 64-bit pointer
    // (iterator.hpp:484-516, SimpleIterator<GeoPointT>): only its first four bytes survive, Long is 0
    a.enabled = ct != ConstInt;
    memcpy(a.c, &c.Value.GeoPointVal, 4);
    return;
  }
  a.enabled = 1;  // val = static_cast<element type>(constant) (functor.hpp:655)
  const bool cf = ct == ConstFloat;
  const float fv = c.Value.FloatVal;
  const int32_t iv = c.Value.IntVal;
  switch (vp.DataType) {
    case Bool: a.c[0] = cf ? fv != 0.0f : iv != 0; break;
    case Int8: a.c[0] = static_cast<uint8_t>(cf ? static_cast<int8_t>(fv) : static_cast<int8_t>(iv)); break;
    case Uint8: a.c[0] = cf ? static_cast<uint8_t>(fv) : static_cast<uint8_t>(iv); break;
    case Int16: a.c[0] = static_cast<uint16_t>(cf ? static_cast<int16_t>(fv) : static_cast<int16_t>(iv)); break;
    case Uint16: a.c[0] = cf ? static_cast<uint16_t>(fv) : static_cast<uint16_t>(iv); break;
    case Int32: a.c[0] = static_cast<uint32_t>(cf ? static_cast<int32_t>(fv) : iv); break;
    case Uint32: a.c[0] = cf ? static_cast<uint32_t>(fv) : static_cast<uint32_t>(iv); break;
    case Float32: { const float x = cf ? fv : static_cast<float>(iv); uint32_t b; memcpy(&b, &x, 4); a.c[0] = b; break; }
    default: a.c[0] = static_cast<uint64_t>(cf ? static_cast<int64_t>(fv) : static_cast<int64_t>(iv)); break;  // Int64
  }
}

static void launch_array(const ArrayD &a, const SinkD &s, int n, hipStream_t stream) {
  const int grid = capped_grid((static_cast<int64_t>(n) + kBlock - 1) / kBlock);
  ARES_LAUNCH("array_transform_kernel", array_transform_kernel, grid, kBlock, stream, a, s, n);
}

// reports what a transform launched NOW writes (deferred ones report when their queue is launched)
static void note_sink(int device, const SinkD &s, int n) {
  if (n <= 0) return;
  mem_note_write(device, s.values, static_cast<size_t>(s.width) * n);
  if (s.nulls) mem_note_write(device, s.nulls, static_cast<size_t>(n));
}

// host twin of cvt32 (device_model.hpp)
static uint32_t host_cvt32(uint32_t bits, int from, int to) {
  if (from == to) return bits;
  auto asf = [](uint32_t b) { float f; memcpy(&f, &b, 4); return f; };
  auto fb = [](float f) { uint32_t b; memcpy(&b, &f, 4); return b; };
  switch (to) {
    case K_BOOL: return from == K_F32 ? (asf(bits) != 0.0f) : (bits != 0u);
    case K_I32: return from == K_F32 ? static_cast<uint32_t>(static_cast<int32_t>(asf(bits))) : bits;
    case K_U32: return from == K_F32 ? static_cast<uint32_t>(asf(bits)) : bits;
    default: return from == K_I32 ? fb(static_cast<float>(static_cast<int32_t>(bits))) : fb(static_cast<float>(bits));
  }
}

// What store_measure32 (device_model.hpp) stores for every row of `Noop(constant)`: host twin, run length 1.
static bool constant_measure_bits(const EvalParams &p, const SinkD &s, uint64_t *out) {
  if (p.rk != p.I || !(p.I == K_I32 || p.I == K_U32 || p.I == K_F32) || !(p.a.kind == K_I32 || p.a.kind == K_U32 || p.a.kind == K_F32))
    return false;
  if (!p.a.cok) {
    *out = s.width == 8 ? s.identity : static_cast<uint32_t>(s.identity);
    return true;
  }
  const uint32_t bits = host_cvt32(p.a.cbits, p.a.kind, p.I);
  const int rk = p.rk;
  auto asf = [](uint32_t b) { float f; memcpy(&f, &b, 4); return f; };
  switch (s.dtype) {
    case Int32: *out = host_cvt32(bits, rk, K_I32); return s.width == 4;
    case Uint32: *out = host_cvt32(bits, rk, K_U32); return s.width == 4;
    case Float32: *out = host_cvt32(bits, rk, K_F32); return s.width == 4;
    case Int64: {
      const int64_t v = rk == K_F32 ? static_cast<int64_t>(asf(bits)) : rk == K_I32 ? static_cast<int64_t>(static_cast<int32_t>(bits))
                                                                                    : static_cast<int64_t>(bits);
      *out = static_cast<uint64_t>(v);
      return s.width == 8;
    }
    case Float64: {
      const double d = rk == K_F32 ? static_cast<double>(asf(bits)) : rk == K_I32 ? static_cast<double>(static_cast<int32_t>(bits))
                                                                                  : static_cast<double>(bits);
      memcpy(out, &d, 8);
      return s.width == 8;
    }
    default: return false;
  }
}

static bool foreign_gather_enabled() {
  static EnvSwitch<bool> on("ARES_FOREIGN_GATHER", [](const char *e) { return !(e && e[0] == '0'); });
  return on.get();
}

static int run_transform(const InputVector *ins, int arity, const OutputVector &output, uint32_t *indexVector,
                         int n, uint32_t *baseCounts, uint32_t startCount, int functor, hipStream_t stream, int device) {
  if (n <= 0) {
    flush_deferred(device);
    return n < 0 ? 0 : n;
  }
  if (ins[0].Type == ArrayVectorPartyInput) {
    SinkD s;
    bind_sink(output, baseCounts, s);
    ArrayD a;
    bind_array(ins, arity, functor, s.dtype, a);
    if (s.type == SINK_MEASURE && s.baseCounts) a.idx = indexVector;
    if (s.type == SINK_DIM || s.type == SINK_MEASURE) {
      grouped_note_write(device, s.values, static_cast<size_t>(s.width) * n);
      if (s.nulls) grouped_note_write(device, s.nulls, static_cast<size_t>(n));
    }
    flush_deferred(device);
    launch_array(a, s, n, stream);
    note_sink(device, s, n);
    return n;
  }
  EvalParams p;
  SinkD s;
  CallTemps temps;
  build_params(ins, arity, stream, indexVector, baseCounts, startCount, functor, p, temps);
  // (archive batches: a run-length encoded column is read through its decoded copy — every fast path below applies)
  const std::shared_ptr<StreamBuffer> decoded = decode_run_length_operand(device, stream, p.a, indexVector, baseCounts, startCount);
  bind_sink(output, baseCounts, s);
  if (s.type == SINK_MEASURE && s.baseCounts && indexVector) p.needRow = 1;
  if (s.type == SINK_DIM || s.type == SINK_MEASURE) {  // rows of a result vector are (about to be) rewritten
    grouped_note_write(device, s.values, static_cast<size_t>(s.width) * n);
    if (s.nulls) grouped_note_write(device, s.nulls, static_cast<size_t>(n));
  }
  // a constant measure (COUNT(*) is SUM over the literal 1, query/aql_compiler.go:1191-1197): the rows are defined,
  // not written — Reduce over zero dimensions adds them up without anybody storing them (sort_reduce.hip)
  if (s.type == SINK_MEASURE && !s.baseCounts && arity == 1 && functor == Noop && p.a.type == OP_CONST && !is_wide(p.a.kind) &&
      s.agg != AGGR_AVG_FLOAT && (s.width == 4 || s.width == 8)) {
    uint64_t pattern = 0;
    if (constant_measure_bits(p, s, &pattern) && defer_fill(device, stream, s.values, static_cast<size_t>(s.width) * n, pattern, s.width))
      return n;
  }
  retire_fills_for_write(device, s.values, static_cast<size_t>(s.width) * n);
  if (s.nulls) retire_fills_for_write(device, s.nulls, static_cast<size_t>(n));
  FastOperands f;
  bool fast = fast_sink(s) && fast_operands(p, f, false);
  if (fast && s.type == SINK_DIM && s.width < 4 && !(f.rk == K_I32 || f.rk == K_U32)) fast = false;  // float -> narrow integer: generic kernel
  if (fast && f.idx && virtual_iota(device, indexVector, n, false)) f.idx = nullptr;  // rows = position
  // root outputs of the hot shape are held back and fused with their siblings (same index vector)
  if (fast && (s.type == SINK_DIM || s.type == SINK_MEASURE) && defer_transform(device, stream, f, s, n, p.a.length, decoded)) return n;
  flush_deferred(device);
  materialize_index_vector(device, indexVector);
  note_sink(device, s, n);
  if (fast) {
    f.pad = s.type == SINK_MEASURE ? 0 : static_cast<int>(reinterpret_cast<uintptr_t>(s.nulls) & 3);
    const int64_t numQuads = (static_cast<int64_t>(n) + f.pad + 3) / 4;
    const int64_t tiles = (numQuads + kBlock * kTQ - 1) / (kBlock * kTQ);
    ARES_LAUNCH("transform_fast_kernel", transform_fast_kernel, capped_grid(tiles, 256 * 16), kBlock, stream, f, s, n,
                numQuads);
  } else if (is_wide(p.a.kind)) {
    const int grid = capped_grid((static_cast<int64_t>(n) + kBlock - 1) / kBlock);
    ARES_LAUNCH("transform_wide_kernel", transform_wide_kernel, grid, kBlock, stream, p, s, n);
  } else if (foreign_gather_enabled() && p.arity == 1 && functor == Noop && p.a.type == OP_FOREIGN && !p.a.tz && !p.needRow && !s.baseCounts) {
    // a joined column into a dimension / measure vector: the join's transform on its own kernel
    constexpr int ITEMS = 8;
    const int grid = capped_grid((static_cast<int64_t>(n) + kBlock * ITEMS - 1) / (kBlock * ITEMS), 256 * 16);
    ARES_LAUNCH("transform_foreign_kernel", transform_foreign_kernel<ITEMS>, grid, kBlock, stream, p, s, n);
  } else {
    constexpr int ITEMS = 4;
    const int grid = capped_grid((static_cast<int64_t>(n) + kBlock * ITEMS - 1) / (kBlock * ITEMS), 256 * 16);
    ARES_LAUNCH("transform32_kernel", transform32_kernel<ITEMS>, grid, kBlock, stream, p, s, n);
  }
  return n;
}

// ---- ARES_FILTER_CHECK=<log file>: diagnostics for the two-phase filter (race hunting) -----------------------
// Around the real predicate kernel the inputs (index vector, column values, validity bitmap) are copied to pinned
// host memory on the SAME stream before and after it, the kernel is run a second time into scratch outputs, and
// once the call's own read-back has synchronised the stream everything is compared: the two runs with each other,
// the two input snapshots with each other, and a host evaluation of the predicate with both.  Same-stream work
// only: nothing here synchronises the device or another stream.
namespace {
struct FilterCheckBuffers {
  static constexpr int kMaxRows = 1 << 20;
  uint8_t *dPred2 = nullptr;
  uint32_t *dCounts2 = nullptr;  // [tile counts ... ][partials ...]
  uint32_t *hIdx[2] = {nullptr, nullptr}, *hVals[2] = {nullptr, nullptr};
  uint8_t *hNulls[2] = {nullptr, nullptr}, *hPred[2] = {nullptr, nullptr};
  uint32_t *hParts2 = nullptr;
  bool ok = false;
  FilterCheckBuffers() {
    bool good = hipMalloc(reinterpret_cast<void **>(&dPred2), kMaxRows + 64) == hipSuccess &&
                hipMalloc(reinterpret_cast<void **>(&dCounts2), 4 * (kMaxRows / 4096 + 2 + 4096 + 16)) == hipSuccess;
    for (int k = 0; k < 2 && good; k++) {
      good = good && hipHostMalloc(reinterpret_cast<void **>(&hIdx[k]), 4ull * kMaxRows + 64, hipHostMallocPortable) == hipSuccess;
      good = good && hipHostMalloc(reinterpret_cast<void **>(&hVals[k]), 4ull * kMaxRows + 64, hipHostMallocPortable) == hipSuccess;
      good = good && hipHostMalloc(reinterpret_cast<void **>(&hNulls[k]), kMaxRows / 8 + 64, hipHostMallocPortable) == hipSuccess;
      good = good && hipHostMalloc(reinterpret_cast<void **>(&hPred[k]), kMaxRows + 64, hipHostMallocPortable) == hipSuccess;
    }
    good = good && hipHostMalloc(reinterpret_cast<void **>(&hParts2), 4 * 4096 + 64, hipHostMallocPortable) == hipSuccess;
    ok = good;
    if (!good) (void)hipGetLastError();
  }
};
const char *filter_check_path() {
  static const char *p = getenv("ARES_FILTER_CHECK");
  return (p && p[0]) ? p : nullptr;
}
uint32_t host_compare_fast(const FastOperands &f, uint32_t bits, uint32_t ok) {
  auto asf = [](uint32_t b) { float x; memcpy(&x, &b, 4); return x; };
  const uint32_t x = host_cvt32(bits, f.akind, f.I), y = host_cvt32(f.bbits, f.bkind, f.I);
  const int ft = f.functor;
  bool c;
  if (f.I == K_F32) {
    const float a = asf(x), b = asf(y);
    c = ft == Equal ? a == b : ft == NotEqual ? a != b : ft == LessThan ? a < b : ft == LessThanOrEqual ? a <= b : ft == GreaterThan ? a > b : a >= b;
  } else if (f.I == K_I32) {
    const int32_t a = static_cast<int32_t>(x), b = static_cast<int32_t>(y);
    c = ft == Equal ? a == b : ft == NotEqual ? a != b : ft == LessThan ? a < b : ft == LessThanOrEqual ? a <= b : ft == GreaterThan ? a > b : a >= b;
  } else {
    c = ft == Equal ? x == y : ft == NotEqual ? x != y : ft == LessThan ? x < y : ft == LessThanOrEqual ? x <= y : ft == GreaterThan ? x > y : x >= y;
  }
  return (ok && f.bok && c) ? 1u : 0u;
}
struct FilterCheck {
  FilterCheckBuffers *b = nullptr;
  FastOperands f;
  uint32_t colRows = 0;
  int n = 0, tiles = 0, predGrid = 0;
  size_t nullBytes = 0;
  const uint8_t *pred = nullptr;
  hipStream_t stream = nullptr;
  bool active = false;
  void snapshot(int k) {
    if (f.idx) (void)hipMemcpyAsync(b->hIdx[k], f.idx, 4ull * n, hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(b->hVals[k], f.vals, 4ull * colRows, hipMemcpyDeviceToHost, stream);
    if (f.nulls) (void)hipMemcpyAsync(b->hNulls[k], f.nulls, nullBytes, hipMemcpyDeviceToHost, stream);
  }
  void begin(const FastOperands &fo, uint32_t rows, int n_, int tiles_, int predGrid_, const uint8_t *pred_, hipStream_t s) {
    if (!filter_check_path() || fo.step != 4 || n_ > FilterCheckBuffers::kMaxRows || rows > static_cast<uint32_t>(FilterCheckBuffers::kMaxRows)) return;
    thread_local FilterCheckBuffers bufs;
    if (!bufs.ok) return;
    b = &bufs;
    f = fo;
    colRows = rows;
    n = n_;
    tiles = tiles_;
    predGrid = predGrid_;
    pred = pred_;
    stream = s;
    nullBytes = f.nulls ? (static_cast<size_t>(colRows) + f.bitOff + 7) / 8 : 0;
    active = true;
    snapshot(0);
  }
  // right behind the real kernel, before anything rewrites the index vector
  void after_kernel() {
    if (!active) return;
    snapshot(1);
    (void)hipMemcpyAsync(b->hPred[0], pred, static_cast<size_t>(n), hipMemcpyDeviceToHost, stream);
    (void)hipMemsetAsync(b->dCounts2, 0, 4ull * (tiles + predGrid + 4), stream);
    // pred2 keeps the alignment phase of pred: the kernel's quad grid depends on it
    uint8_t *pred2 = b->dPred2 + (reinterpret_cast<uintptr_t>(pred) & 3) + ((4 - (reinterpret_cast<uintptr_t>(b->dPred2) & 3)) & 3);
    hipLaunchKernelGGL(filter_pred_kernel, dim3(predGrid), dim3(kBlock), 0, stream, f, pred2, b->dCounts2, n, tiles, b->dCounts2 + tiles + 2);
    (void)hipMemcpyAsync(b->hPred[1], pred2, static_cast<size_t>(n), hipMemcpyDeviceToHost, stream);
    (void)hipMemcpyAsync(b->hParts2, b->dCounts2 + tiles + 2, 4ull * predGrid, hipMemcpyDeviceToHost, stream);
  }
  // the stream has been synchronised by the call's own read-back
  void finish(uint32_t count1) {
    if (!active) return;
    (void)hipStreamSynchronize(stream);
    uint32_t count2 = 0;
    for (int k = 0; k < predGrid; k++) count2 += b->hParts2[k];
    uint32_t countHost[2] = {0, 0}, countPred[2] = {0, 0};
    int firstIdxDiff = -1, idxDiffs = 0, firstValDiff = -1, valDiffs = 0, firstNullDiff = -1, nullDiffs = 0, firstPredDiff = -1, predDiffs = 0;
    int firstHostDiff[2] = {-1, -1}, hostDiffs[2] = {0, 0};
    for (int i = 0; i < n; i++) {
      countPred[0] += b->hPred[0][i] != 0;
      countPred[1] += b->hPred[1][i] != 0;
      if (b->hPred[0][i] != b->hPred[1][i]) { if (firstPredDiff < 0) firstPredDiff = i; predDiffs++; }
      if (f.idx && b->hIdx[0][i] != b->hIdx[1][i]) { if (firstIdxDiff < 0) firstIdxDiff = i; idxDiffs++; }
      for (int k = 0; k < 2; k++) {
        const uint32_t row = f.idx ? b->hIdx[k][i] : static_cast<uint32_t>(i);
        uint32_t e = 0;
        if (row < colRows) {
          const uint32_t ok = f.nulls ? (b->hNulls[k][(row + f.bitOff) >> 3] >> ((row + f.bitOff) & 7)) & 1u : 1u;
          e = host_compare_fast(f, b->hVals[k][row], ok);
        }
        countHost[k] += e;
        if ((e != 0) != (b->hPred[0][i] != 0)) { if (firstHostDiff[k] < 0) firstHostDiff[k] = i; hostDiffs[k]++; }
      }
    }
    for (uint32_t r = 0; r < colRows; r++)
      if (b->hVals[0][r] != b->hVals[1][r]) { if (firstValDiff < 0) firstValDiff = static_cast<int>(r); valDiffs++; }
    for (size_t k = 0; k < nullBytes; k++)
      if (b->hNulls[0][k] != b->hNulls[1][k]) { if (firstNullDiff < 0) firstNullDiff = static_cast<int>(k); nullDiffs++; }
    const bool bad = count1 != count2 || count1 != countPred[0] || predDiffs || idxDiffs || valDiffs || nullDiffs || hostDiffs[0] || hostDiffs[1];
    static std::mutex logMutex;
    std::lock_guard<std::mutex> lock(logMutex);
    FILE *out = fopen(filter_check_path(), "a");
    if (!out) return;
    static long calls = 0;
    calls++;
    if (bad) {
      fprintf(out, "FILTERCHECK MISMATCH call %ld n %d colRows %u idx %p vals %p nulls %p bitOff %u pred %p stream %p functor %d I %d akind %d bbits %u: "
                   "count1 %u count2 %u countPred %u/%u countHost %u/%u predDiffs %d (first %d) idxDiffs %d (first %d) valDiffs %d (first %d) "
                   "nullDiffs %d (first byte %d) hostVsPred %d/%d (first %d/%d)\n",
              calls, n, colRows, (const void *)f.idx, (const void *)f.vals, (const void *)f.nulls, f.bitOff, (const void *)pred, (void *)stream, f.functor, f.I,
              f.akind, f.bbits, count1, count2, countPred[0], countPred[1], countHost[0], countHost[1], predDiffs, firstPredDiff, idxDiffs,
              firstIdxDiff, valDiffs, firstValDiff, nullDiffs, firstNullDiff, hostDiffs[0], hostDiffs[1], firstHostDiff[0], firstHostDiff[1]);
      int shown = 0;
      for (int i = 0; i < n && shown < 16; i++) {
        const bool d = b->hPred[0][i] != b->hPred[1][i] || (f.idx && b->hIdx[0][i] != b->hIdx[1][i]);
        bool hd = false;
        uint32_t rows[2], vals[2] = {0, 0}, oks[2] = {1, 1};
        for (int k = 0; k < 2; k++) {
          rows[k] = f.idx ? b->hIdx[k][i] : static_cast<uint32_t>(i);
          if (rows[k] < colRows) {
            vals[k] = b->hVals[k][rows[k]];
            oks[k] = f.nulls ? (b->hNulls[k][(rows[k] + f.bitOff) >> 3] >> ((rows[k] + f.bitOff) & 7)) & 1u : 1u;
            hd = hd || ((host_compare_fast(f, vals[k], oks[k]) != 0) != (b->hPred[0][i] != 0));
          }
        }
        if (d || hd) {
          fprintf(out, "  pos %d: pred %u/%u row %u/%u val %u/%u ok %u/%u\n", i, b->hPred[0][i], b->hPred[1][i], rows[0], rows[1], vals[0], vals[1], oks[0], oks[1]);
          shown++;
        }
      }
    } else if (calls % 2000 == 1) {
      fprintf(out, "filtercheck ok: %ld calls so far\n", calls);
    }
    fclose(out);
  }
};
}  // namespace

// ---- filters counted in row space (filter_rows_kernel) -----------------------------------------------------------
namespace {
// workgroups of `kernel` the device holds at once (occupancy x compute units), per device and kernel, asked once
int resident_blocks(int device, const void *kernel, int blockSize) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, int> known;
  std::lock_guard<std::mutex> lock(mu);
  auto it = known.find({device, kernel});
  if (it != known.end()) return it->second;
  int perCU = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, blockSize, 0) != hipSuccess || perCU < 1) {
    (void)hipGetLastError();
    perCU = 4;
  }
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) {
    (void)hipGetLastError();
    cus = 256;
  }
  return known[{device, kernel}] = perCU * cus;
}
int compute_units(int device) {
  static std::mutex mu;
  static std::map<int, int> known;
  std::lock_guard<std::mutex> lock(mu);
  auto it = known.find(device);
  if (it != known.end()) return it->second;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) {
    (void)hipGetLastError();
    cus = 256;
  }
  return known[device] = cus;
}

// ARES_FILTER_ROWSPACE=0: every filter of the hot shape takes the predicate-vector path (rounds 1-3)
bool row_space_enabled() {
  static EnvSwitch<bool> on("ARES_FILTER_ROWSPACE", [](const char *e) { return !(e && e[0] == '0'); });
  return on.get();
}
bool same_column(const FastOperands &a, const FastOperands &b) { return a.vals == b.vals && a.nulls == b.nulls && a.bitOff == b.bitOff; }
bool same_filter(const FastOperands &a, const FastOperands &b) {
  return same_column(a, b) && a.akind == b.akind && a.arity == b.arity && a.functor == b.functor && a.I == b.I && a.bkind == b.bkind &&
         a.bbits == b.bbits && a.bok == b.bok;
}

// May the call "filter the first n entries of indexVector by a hot-shape comparison over a column of colRows rows" be
// counted in row space?  The index vector must be iota(0 .. n0) with nothing but journalled filters since, all of them
// either applied or held as pending row-space filters, and n must be what the last of them returned.
bool row_space_eligible(int device, hipStream_t stream, const uint32_t *indexVector, int n, uint32_t colRows) {
  if (!fuse_available() || !row_space_enabled()) return false;
  DeferLock lock(device);
  auto j = t_state->journals.find(indexVector);
  if (j == t_state->journals.end() || !j->second.valid || j->second.device != device || j->second.stream != stream || j->second.start != 0 ||
      j->second.filters.size() >= static_cast<size_t>(kFusedFilters) || colRows < static_cast<uint32_t>(j->second.n0))
    return false;
  auto c = t_state->compactions.find(indexVector);
  if (j->second.filters.empty()) return n == j->second.n0 && c == t_state->compactions.end();
  return c != t_state->compactions.end() && !c->second.todo.empty() && c->second.stream == stream &&
         c->second.applied + c->second.todo.size() == j->second.filters.size() && c->second.lastCount == n;
}

// The call itself.  f: the filter (idx and pad are ignored: rows = positions).  Returns the survivor count.
int run_filter_rows(int device, hipStream_t stream, FastOperands f, uint32_t *indexVector, uint8_t *pred, int n, uint32_t colRows,
                    bool virtualIdx, const std::shared_ptr<StreamBuffer> &keep = nullptr) {
  f.idx = nullptr;
  f.pad = 0;
  // streaming (non-temporal) column loads: the column is read once by this kernel — 0.060 -> 0.054 ms per 64 Mi rows
  // (ARES_F_DEBUG=1024 switches them off)
  if (!(f.debug & 1024)) f.debug |= 128;
  constexpr int kGridCap = 2048;  // two partial counts per workgroup come back in one copy
  static_assert(2 * kGridCap <= kPinnedWords, "the partial counts are read back in one copy");
  std::shared_ptr<StreamBuffer> bits1, bits2;
  int grid = 0;
  bool two = false;
  FastOperands g = f;
  uint64_t generation = 0;
  {
    DeferLock lock(device);
    auto ins = t_state->compactions.find(indexVector);
    {
      // The eligibility the caller established (row_space_eligible) is re-established HERE, under the lock that books the
      // filter: between the caller's check and this point another thread's flush may have applied this vector's pending
      // filters (replay + compaction) and dropped the entry — the vector is then no longer "iota with filters pending",
      // and counting this filter over the batch's first n rows would count the wrong rows (found by the soak test: one
      // count in 512 four-thread programs).  The journal already holds this filter (journal_filter ran first).
      auto jr = t_state->journals.find(indexVector);
      const size_t journalled = (jr != t_state->journals.end() && jr->second.valid) ? jr->second.filters.size() : 0;
      const bool first = ins == t_state->compactions.end() && journalled == 1 && jr->second.n0 == n;
      const bool next = ins != t_state->compactions.end() && !ins->second.todo.empty() && ins->second.stream == stream &&
                        ins->second.applied + ins->second.todo.size() + 1 == journalled && ins->second.lastCount == n;
      if (!first && !next) return -1;
    }
    if (ins == t_state->compactions.end()) {  // the batch's first filter: the vector is iota(0 .. n)
      PendingCompact fresh{};
      fresh.device = device;
      fresh.stream = stream;
      fresh.idx = indexVector;
      fresh.virtualIdx = virtualIdx;
      fresh.n0 = n;
      fresh.applied = 0;
      fresh.lastCount = n;
      ins = t_state->compactions.emplace(indexVector, fresh).first;
    }
    PendingCompact &c = ins->second;
    FilterHistory &h = t_state->filterHistory[{device, stream}];
    FilterShape shape{f.akind, f.functor, f.I, f.bkind, f.bbits, f.bok, !c.todo.empty() && same_column(c.todo.back().f, f)};
    const size_t k = h.current.size();
    h.current.push_back(shape);
    if (c.predicted.valid && same_filter(c.predicted.g, f)) {  // counted together with the previous filter: no kernel
      const int count = c.predicted.count;
      c.todo.push_back(LazyFilter{f, colRows, pred, n, keep});
      c.bits = c.predicted.bits;
      c.lastCount = count;
      c.predicted = PredictedFilter{};
      auto j = t_state->journals.find(indexVector);
      if (j != t_state->journals.end() && j->second.valid) j->second.lastCount = count;
      return count;
    }
    c.predicted = PredictedFilter{};
    // the filter the previous batch of this stream had right behind this one, if it was on the same column
    if (k + 1 < h.previous.size() && h.previous[k].same_as(shape) && h.previous[k + 1].sameColumnAsPrevious &&
        h.previous[k + 1].akind == f.akind && c.applied + c.todo.size() + 2 <= static_cast<size_t>(kFusedFilters)) {
      const FilterShape &nx = h.previous[k + 1];
      g.functor = nx.functor;
      g.I = nx.I;
      g.bkind = nx.bkind;
      g.bbits = nx.bbits;
      g.bok = nx.bok;
      two = true;
    }
    const int64_t numQuads = (static_cast<int64_t>(c.n0) + 3) / 4;
    const int tiles = static_cast<int>((numQuads + kBlock * kPQ - 1) / (kBlock * kPQ));
    // one wave of workgroups: as many as the device holds at once (a grid of 1.3 x that runs two rounds, the second a
    // third full — measured: 0.104 ms instead of 0.079 per 64 Mi rows), each striding over its share of the tiles
    const bool hasIn = static_cast<bool>(c.bits);
    auto kernel = two ? (hasIn ? &filter_rows_kernel<true, true> : &filter_rows_kernel<true, false>)
                      : (hasIn ? &filter_rows_kernel<false, true> : &filter_rows_kernel<false, false>);
    {
      // Workgroups: a multiple of the compute units, one fewer per unit than fits.  Measured per 64 Mi rows (16 384 tiles,
      // 7 workgroups fit per unit): 4 per unit 0.083 ms, 5: 0.077, 6: 0.073, 7: 0.094, 2048 in all: 0.089, 1 639 (every
      // workgroup exactly ten tiles, but 103 units with a seventh workgroup): 0.099 — the units, not the workgroups, must
      // carry equal shares (profiles/r4_experiments.md).
      const int resident = resident_blocks(device, reinterpret_cast<const void *>(kernel), kBlock), cus = compute_units(device);
      const int perUnit = std::max(1, std::min(resident / cus - 1, 6));
      grid = capped_grid(tiles, std::min(kGridCap, cus * perUnit));
    }
    if (f.debug & 16) grid = capped_grid(tiles, kGridCap);        // (experiments: 2048 workgroups whatever fits at once,
    if (f.debug & 32) grid = capped_grid(tiles, 256 * 4);         //  four / eight per compute unit)
    if (f.debug & 64) grid = capped_grid(tiles, 256 * 8);
    if ((f.debug >> 12) & 15) grid = capped_grid(tiles, 256 * ((f.debug >> 12) & 15));  // (experiment: workgroups per compute unit)
    grid = std::min(grid, kGridCap);  // whatever the experiment switches asked for: two partial counts per workgroup fit the pinned slot
    const size_t bitBytes = static_cast<size_t>(tiles) * kBlock * sizeof(uint16_t);  // 16 rows per lane and tile
    bits1 = std::make_shared<StreamBuffer>(bitBytes, stream);
    if (two) bits2 = std::make_shared<StreamBuffer>(bitBytes, stream);
    // the partial counts go straight into the calling thread's pinned result slot (host memory the device can write):
    // the call then only waits for the stream — no copy command behind the kernel
    uint32_t *hostPartials = reinterpret_cast<uint32_t *>(pinned_words());
    ARES_LAUNCH("filter_rows_kernel", kernel, grid, kBlock, stream, f, g,
                hasIn ? c.bits->as<uint16_t>() : static_cast<const uint16_t *>(nullptr), bits1->as<uint16_t>(),
                two ? bits2->as<uint16_t>() : static_cast<uint16_t *>(nullptr), c.n0, tiles, hostPartials);
    // booked before the count is known: a flush from another thread that applies the pending filters meanwhile
    // applies this one too
    c.todo.push_back(LazyFilter{f, colRows, pred, n, keep});
    c.bits = bits1;
    c.lastCount = -1;
    generation = ++t_state->filterGeneration;
    c.generation = generation;
  }
  hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
  const volatile uint32_t *parts = reinterpret_cast<const volatile uint32_t *>(pinned_words());
  uint32_t count = 0, count2 = 0;
  for (int b = 0; b < grid; b++) {
    count += parts[2 * b];
    count2 += parts[2 * b + 1];
  }
  {
    DeferLock lock(device);
    auto it = t_state->compactions.find(indexVector);
    if (it != t_state->compactions.end() && it->second.generation == generation) {
      it->second.lastCount = static_cast<int>(count);
      if (two) {
        it->second.predicted.valid = true;
        it->second.predicted.g = g;
        it->second.predicted.count = static_cast<int>(count2);
        it->second.predicted.bits = bits2;
      }
    }
    auto j = t_state->journals.find(indexVector);
    if (j != t_state->journals.end() && j->second.valid) j->second.lastCount = static_cast<int>(count);
  }
  return static_cast<int>(count);
}
}  // namespace

static int run_filter(const InputVector *ins, int arity, uint32_t *indexVector, uint8_t *pred, int n,
                      RecordID **recordIDVectors, int numForeignTables, uint32_t *baseCounts, uint32_t startCount,
                      int functor, hipStream_t stream, int device) {
  if (numForeignTables < 0 || numForeignTables > 8) throw std::invalid_argument("only support up to 8 foreign tables");
  if (n <= 0) {
    flush_deferred(device);
    return 0;
  }
  EvalParams p;
  CallTemps temps;
  const bool isArray = ins[0].Type == ArrayVectorPartyInput;
  ArrayD arr;
  std::shared_ptr<StreamBuffer> decoded;  // the decoded copy of a run-length encoded column (archive batches), or null
  if (isArray) {
    bind_array(ins, arity, functor, Bool, arr);
    memset(&p, 0, sizeof(p));
    p.a.kind = K_UUID;  // routed like a wide operand: predicate first, then compaction by predicate
  } else {
    build_params(ins, arity, stream, indexVector, baseCounts, startCount, functor, p, temps);
    decoded = decode_run_length_operand(device, stream, p.a, indexVector, baseCounts, startCount);
  }
  p.needRow = 1;
  static const bool onePass = [] {
    const char *e = getenv("ARES_FILTER");
    return e && strcmp(e, "onepass") == 0;
  }();
  // a filter of the hot shape consumes a not-yet-written iota index vector directly; everything
  // else that is pending on the device is launched first
  bool virtualIdx = false, rowSpace = false;
  {
    FastOperands probe;
    if (!is_wide(p.a.kind) && indexVector != nullptr && numForeignTables == 0 && fast_operands(p, probe, true)) {
      virtualIdx = virtual_iota(device, indexVector, n, true);
      // ... and, while every filter of the batch has been of that shape, is only counted — over the batch's rows, with
      // the filters before it (run_filter_rows): the pending ones of this very vector stay pending through the flush
      rowSpace = !onePass && baseCounts == nullptr && row_space_eligible(device, stream, indexVector, n, p.a.length);
    }
  }
  if (rowSpace) flush_deferred_impl(device, nullptr, nullptr, indexVector);
  else flush_deferred(device);
  // (the flush may have applied them after all: queued transforms that read the vector were launched)
  rowSpace = rowSpace && row_space_eligible(device, stream, indexVector, n, p.a.length);
  if (!rowSpace) mem_note_write(device, pred, static_cast<size_t>(n));  // (a counted filter writes no predicate bytes unless it is replayed)
  materialize_index_vector(device, indexVector);
  if (is_wide(p.a.kind)) {
    // wide operands: evaluate the predicate with the wide transform kernel, then compact by pred
    SinkD s;
    bind_pred_sink(pred, s);
    p.needRow = indexVector != nullptr;
    const int grid = capped_grid((static_cast<int64_t>(n) + kBlock - 1) / kBlock);
    if (isArray)
      launch_array(arr, s, n, stream);
    else
      ARES_LAUNCH("transform_wide_kernel", transform_wide_kernel, grid, kBlock, stream, p, s, n);
  }
  FastOperands f;
  const bool fast = !is_wide(p.a.kind) && indexVector != nullptr && fast_operands(p, f, true);
  if (fast && !onePass && numForeignTables == 0 && baseCounts == nullptr)
    journal_filter(device, indexVector, &f, p.a.length, n, decoded);
  else
    journal_filter(device, indexVector, nullptr, 0, 0);
  if (rowSpace && fast && journal_is_valid(device, indexVector)) {
    const int counted = run_filter_rows(device, stream, f, indexVector, pred, n, p.a.length, virtualIdx, decoded);
    if (counted >= 0) return counted;
    // (-1: somebody's flush applied the vector's pending filters since the eligibility check: the ordinary filter below)
  }
  if (rowSpace) {  // the predicate vector is written after all
    mem_note_write(device, pred, static_cast<size_t>(n));
    DeferLock lock(device);
    run_compaction(indexVector);
  }
  if (fast && !onePass) {
    // two-phase path: predicate + tile counts, scan, chain-free compaction of the index vector and
    // of every RecordID vector
    f.pad = static_cast<int>(reinterpret_cast<uintptr_t>(pred) & 3);
    const int64_t numQuads = (static_cast<int64_t>(n) + f.pad + 3) / 4;
    const int tiles = static_cast<int>((numQuads + kBlock * kPQ - 1) / (kBlock * kPQ));
    const int passes = 1 + numForeignTables;
    // layout: [total, error, tickets[passes], pad][tileCounts][tileOffsets + 1][loaded x passes][one partial count per workgroup]
    const size_t head = 64;
    constexpr int kPredGridCap = 256 * 16;
    static_assert(kPredGridCap <= kPinnedWords, "the partial counts are read back in one copy");
    const int predGrid = capped_grid(tiles, kPredGridCap);
    const size_t words = static_cast<size_t>(tiles) * (2 + passes) + 1 + static_cast<size_t>(predGrid);
    auto wsBuf = std::make_shared<StreamBuffer>(head + 4 * words, stream);
    uint32_t *w = wsBuf->as<uint32_t>();
    uint32_t *total = w, *error = w + 1;
    unsigned int *tickets = w + 2;
    uint32_t *tileCounts = w + 16, *tileOffsets = tileCounts + tiles, *loaded = tileOffsets + tiles + 1;
    hip_check(hipMemsetAsync(w, 0, head + 4 * words, stream), "hipMemsetAsync");  // one fill: head, counts, flags
    if (virtualIdx) f.idx = nullptr;  // rows = position
    const bool lazy = numForeignTables == 0 && journal_is_valid(device, indexVector);
    // ARES_FILTER_LAZY_SCAN=0: the tile offsets are computed at once, as before round 3
    static EnvSwitch<bool> lazyScan("ARES_FILTER_LAZY_SCAN", [](const char *e) { return !(e && e[0] == '0'); });
    const bool scanLater = lazy && lazyScan.get();
    uint32_t *partials = loaded + static_cast<size_t>(tiles) * passes;
    FilterCheck check;
    check.begin(f, p.a.length, n, tiles, predGrid, pred, stream);
    ARES_LAUNCH("filter_pred_kernel", filter_pred_kernel, predGrid, kBlock, stream, f, pred, tileCounts, n, tiles,
                scanLater ? partials : nullptr);
    check.after_kernel();
    if (!scanLater) ARES_LAUNCH("filter_scan_kernel", filter_scan_kernel, 1, 1024, stream, tileCounts, tileOffsets, tiles, total);
    if (lazy) {
      // The count is known; the compaction waits until somebody needs the compacted vector — a
      // HashReduce that re-derives the survivors from the journal never does.
      uint32_t result[2] = {0, 0};
      if (scanLater) {
        uint32_t parts[kPredGridCap];
        read_back_u32(partials, parts, predGrid, stream);
        for (int b = 0; b < predGrid; b++) result[0] += parts[b];
      } else {
        read_back_u32(total, result, 2, stream);
      }
      PendingCompact c;
      c.device = device;
      c.stream = stream;
      c.idx = indexVector;
      c.pred = pred;
      c.n = n;
      c.pad = f.pad;
      c.tiles = tiles;
      c.virtualIdx = virtualIdx;
      c.ws = wsBuf;
      c.ticket = tickets;
      c.error = error;
      c.tileOffsets = tileOffsets;
      c.loaded = loaded;
      if (scanLater) {
        c.tileCounts = tileCounts;
        c.total = total;
      }
      check.finish(result[0]);
      DeferLock lock(device);
      t_state->compactions[indexVector] = c;
      return static_cast<int>(result[0]);
    }
    const int cgrid = capped_grid((tiles + kTilesPerTicket - 1) / kTilesPerTicket, 256 * 8);
    if (virtualIdx) f.idx = nullptr;
    mem_note_write(device, indexVector, 4ull * static_cast<size_t>(n));
    for (int t = 0; t < numForeignTables; t++) mem_note_write(device, recordIDVectors[t], 8ull * static_cast<size_t>(n));
    for (int pass = 0; pass < passes; pass++) {
      CompactWorkspace cw;
      cw.ticket = tickets + pass;
      cw.error = error;
      cw.tileOffsets = tileOffsets;
      cw.loaded = loaded + static_cast<size_t>(tiles) * pass;
      if (pass == 0 && virtualIdx)
        ARES_LAUNCH("filter_compact_kernel<iota>", (filter_compact_kernel<uint32_t, true>), cgrid, kBlock, stream, pred, indexVector,
                    0u, f.pad, cw, n, tiles);
      else if (pass == 0)
        ARES_LAUNCH("filter_compact_kernel", (filter_compact_kernel<uint32_t, false>), cgrid, kBlock, stream, pred, indexVector, 0u,
                    f.pad, cw, n, tiles);
      else
        ARES_LAUNCH("filter_compact_kernel<rid>", (filter_compact_kernel<uint64_t, false>), cgrid, kBlock, stream, pred,
                    reinterpret_cast<uint64_t *>(recordIDVectors[pass - 1]), 0u, f.pad, cw, n, tiles);
    }
    uint32_t result[2] = {0, 0};  // {survivors, error}
    read_back_u32(total, result, 2, stream);
    check.finish(result[0]);
    if (result[1]) throw AlgorithmError("ERROR: filter: compaction wait timed out");
    return static_cast<int>(result[0]);
  }
  if (virtualIdx) {  // the one-pass kernels read the index vector: write it now
    DeferLock lock(device);
    launch_init_index(indexVector, 0, n, stream);
  }
  mem_note_write(device, indexVector, 4ull * static_cast<size_t>(n));
  for (int t = 0; t < numForeignTables; t++) mem_note_write(device, recordIDVectors[t], 8ull * static_cast<size_t>(n));
  int numTiles = static_cast<int>((static_cast<int64_t>(n) + kFilterTile - 1) / kFilterTile);
  int fastTiles = 0;
  if (fast) {
    f.pad = static_cast<int>(reinterpret_cast<uintptr_t>(pred) & 3);
    const int64_t numQuads = (static_cast<int64_t>(n) + f.pad + 3) / 4;
    fastTiles = static_cast<int>((numQuads + kBlock * kFQ - 1) / (kBlock * kFQ));
  }
  const int passes = 1 + numForeignTables;
  const size_t passBytes = 16 + sizeof(uint64_t) * static_cast<size_t>(numTiles);
  StreamBuffer wsBuf(passBytes * passes, stream);
  hip_check(hipMemsetAsync(wsBuf.get(), 0, passBytes * passes, stream), "hipMemsetAsync");
  const int grid = capped_grid(numTiles);
  uint32_t *totalDev = nullptr;
  for (int pass = 0; pass < passes; pass++) {
    uint8_t *base = wsBuf.as<uint8_t>() + passBytes * pass;
    ScanWorkspace ws;
    ws.ticket = reinterpret_cast<unsigned int *>(base);
    ws.total = reinterpret_cast<uint32_t *>(base + 4);
    ws.error = reinterpret_cast<uint32_t *>(wsBuf.as<uint8_t>() + 8);  // shared by all passes
    ws.status = reinterpret_cast<uint64_t *>(base + 16);
    if (pass == 0) {
      totalDev = ws.total;
      if (is_wide(p.a.kind)) {
        ARES_LAUNCH("filter_kernel<2>", filter_kernel<2>, grid, kBlock, stream, p, pred, indexVector,
                           static_cast<uint64_t *>(nullptr), ws, n, numTiles);
      } else if (fast) {
        ARES_LAUNCH("filter_fast_kernel", filter_fast_kernel, capped_grid(fastTiles, (f.debug & 4) ? 256 * 3 : 256 * 5), kBlock, stream, f, pred,
                    indexVector, ws, n, fastTiles);
      } else {
        ARES_LAUNCH("filter_kernel<0>", filter_kernel<0>, grid, kBlock, stream, p, pred, indexVector,
                    static_cast<uint64_t *>(nullptr), ws, n, numTiles);
      }
    } else {
      ARES_LAUNCH("filter_kernel<1>", filter_kernel<1>, grid, kBlock, stream, p, pred, indexVector,
                  reinterpret_cast<uint64_t *>(recordIDVectors[pass - 1]), ws, n, numTiles);
    }
  }
  uint32_t result[2] = {0, 0};  // {survivors, error}
  read_back_u32(totalDev, result, 2, stream);
  if (result[1]) throw AlgorithmError("ERROR: filter: inter-tile scan timed out");
  return static_cast<int>(result[0]);
}

// ---------------------------------------------------------------------------------------------
// second-stage fusion: notifications from libmem.so, and HashReduce consuming the pending queue
// ---------------------------------------------------------------------------------------------
namespace {
ByteRange range_of(const void *ptr, size_t bytes) {
  const uint8_t *lo = static_cast<const uint8_t *>(ptr);
  return ByteRange{lo, lo + (bytes ? bytes : 1)};
}
bool touches(const std::vector<ByteRange> &v, const ByteRange &r) {
  for (const ByteRange &x : v)
    if (x.overlaps(r)) return true;
  return false;
}
// the index vector itself (its first n0 entries) or a column one of the journalled filters read
bool journal_touched(const uint32_t *indexVector, const FilterJournal &j, const ByteRange &r, bool *indexOnly) {
  const bool idx = range_of(indexVector, 4ull * (j.n0 > 0 ? j.n0 : 1)).overlaps(r);
  bool cols = false;
  for (size_t k = 0; k < j.filters.size(); k++) {
    const FastOperands &f = j.filters[k];
    cols = cols || range_of(f.vals, fast_value_bytes(f, j.colRows[k])).overlaps(r);
    if (f.nulls) cols = cols || range_of(f.nulls, (static_cast<uint64_t>(j.colRows[k]) + f.bitOff + 7) / 8 + 2).overlaps(r);
  }
  if (indexOnly) *indexOnly = idx && !cols;
  return idx || cols;
}

struct DeviceGuard {  // hooks run on libmem.so's threads: select the device, restore it afterwards
  int previous = -1;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&previous) != hipSuccess) previous = -1;
    if (previous != device) (void)hipSetDevice(device);
  }
  ~DeviceGuard() {
    if (previous >= 0) (void)hipSetDevice(previous);
  }
};

// WaitForCudaStream: a queue whose results only a HashReduce on the same stream will consume may stay
void hook_on_wait(int device, void *streamPtr) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(streamPtr);
  try {
    DeviceGuard guard(device);
    DeferLock lock(device);
    // a wait on one stream says nothing about the others: their pending work is left alone
    for (auto it = t_state->iotas.begin(); it != t_state->iotas.end();) {
      if (it->second.device == device && it->second.stream == stream && !it->second.consumed) {
        launch_init_index(it->first, it->second.start, it->second.n, it->second.stream);
        it = t_state->iotas.erase(it);
      } else {
        ++it;
      }
    }
    for (auto &kv : t_state->pending) {
      if (kv.first.first != device || kv.first.second != stream || kv.second.jobs.count == 0 || kv.second.overWait) continue;
      PendingQueue &q = kv.second;
      bool keep = g_fuseEnabled;  // nobody will consume the queue otherwise
      if (keep && q.idx) {  // the survivors must be re-derivable from the filter journal
        auto j = t_state->journals.find(q.idx);
        keep = j != t_state->journals.end() && j->second.valid && j->second.device == device && j->second.stream == stream &&
               j->second.start == 0;
        if (keep)  // the filters' columns are inputs of the pending work from now on
          for (size_t k = 0; k < j->second.filters.size(); k++) {
            const FastOperands &f = j->second.filters[k];
            q.reads.push_back(range_of(f.vals, fast_value_bytes(f, j->second.colRows[k])));
            if (f.nulls) q.reads.push_back(range_of(f.nulls, (static_cast<uint64_t>(j->second.colRows[k]) + f.bitOff + 7) / 8 + 2));
          }
      }
      if (keep)
        q.overWait = true;
      else
        launch_queue(kv.first.second, q);
    }
  } catch (std::exception &e) {
    fprintf(stderr, "Exception happened when handling a stream wait: %s\n", e.what());
  }
}

// DeviceFree: nonzero = keep the block aside on behalf of that stream's pending (or skipped) work,
// which still reads it (the value is the tag AresMemReleaseHeld will be called with)
uintptr_t hook_on_free(int device, void *ptr, size_t bytes) {
  const ByteRange r = range_of(ptr, bytes);
  grouped_note_write(device, ptr, bytes);
  uintptr_t hold = 0;
  ReleaseSet released;
  try {
    DeviceGuard guard(device);
    DeferLock lock(device);
    for (auto it = t_state->sorts.begin(); it != t_state->sorts.end();) {  // a buffer a lazily defined Sort (+ Reduce) works on goes
      if (it->second.device == device && sort_touches(it->second, r)) {
        drop_sort(it->first);
        it = t_state->sorts.begin();
      } else {
        ++it;
      }
    }
    for (auto it = t_state->iotas.begin(); it != t_state->iotas.end();) {  // an index vector nobody has read yet
      const ByteRange v = range_of(it->first, 4ull * it->second.n);
      it = (it->second.device == device && v.overlaps(r)) ? t_state->iotas.erase(it) : std::next(it);
    }
    retire_fills(device, r, true);  // lazy fills of the block die unwritten
    forget_run_expansions(device, r);
    for (auto it = t_state->journals.begin(); it != t_state->journals.end();) {
      // a journal dies with its index vector or with a column its filters read — unless a queue the
      // host has already waited for still refers to it (then the block is held below, contents intact)
      const bool dead = it->second.device == device && journal_touched(it->first, it->second, r, nullptr);
      bool kept = false;
      if (dead)
        for (auto &kv : t_state->pending) kept = kept || (kv.second.jobs.count && kv.second.overWait && kv.second.idx == it->first);
      it = (dead && !kept) ? t_state->journals.erase(it) : std::next(it);
    }
    for (auto &kv : t_state->pending) {
      if (kv.first.first != device || kv.second.jobs.count == 0) continue;
      if (touches(kv.second.writes, r)) {
        released.add(kv.first.second);
        launch_queue(kv.first.second, kv.second);  // an output is freed: run the work, the fence follows it
      } else if (touches(kv.second.reads, r)) {
        if (kv.second.overWait) {
          hold = hold_tag(kv.first.second);
        } else {  // not even waited for: run it now, the free is fenced behind it
          released.add(kv.first.second);
          launch_queue(kv.first.second, kv.second);
        }
      }
    }
    for (auto it = t_state->compactions.begin(); it != t_state->compactions.end();) {
      if (it->second.device != device || !compaction_touches(it->second, r)) {
        ++it;
        continue;
      }
      // the index or predicate vector of a pending compaction is freed
      const uint32_t *key = it->first;
      const hipStream_t owner = it->second.stream;
      bool waited = false, queued = false, skipped = false;
      for (auto &kv : t_state->pending)
        if (kv.second.jobs.count && kv.second.idx == key) (kv.second.overWait ? waited : queued) = true;
      for (auto &kv : t_state->limbo) skipped = skipped || kv.second.idx == key;
      if (waited || skipped) {  // still needed if that work is launched after all: keep the block intact
        hold = hold_tag(owner);
        ++it;
      } else if (queued) {  // transforms the host has not even waited for: run everything now
        for (auto &kv : t_state->pending)
          if (kv.second.jobs.count && kv.second.idx == key) {
            released.add(kv.first.second);
            launch_queue(kv.first.second, kv.second);
          }
        it = t_state->compactions.begin();  // (launch_queue erased the entry)
      } else if (range_of(it->second.idx, 4ull * (it->second.todo.empty() ? it->second.n : it->second.n0)).overlaps(r)) {
        it = t_state->compactions.erase(it);  // the index vector itself goes: nobody will read the compacted vector
      } else {
        // only the predicate vector goes, the index vector stays live (a later transform, filter or
        // copy may read it): compact now — the free is fenced behind the launch
        run_compaction(key);
        it = t_state->compactions.begin();
      }
    }
    for (auto it = t_state->limbo.begin(); it != t_state->limbo.end();) {
      if (it->first.first == device && touches(it->second.writes, r)) {  // the skipped outputs die unseen
        if (it->second.idx) t_state->compactions.erase(it->second.idx);
        released.add(it->first.second);
        const uint32_t *sortIdx = it->second.sortIdx;
        it = t_state->limbo.erase(it);
        if (sortIdx) drop_sort(sortIdx);
      } else {
        if (it->first.first == device && touches(it->second.reads, r)) hold = hold_tag(it->first.second);
        ++it;
      }
    }
  } catch (std::exception &e) {
    fprintf(stderr, "Exception happened when handling a device free: %s\n", e.what());
  }
  released.run(device);
  return hold;
}

// a copy is about to touch [ptr, ptr + bytes)
void hook_on_access(int device, const void *ptr, size_t bytes) {
  const ByteRange r = range_of(ptr, bytes);
  ReleaseSet released;
  try {
    DeviceGuard guard(device);
    grouped_materialize_for_read(device, ptr, bytes);  // (measure rows a table image defines)
    DeferLock lock(device);
    for (auto it = t_state->iotas.begin(); it != t_state->iotas.end();) {
      const ByteRange v = range_of(it->first, 4ull * it->second.n);
      if (it->second.device == device && v.overlaps(r) && it->second.sorted) {
        t_syncAfterUnlock.push_back(it->second.stream);
        materialize_sort(it->first);
        it = t_state->iotas.begin();
      } else if (it->second.device == device && v.overlaps(r)) {
        launch_init_index(it->first, it->second.start, it->second.n, it->second.stream);
        t_syncAfterUnlock.push_back(it->second.stream);
        it = t_state->iotas.erase(it);
      } else {
        ++it;
      }
    }
    for (auto &kv : t_state->pending) {
      if (kv.first.first != device || kv.second.jobs.count == 0) continue;
      if (touches(kv.second.writes, r) || touches(kv.second.reads, r)) {
        released.add(kv.first.second);
        launch_queue(kv.first.second, kv.second);
      }
    }
    forget_run_expansions(device, r);  // (a copy INTO a run-length column: later calls decode it again)
    materialize_fills(device, &r, &t_syncAfterUnlock);  // (the copy may run on another stream: wait for the fill)
    materialize_limbo(device, &r, &released);
    for (auto it = t_state->compactions.begin(); it != t_state->compactions.end();) {
      if (it->second.device == device && compaction_touches(it->second, r)) {
        const hipStream_t cs = it->second.stream;
        run_compaction(it->first);
        t_syncAfterUnlock.push_back(cs);
        it = t_state->compactions.begin();
      } else {
        ++it;
      }
    }
    // a copy into an index vector or into a column a journalled filter has read: the survivors can
    // no longer be re-derived (queues that depended on the journal were launched above)
    for (auto it = t_state->journals.begin(); it != t_state->journals.end();)
      it = (it->second.device == device && journal_touched(it->first, it->second, r, nullptr)) ? t_state->journals.erase(it) : std::next(it);
  } catch (std::exception &e) {
    fprintf(stderr, "Exception happened when handling a device copy: %s\n", e.what());
  }
  released.run(device);
}
}  // namespace

namespace {
void hook_on_write(int device, const void *ptr, size_t bytes) { grouped_note_write(device, ptr, bytes); }

void hook_trim(int device) {
  try {
    DeviceGuard guard(device);
    stream_cache_trim(device);
    grouped_trim(device);
  } catch (std::exception &) {
  }
}

// DestroyCudaStream (the stream has been flushed and synchronised): nothing may outlive the handle —
// a later stream can get the same address
void hook_on_stream_destroy(int device, void *streamPtr) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(streamPtr);
  try {
    DeviceGuard guard(device);
    {
      DeferLock lock(device);
      for (auto it = t_state->sorts.begin(); it != t_state->sorts.end();) {
        if (it->second.device == device && it->second.stream == stream) {
          drop_sort(it->first);
          it = t_state->sorts.begin();
        } else {
          ++it;
        }
      }
      t_state->pending.erase({device, stream});
      t_state->expansions.erase({device, stream});
      auto lim = t_state->limbo.find({device, stream});
      if (lim != t_state->limbo.end()) {
        if (lim->second.idx) t_state->compactions.erase(lim->second.idx);
        t_state->limbo.erase(lim);
      }
      for (auto it = t_state->journals.begin(); it != t_state->journals.end();)
        it = (it->second.device == device && it->second.stream == stream) ? t_state->journals.erase(it) : std::next(it);
      for (auto it = t_state->compactions.begin(); it != t_state->compactions.end();)
        it = (it->second.device == device && it->second.stream == stream) ? t_state->compactions.erase(it) : std::next(it);
      for (auto it = t_state->iotas.begin(); it != t_state->iotas.end();)
        it = (it->second.device == device && it->second.stream == stream) ? t_state->iotas.erase(it) : std::next(it);
      t_state->filterHistory.erase({device, stream});
      // lazy fills that were defined on this stream and are still unwritten: from now on they are written on the stream
      // of whoever needs them (the null stream stands for "the caller's", see launch_fill)
      for (auto &kv : t_state->fills)
        if (kv.second.device == device && kv.second.stream == stream) kv.second.streamGone = true;
      // error words of this stream's lazy compactions: the stream is idle, their copies have landed; the events go now
      for (size_t i = 0; i < t_state->errorChecks.size();) {
        ErrorCheck &c = t_state->errorChecks[i];
        if (c.device == device && c.stream == stream) {
          t_state->errorSeen = t_state->errorSeen || *c.pinned != 0;
          (void)hipEventDestroy(c.done);
          t_state->errorSlots.push_back(c.pinned);
          t_state->errorChecks[i] = std::move(t_state->errorChecks.back());
          t_state->errorChecks.pop_back();
        } else {
          i++;
        }
      }
    }
    profiler_stream_gone(stream);
    if (g_releaseHeld) g_releaseHeld(device, hold_tag(stream));
    stream_cache_purge(device, stream);
  } catch (std::exception &e) {
    fprintf(stderr, "Exception happened when handling a stream destruction: %s\n", e.what());
  }
}
}  // namespace

// An entry point reads [ptr, ptr + bytes) with a kernel without going through a flush (Reduce over zero dimensions
// keeps everything else lazy): queued transforms that write into the range, and work a HashReduce skipped there, run
// first — in call order, on their own stream (a queue the host has already waited for is synchronised after its late
// launch, the blocks held for it are fenced behind it).
void launch_pending_writers(int device, const void *ptr, size_t bytes) {
  if (!ptr || bytes == 0 || !defer_available()) return;
  grouped_materialize_for_read(device, ptr, bytes);
  const ByteRange r = range_of(ptr, bytes);
  ReleaseSet released;
  {
    DeferLock lock(device);
    for (auto &kv : t_state->pending) {
      if (kv.first.first != device || kv.second.jobs.count == 0 || !touches(kv.second.writes, r)) continue;
      if (kv.second.overWait) released.add(kv.first.second);
      launch_queue(kv.first.second, kv.second);
    }
    materialize_limbo(device, &r, &released);
  }
  released.run(device);
}

// HashReduce's first move: when the stream's pending queue is exactly "the dimension columns and the
// measure of rows [prev, prev + n) of inputKeys / inputValues", evaluate it on the fly (the fused
// scan of hash_reduce_lds.hip) instead of launching it.  Returns false when the call must take the
// ordinary path (whatever was pending has been launched, in stream order).
bool fuse_pending_into_hash_reduce(int device, hipStream_t stream, const DimensionVector &in, const uint8_t *inValues,
                                   const DimensionVector &out, uint8_t *outValues, int valueBytes, int length, int aggFunc,
                                   int *groups) {
  if (!fuse_available()) return false;
  const bool forcedGlobal = global_table_forced();
  PendingQueue q;
  FusedPlanD plan;
  memset(&plan, 0, sizeof(plan));
  int nd = 0, prev = 0, n0 = 0;
  AggSpec a;
  {
    DeferLock lock(device);
    auto it = t_state->pending.find({device, stream});
    if (it == t_state->pending.end() || it->second.jobs.count == 0) return false;
    PendingQueue &pq = it->second;
    bool ok = !forcedGlobal;
    // dimension slots of 4, 2 or 1 bytes, in the vector's (descending width) order
    nd = in.NumDimsPerDimWidth[2] + in.NumDimsPerDimWidth[3] + in.NumDimsPerDimWidth[4];
    for (int k = 0; k < NUM_DIM_WIDTH; k++)
      ok = ok && (k >= 2 || in.NumDimsPerDimWidth[k] == 0) && out.NumDimsPerDimWidth[k] == in.NumDimsPerDimWidth[k];
    ok = ok && nd >= 1 && nd <= kFusedDims && pq.jobs.count == nd + 1;
    DimLayoutD L;
    memset(&L, 0, sizeof(L));
    if (ok) L = make_dim_layout(in.NumDimsPerDimWidth);
    prev = length - pq.n;
    ok = ok && prev >= 0 && pq.n > 0;
    if (ok) {
      try {
        a = make_agg_spec(aggFunc, valueBytes);
        ok = hash_reduce_lds_supported(a);
      } catch (std::exception &) {
        ok = false;
      }
    }
    // survivors: the whole batch (no filter ran) or what the journalled filters keep
    const FilterJournal *journal = nullptr;
    n0 = pq.n;
    if (ok && pq.idx) {
      auto j = t_state->journals.find(pq.idx);
      ok = j != t_state->journals.end() && j->second.valid && j->second.start == 0 && j->second.device == device;
      if (ok) {
        journal = &j->second;
        n0 = journal->n0;
      }
    }
    // every dimension column and the measure of rows [prev, prev + n) must be a pending sink
    const size_t cap = static_cast<size_t>(in.VectorCapacity);
    int dimJob[kFusedDims], measureJob = -1;
    for (int c = 0; c < kFusedDims; c++) dimJob[c] = -1;
    for (int k = 0; ok && k < pq.jobs.count; k++) {
      const SinkD &s = pq.jobs.s[k];
      ok = pq.colRows[k] >= static_cast<uint32_t>(n0);
      if (s.type == SINK_MEASURE) {
        ok = ok && measureJob < 0 && s.values == inValues + static_cast<size_t>(valueBytes) * prev && s.width == valueBytes &&
             s.agg == aggFunc && s.baseCounts == nullptr;
        measureJob = k;
      } else {
        int d = -1;
        for (int c = 0; c < nd; c++)
          if (s.values == in.DimValues + static_cast<size_t>(L.valueOff[c]) * cap + static_cast<size_t>(L.width[c]) * prev &&
              s.nulls == in.DimValues + static_cast<size_t>(L.valueBytes) * cap + cap * c + prev && s.width == L.width[c])
            d = c;
        ok = ok && s.type == SINK_DIM && d >= 0 && dimJob[d] < 0;
        if (ok) dimJob[d] = k;
      }
    }
    ok = ok && measureJob >= 0;
    for (int c = 0; ok && c < nd; c++) ok = dimJob[c] >= 0;
    if (ok) {
      auto column_of = [](const FastOperands &f) { return FusedColumn{f.vals, f.nulls, f.bitOff, static_cast<uint32_t>(f.step ? f.step : 4)}; };
      auto strip = [](FastOperands f) {
        f.vals = nullptr;
        f.nulls = nullptr;
        f.idx = nullptr;
        f.pad = 0;
        return f;
      };
      auto kind_of = [](int dtype) { return (dtype == Int32 || dtype == Int16 || dtype == Int8) ? K_I32 : (dtype == Uint32 || dtype == Uint16 || dtype == Uint8) ? K_U32 : K_F32; };
      for (int c = 0; c < nd; c++) {
        const int k = dimJob[c];
        plan.cols[c] = column_of(pq.jobs.f[k]);
        plan.dims[c].f = strip(pq.jobs.f[k]);
        plan.dims[c].col = c;
        plan.dims[c].outKind = kind_of(pq.jobs.s[k].dtype);
        plan.dimWidth[c] = L.width[c];
      }
      plan.cols[nd] = column_of(pq.jobs.f[measureJob]);
      plan.measure.f = strip(pq.jobs.f[measureJob]);
      plan.measure.col = nd;
      plan.measure.outKind = kind_of(pq.jobs.s[measureJob].dtype);
      plan.measureDtype = pq.jobs.s[measureJob].dtype;
      plan.measureWidth = valueBytes;
      plan.identity = pq.jobs.s[measureJob].identity;
      plan.numCols = nd + 1;
      ok = !(valueBytes == 8 && plan.identity != 0);  // records carry 4 bytes: a null must widen to the identity
      if (journal) {
        ok = ok && journal->filters.size() <= static_cast<size_t>(kFusedFilters);
        for (size_t k = 0; ok && k < journal->filters.size(); k++) {
          const FastOperands &f = journal->filters[k];
          ok = journal->colRows[k] >= static_cast<uint32_t>(n0);
          int col = -1;
          for (int c = 0; c < plan.numCols; c++)
            if (plan.cols[c].vals == f.vals && plan.cols[c].nulls == f.nulls && plan.cols[c].bitOff == f.bitOff &&
                plan.cols[c].step == static_cast<uint32_t>(f.step ? f.step : 4))
              col = c;
          if (col < 0 && plan.numCols < kFusedCols) {
            col = plan.numCols++;
            plan.cols[col] = column_of(f);
          }
          ok = ok && col >= 0;
          plan.filters[k].f = strip(f);
          plan.filters[k].col = col;
          plan.filters[k].outKind = K_BOOL;
        }
        plan.numFilters = static_cast<int>(journal->filters.size());
      }
      // the precompiled generic scan holds nd + 2 column slots; a narrow plan only ever runs on generated kernels (kFusedCols)
      ok = ok && (plan.numCols <= nd + 2 || fused_plan_narrow(plan, nd));
    }
    if (!ok) {  // the ordinary path: the pending work runs now, ahead of this call on the same stream
      launch_queue(stream, pq, /*inOrder=*/true);
      g_releaseHeld(device, hold_tag(stream));
      return false;
    }
    q = pq;  // taken out of the queue: nobody else launches it
    pq.jobs.count = 0;
    pq.reads.clear();
    pq.writes.clear();
    pq.keep.clear();  // (the copy holds the decoded columns from here on: left in place they piled up, two per batch, for the life
                      // of the stream — 86 GB after 143 archive batches — and every later copy of the queue took the pile along)
    pq.overWait = false;
    if (q.idx) {  // (decoded columns only the filters read live as long as the consumed queue does: the scan is launched below)
      auto jk = t_state->journals.find(q.idx);
      if (jk != t_state->journals.end()) q.keep.insert(q.keep.end(), jk->second.keep.begin(), jk->second.keep.end());
      t_state->journals.erase(q.idx);
    }
  }
  DimensionVector prevKeys = in;
  const int result = fused_hash_reduce_run(device, plan, n0, prevKeys, inValues, prev, out, outValues, a, stream);
  DeferLock lock(device);
  if (result < 0) {  // a partition region overflowed, or the call was declined: materialise the inputs after all
    launch_queue(stream, q, /*inOrder=*/true);
    lock.unlock();
    g_releaseHeld(device, hold_tag(stream));
    return false;
  }
  t_state->limbo[{device, stream}] = q;  // launchable until the next batch begins (begin_batch)
  *groups = result;
  return true;
}

// ---- Sort + Reduce over pending transforms (sort_reduce_fused.hip) ---------------------------------------------------------
namespace {
// The queue `pq` is "the dimension columns [and the measure] of rows [prev, prev + pq.n) of `in`": which job writes which
// dimension, which one the measure (-1: none queued).  Caller holds the device's DeferLock.
bool match_sort_jobs(const PendingQueue &pq, const DimensionVector &in, const DimLayoutD &L, int prev, int (&dimJob)[kFusedDims], int *measureJob) {
  const int nd = L.numDims;
  const size_t cap = static_cast<size_t>(in.VectorCapacity);
  if (nd < 1 || nd > kFusedDims || in.NumDimsPerDimWidth[0] || in.NumDimsPerDimWidth[1]) return false;
  if (pq.jobs.count != nd && pq.jobs.count != nd + 1) return false;
  for (int c = 0; c < kFusedDims; c++) dimJob[c] = -1;
  *measureJob = -1;
  for (int k = 0; k < pq.jobs.count; k++) {
    const SinkD &s = pq.jobs.s[k];
    if (s.type == SINK_MEASURE) {
      if (*measureJob >= 0 || s.baseCounts) return false;
      *measureJob = k;
      continue;
    }
    int d = -1;
    for (int c = 0; c < nd; c++)
      if (s.values == in.DimValues + static_cast<size_t>(L.valueOff[c]) * cap + static_cast<size_t>(L.width[c]) * prev &&
          s.nulls == in.DimValues + static_cast<size_t>(L.valueBytes) * cap + cap * c + prev && s.width == L.width[c])
        d = c;
    if (s.type != SINK_DIM || d < 0 || dimJob[d] >= 0) return false;
    dimJob[d] = k;
  }
  for (int c = 0; c < nd; c++)
    if (dimJob[c] < 0) return false;
  return true;
}
bool same_vector(const DimensionVector &a, const DimensionVector &b) {
  return a.DimValues == b.DimValues && a.HashValues == b.HashValues && a.IndexVector == b.IndexVector && a.VectorCapacity == b.VectorCapacity &&
         memcmp(a.NumDimsPerDimWidth, b.NumDimsPerDimWidth, sizeof(a.NumDimsPerDimWidth)) == 0;
}
}  // namespace

// caller holds the device's DeferLock.  A lazily defined Sort has been consumed by a Reduce whose kernels wrote `result` groups:
// the hash vector, the input's index vector and the output's index vector stay DEFINED (what Sort and Reduce would have left
// there); whoever reads them replays the sequence (materialize_sort)
static void enter_reduced_sort(int device, hipStream_t stream, PendingSort ps, const DimensionVector &in, uint8_t *inValues,
                               const DimensionVector &out, uint8_t *outValues, int valueBytes, int length, int aggFunc, int result) {
  ps.reduced = true;
  ps.outKeys = out;
  ps.inValues = inValues;
  ps.outValues = outValues;
  ps.valueBytes = valueBytes;
  ps.aggFunc = aggFunc;
  ps.groups = result;
  t_state->sorts[in.IndexVector] = ps;
  PendingIota io{device, stream, 0, length};
  io.consumed = true;
  io.sorted = true;
  t_state->iotas[in.IndexVector] = io;
  uint8_t *hv = reinterpret_cast<uint8_t *>(in.HashValues);
  retire_fills(device, ByteRange{hv, hv + 8ull * static_cast<size_t>(length)}, false);
  PendingFill marker{device, stream, 8ull * static_cast<size_t>(length), 0, 8, false};
  marker.sortIdx = in.IndexVector;
  t_state->fills[hv] = marker;
  if (result > 0) {
    uint8_t *oi = reinterpret_cast<uint8_t *>(out.IndexVector);
    retire_fills(device, ByteRange{oi, oi + 4ull * static_cast<size_t>(result)}, false);
    PendingFill om{device, stream, 4ull * static_cast<size_t>(result), 0, 4, false};
    om.sortIdx = in.IndexVector;
    t_state->fills[oi] = om;
  }
}

// Sort(keys, length) when the rows [length - n, length) of `keys` are what this stream's pending transforms would write and
// the index vector is still the iota InitIndexVector defined: nothing is hashed or sorted — the hash vector (a marker among
// the lazy fills) and the index vector (its iota entry, `sorted`) are DEFINED as what Sort leaves.  Reduce consumes the
// definition (below); every other reader runs it (materialize_sort).
bool define_lazy_sort(int device, hipStream_t stream, const DimensionVector &keys, int length) {
  if (!fuse_available() || !fused_sort_reduce_enabled() || length <= 0 || !keys.DimValues || !keys.HashValues || !keys.IndexVector ||
      keys.VectorCapacity < length)
    return false;
  DimLayoutD L;
  try {
    L = make_dim_layout(keys.NumDimsPerDimWidth);
  } catch (std::exception &) {
    return false;
  }
  DeferLock lock(device);
  auto io = t_state->iotas.find(keys.IndexVector);
  if (io == t_state->iotas.end() || io->second.device != device || io->second.start != 0 || io->second.n != length || io->second.consumed ||
      io->second.sorted)
    return false;
  auto it = t_state->pending.find({device, stream});
  if (it == t_state->pending.end() || it->second.jobs.count == 0) return false;
  const PendingQueue &pq = it->second;
  const int prev = length - pq.n;
  int dimJob[kFusedDims], measureJob = -1;
  if (pq.n <= 0 || prev < 0 || !match_sort_jobs(pq, keys, L, prev, dimJob, &measureJob)) return false;
  if (pq.idx) {  // the survivors must be re-derivable from the filter journal
    auto j = t_state->journals.find(pq.idx);
    if (j == t_state->journals.end() || !j->second.valid || j->second.start != 0 || j->second.device != device) return false;
  }
  uint8_t *hv = reinterpret_cast<uint8_t *>(keys.HashValues);
  retire_fills(device, ByteRange{hv, hv + 8ull * static_cast<size_t>(length)}, false);
  PendingSort ps{};
  ps.device = device;
  ps.stream = stream;
  ps.keys = keys;
  ps.length = length;
  t_state->sorts[keys.IndexVector] = ps;
  io->second.consumed = true;
  io->second.sorted = true;
  PendingFill marker{device, stream, 8ull * static_cast<size_t>(length), 0, 8, false};
  marker.sortIdx = keys.IndexVector;
  t_state->fills[hv] = marker;
  return true;
}

// ---- Sort + Reduce over materialised vectors -----------------------------------------------------------------------------
// The batch's dimension rows were written by kernels (a joined column, a generic expression, an eager host): Sort still need
// not order ROWS for Reduce to order GROUPS (fused_sort_reduce_vectors).  Sort asks before its flush (the index vector's lazy
// iota is marked consumed: the flush does not write it) ...
namespace {
bool vector_sort_enabled() {
  static EnvSwitch<bool> on("ARES_SORT_VECTORS", [](const char *e) { return !(e && e[0] == '0'); });
  return on.get();
}
bool vector_sort_layout(const DimensionVector &keys) {
  int nd = 0;
  for (int w = 0; w < NUM_DIM_WIDTH; w++) nd += keys.NumDimsPerDimWidth[w];
  return nd >= 1 && nd <= kFusedDims && !keys.NumDimsPerDimWidth[0] && !keys.NumDimsPerDimWidth[1];  // slots of 4 / 2 / 1 bytes
}
}  // namespace

bool lazy_vector_sort_candidate(int device, const DimensionVector &keys, int length) {
  static const bool trace = getenv("ARES_HR_TRACE") != nullptr;  // diagnostics
  if (trace)
    fprintf(stderr, "lazy_vector_sort_candidate: rows %d fuse %d enabled %d switch %d layout %d capacity %d\n", length, fuse_available() ? 1 : 0,
            fused_sort_reduce_enabled() ? 1 : 0, vector_sort_enabled() ? 1 : 0, vector_sort_layout(keys) ? 1 : 0, keys.VectorCapacity);
  if (!fuse_available() || !fused_sort_reduce_enabled() || !vector_sort_enabled() || length <= 0 || !keys.DimValues || !keys.HashValues ||
      !keys.IndexVector || keys.VectorCapacity < length || !vector_sort_layout(keys))
    return false;
  DeferLock lock(device);
  auto io = t_state->iotas.find(keys.IndexVector);
  if (io == t_state->iotas.end() || io->second.device != device || io->second.start != 0 || io->second.n != length || io->second.consumed ||
      io->second.sorted) {
    if (trace) fprintf(stderr, "lazy_vector_sort_candidate: index vector is %s\n", io == t_state->iotas.end() ? "not a lazy iota" : "a lazy iota of another kind");
    return false;
  }
  io->second.consumed = true;
  return true;
}

// ... and defines itself after it (every pending writer of the rows has been launched).  false: the index vector is written,
// the caller sorts.
bool define_lazy_sort_vectors(int device, hipStream_t stream, const DimensionVector &keys, int length) {
  {
    DeferLock lock(device);
    auto io = t_state->iotas.find(keys.IndexVector);
    if (io != t_state->iotas.end() && io->second.device == device && io->second.start == 0 && io->second.n == length && io->second.consumed &&
        !io->second.sorted) {
      uint8_t *hv = reinterpret_cast<uint8_t *>(keys.HashValues);
      retire_fills(device, ByteRange{hv, hv + 8ull * static_cast<size_t>(length)}, false);
      PendingSort ps{};
      ps.device = device;
      ps.stream = stream;
      ps.keys = keys;
      ps.length = length;
      ps.fromVectors = true;
      t_state->sorts[keys.IndexVector] = ps;
      io->second.sorted = true;
      PendingFill marker{device, stream, 8ull * static_cast<size_t>(length), 0, 8, false};
      marker.sortIdx = keys.IndexVector;
      t_state->fills[hv] = marker;
      return true;
    }
  }
  materialize_index_vector(device, keys.IndexVector);
  return false;
}

// Reduce over a Sort defined that way.  Caller: fuse_pending_into_sort_reduce.
static bool reduce_lazy_vector_sort(int device, hipStream_t stream, const DimensionVector &in, uint8_t *inValues, const DimensionVector &out,
                                    uint8_t *outValues, int valueBytes, int length, int aggFunc, int *groups) {
  PendingSort ps{};
  AggSpec a{};
  {
    DeferLock lock(device);
    auto st = t_state->sorts.find(in.IndexVector);
    if (st == t_state->sorts.end()) return false;
    ps = st->second;
    bool ok = !ps.reduced && ps.fromVectors && ps.device == device && ps.stream == stream && ps.length == length && same_vector(ps.keys, in) &&
              inValues && out.DimValues && out.IndexVector && outValues && out.VectorCapacity >= length && in.VectorCapacity >= length &&
              memcmp(out.NumDimsPerDimWidth, in.NumDimsPerDimWidth, sizeof(in.NumDimsPerDimWidth)) == 0 && (valueBytes == 4 || valueBytes == 8);
    if (ok) {
      try {
        a = make_agg_spec(aggFunc, valueBytes);
        ok = fused_sort_reduce_supported(a);
      } catch (std::exception &) {
        ok = false;
      }
    }
    if (!ok) {
      materialize_sort(in.IndexVector);
      return false;
    }
    drop_sort(in.IndexVector);  // (while the kernels run nothing is defined; the reduced state is entered below)
  }
  size_t rowBytes = 0;
  for (int w = 0; w < NUM_DIM_WIDTH; w++) rowBytes += static_cast<size_t>(in.NumDimsPerDimWidth[w]) * ((1u << (NUM_DIM_WIDTH - 1 - w)) + 1);
  retire_fills_for_write(device, outValues, static_cast<size_t>(valueBytes) * length);
  retire_fills_for_write(device, out.IndexVector, 4ull * static_cast<size_t>(length));
  grouped_note_write(device, out);
  grouped_note_write(device, outValues, static_cast<size_t>(valueBytes) * length);
  drop_skipped_outputs(device, out.DimValues, rowBytes * static_cast<size_t>(in.VectorCapacity), outValues, static_cast<size_t>(valueBytes) * length);
  // every row is read by this call's kernels: whatever still only defines one is written
  launch_pending_writers(device, in.DimValues, rowBytes * static_cast<size_t>(in.VectorCapacity));
  launch_pending_writers(device, inValues, static_cast<size_t>(valueBytes) * length);
  materialize_fills_for_read(device, in.DimValues, rowBytes * static_cast<size_t>(in.VectorCapacity));
  materialize_fills_for_read(device, inValues, static_cast<size_t>(valueBytes) * length);
  int result;
  try {
    result = fused_sort_reduce_vectors(device, length, in, inValues, out, outValues, a, stream);
  } catch (...) {
    result = -1;
  }
  if (result < 0) {  // declined (a kernel still being compiled, too many rows), or a table overflowed: the real thing
    {
      DeferLock lock(device);
      launch_init_index(in.IndexVector, 0, length, stream);
    }
    sort_keys_now(in, length, stream);
    return false;
  }
  mem_note_dim_rows(device, out, 0, static_cast<size_t>(result));
  mem_note_write(device, outValues, static_cast<size_t>(valueBytes) * static_cast<size_t>(result));
  DeferLock lock(device);
  enter_reduced_sort(device, stream, ps, in, inValues, out, outValues, valueBytes, length, aggFunc, result);
  *groups = result;
  return true;
}

bool fuse_pending_into_sort_reduce(int device, hipStream_t stream, const DimensionVector &in, uint8_t *inValues, const DimensionVector &out,
                                   uint8_t *outValues, int valueBytes, int length, int aggFunc, int *groups) {
  if (!fuse_available() || !in.IndexVector) return false;
  {
    bool vectors = false;
    {
      DeferLock lock(device);
      auto st = t_state->sorts.find(in.IndexVector);
      vectors = st != t_state->sorts.end() && st->second.fromVectors && !st->second.reduced;
    }
    if (vectors) return reduce_lazy_vector_sort(device, stream, in, inValues, out, outValues, valueBytes, length, aggFunc, groups);
  }
  PendingQueue q;
  FusedPlanD plan;
  memset(&plan, 0, sizeof(plan));
  PendingSort ps{};
  AggSpec a{};
  int nd = 0, prev = 0, n0 = 0;
  bool constMeasure = false;
  uint64_t constBits = 0;
  PendingFill fill{};
  uint8_t *fillAt = nullptr;
  {
    DeferLock lock(device);
    auto st = t_state->sorts.find(in.IndexVector);
    if (st == t_state->sorts.end()) return false;  // nothing lazy about this Sort
    ps = st->second;
    bool ok = !ps.reduced && ps.device == device && ps.stream == stream && ps.length == length && same_vector(ps.keys, in) && inValues &&
              out.DimValues && out.IndexVector && outValues && out.VectorCapacity >= length &&
              memcmp(out.NumDimsPerDimWidth, in.NumDimsPerDimWidth, sizeof(in.NumDimsPerDimWidth)) == 0;
    DimLayoutD L;
    memset(&L, 0, sizeof(L));
    if (ok) L = make_dim_layout(in.NumDimsPerDimWidth);
    nd = L.numDims;
    auto it = t_state->pending.find({device, stream});
    ok = ok && it != t_state->pending.end() && it->second.jobs.count > 0;
    int dimJob[kFusedDims], measureJob = -1;
    if (ok) {
      prev = length - it->second.n;
      ok = it->second.n > 0 && prev >= 0 && match_sort_jobs(it->second, in, L, prev, dimJob, &measureJob);
    }
    if (ok) {
      try {
        a = make_agg_spec(aggFunc, valueBytes);
        ok = fused_sort_reduce_supported(a);
      } catch (std::exception &) {
        ok = false;
      }
    }
    const FilterJournal *journal = nullptr;
    if (ok) {
      const PendingQueue &pq = it->second;
      n0 = pq.n;
      if (pq.idx) {
        auto j = t_state->journals.find(pq.idx);
        ok = j != t_state->journals.end() && j->second.valid && j->second.start == 0 && j->second.device == device;
        if (ok) {
          journal = &j->second;
          n0 = journal->n0;
        }
      }
    }
    if (ok) {
      const PendingQueue &pq = it->second;
      for (int k = 0; ok && k < pq.jobs.count; k++) ok = pq.colRows[k] >= static_cast<uint32_t>(n0);
      uint8_t *measureRows = inValues + static_cast<size_t>(valueBytes) * prev;
      if (ok && measureJob >= 0) {
        const SinkD &sm = pq.jobs.s[measureJob];
        ok = sm.values == measureRows && sm.width == valueBytes && sm.agg == aggFunc && !(valueBytes == 8 && sm.identity != 0);
      } else if (ok) {  // a constant measure (COUNT(*)): the rows are a lazy fill
        auto f = t_state->fills.find(measureRows);
        ok = f != t_state->fills.end() && f->second.device == device && !f->second.sortIdx && f->second.unit == valueBytes &&
             f->second.bytes == static_cast<size_t>(valueBytes) * pq.n;
        if (ok) {
          constMeasure = true;
          fill = f->second;
          fillAt = f->first;
          constBits = fill.pattern;
        }
      }
    }
    if (ok) {
      const PendingQueue &pq = it->second;
      auto column_of = [](const FastOperands &f) { return FusedColumn{f.vals, f.nulls, f.bitOff, static_cast<uint32_t>(f.step ? f.step : 4)}; };
      auto strip = [](FastOperands f) {
        f.vals = nullptr;
        f.nulls = nullptr;
        f.idx = nullptr;
        f.pad = 0;
        return f;
      };
      auto kind_of = [](int dtype) { return (dtype == Int32 || dtype == Int16 || dtype == Int8) ? K_I32 : (dtype == Uint32 || dtype == Uint16 || dtype == Uint8) ? K_U32 : K_F32; };
      for (int c = 0; c < nd; c++) {
        const int k = dimJob[c];
        plan.cols[c] = column_of(pq.jobs.f[k]);
        plan.dims[c].f = strip(pq.jobs.f[k]);
        plan.dims[c].col = c;
        plan.dims[c].outKind = kind_of(pq.jobs.s[k].dtype);
        plan.dimWidth[c] = L.width[c];
      }
      plan.numCols = nd;
      plan.measureWidth = valueBytes;
      if (measureJob >= 0) {
        plan.cols[nd] = column_of(pq.jobs.f[measureJob]);
        plan.measure.f = strip(pq.jobs.f[measureJob]);
        plan.measure.col = nd;
        plan.measure.outKind = kind_of(pq.jobs.s[measureJob].dtype);
        plan.measureDtype = pq.jobs.s[measureJob].dtype;
        plan.identity = pq.jobs.s[measureJob].identity;
        plan.numCols = nd + 1;
      } else {
        plan.measure.col = -1;
        plan.measure.f.bbits = static_cast<uint32_t>(constBits);  // (what the scan's records carry; the merge takes constBits)
        plan.measureDtype = valueBytes == 8 ? Int64 : (a.vtype == V_I32 ? Int32 : Uint32);
      }
      if (journal) {
        ok = journal->filters.size() <= static_cast<size_t>(kFusedFilters);
        for (size_t k = 0; ok && k < journal->filters.size(); k++) {
          const FastOperands &f = journal->filters[k];
          ok = journal->colRows[k] >= static_cast<uint32_t>(n0);
          int col = -1;
          for (int c = 0; c < plan.numCols; c++)
            if (plan.cols[c].vals == f.vals && plan.cols[c].nulls == f.nulls && plan.cols[c].bitOff == f.bitOff &&
                plan.cols[c].step == static_cast<uint32_t>(f.step ? f.step : 4))
              col = c;
          if (col < 0 && plan.numCols < kFusedCols) {
            col = plan.numCols++;
            plan.cols[col] = column_of(f);
          }
          ok = ok && col >= 0;
          plan.filters[k].f = strip(f);
          plan.filters[k].col = col;
          plan.filters[k].outKind = K_BOOL;
        }
        plan.numFilters = static_cast<int>(journal->filters.size());
      }
    }
    if (!ok) {  // not this case after all: Sort runs (and the ordinary Reduce follows)
      materialize_sort(in.IndexVector);
      return false;
    }
    PendingQueue &pq = it->second;
    q = pq;  // taken out of the queue: nobody else launches it
    pq.jobs.count = 0;
    pq.reads.clear();
    pq.writes.clear();
    pq.keep.clear();  // (the copy holds the decoded columns from here on: left in place they piled up, two per batch, for the life
                      // of the stream — 86 GB after 143 archive batches — and every later copy of the queue took the pile along)
    pq.overWait = false;
    if (q.idx) {  // (decoded columns only the filters read live as long as the consumed queue does: the scan is launched below)
      auto jk = t_state->journals.find(q.idx);
      if (jk != t_state->journals.end()) q.keep.insert(q.keep.end(), jk->second.keep.begin(), jk->second.keep.end());
      t_state->journals.erase(q.idx);
    }
    if (constMeasure) t_state->fills.erase(fillAt);
    drop_sort(in.IndexVector);  // (while the kernels run nothing is defined; the reduced state is entered below)
  }
  // the outputs are about to be rewritten, the previous result is read by kernels of this call
  retire_fills_for_write(device, outValues, static_cast<size_t>(valueBytes) * length);
  retire_fills_for_write(device, out.IndexVector, 4ull * static_cast<size_t>(length));
  grouped_note_write(device, out);
  grouped_note_write(device, outValues, static_cast<size_t>(valueBytes) * length);
  {
    size_t rowBytes = 0;
    for (int w = 0; w < NUM_DIM_WIDTH; w++) rowBytes += static_cast<size_t>(out.NumDimsPerDimWidth[w]) * ((1u << (NUM_DIM_WIDTH - 1 - w)) + 1);
    drop_skipped_outputs(device, out.DimValues, rowBytes * static_cast<size_t>(in.VectorCapacity), outValues, static_cast<size_t>(valueBytes) * length);
    if (prev > 0) {  // rows [0, prev) are read by this call's kernels: whatever still only defines them is written (no flush:
                     // this batch's own lazy compaction must stay lazy)
      launch_pending_writers(device, in.DimValues, rowBytes * static_cast<size_t>(in.VectorCapacity));
      launch_pending_writers(device, inValues, static_cast<size_t>(valueBytes) * prev);
      materialize_fills_for_read(device, in.DimValues, rowBytes * static_cast<size_t>(in.VectorCapacity));
      materialize_fills_for_read(device, inValues, static_cast<size_t>(valueBytes) * prev);
    }
  }
  int result;
  try {
    result = fused_sort_reduce_run(device, plan, nd, constMeasure, constBits, n0, in, inValues, prev, out, outValues, a, stream);
  } catch (...) {
    result = -1;
  }
  if (result < 0) {  // declined, or a partition overflowed: the batch's rows are written — transforms, the constant rows — ...
    {
      DeferLock lock(device);
      launch_queue(stream, q, /*inOrder=*/true);
      if (constMeasure) launch_fill(fillAt, fill);
    }
    g_releaseHeld(device, hold_tag(stream));
    // ... and the groups are ordered by row hash over the rows that exist now (the wide layout takes results the 512 tables
    // of the scan-fed path do not: millions of groups per batch) ...
    int wide = kFusedUnavailable;
    if (vector_sort_enabled() && vector_sort_layout(in)) {
      try {
        wide = fused_sort_reduce_vectors(device, length, in, inValues, out, outValues, a, stream);
      } catch (...) {
        wide = -1;
      }
    }
    if (wide >= 0) {
      mem_note_dim_rows(device, out, 0, static_cast<size_t>(wide));
      mem_note_write(device, outValues, static_cast<size_t>(valueBytes) * static_cast<size_t>(wide));
      DeferLock lock(device);
      ps.fromVectors = true;
      enter_reduced_sort(device, stream, ps, in, inValues, out, outValues, valueBytes, length, aggFunc, wide);
      *groups = wide;
      return true;
    }
    // ... or the real thing: InitIndexVector, Sort (the ordinary Reduce follows)
    {
      DeferLock lock(device);
      launch_init_index(in.IndexVector, 0, length, stream);
    }
    sort_keys_now(in, length, stream);
    return false;
  }
  mem_note_dim_rows(device, out, 0, static_cast<size_t>(result));
  mem_note_write(device, outValues, static_cast<size_t>(valueBytes) * static_cast<size_t>(result));
  DeferLock lock(device);
  q.sortIdx = in.IndexVector;
  t_state->limbo[{device, stream}] = q;  // launchable until the next batch begins (begin_batch)
  ps.constMeasure = constMeasure;
  ps.fill = fill;
  ps.fillAt = fillAt;
  enter_reduced_sort(device, stream, ps, in, inValues, out, outValues, valueBytes, length, aggFunc, result);
  *groups = result;
  return true;
}

// tile counts of an existing predicate vector, in filter_pred_kernel's tile geometry
__global__ __launch_bounds__(kBlock) void pred_count_kernel(const uint8_t *pred, uint32_t *tileCounts, int pad, int n,
                                                            int numTiles) {
  const int lane = threadIdx.x & 63;
  for (int tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
    const int64_t tq = static_cast<int64_t>(tile) * (kBlock * kPQ);
    uint32_t count = 0;
#pragma unroll
    for (int q = 0; q < kPQ; q++) {
      const int64_t i0 = (tq + threadIdx.x + static_cast<int64_t>(q) * kBlock) * 4 - pad;
      if (i0 >= 0 && i0 + 3 < n) {
        const uint32_t pb = *reinterpret_cast<const uint32_t *>(pred + i0);
        count += ((pb & 0xFFu) ? 1u : 0u) + ((pb & 0xFF00u) ? 1u : 0u) + ((pb & 0xFF0000u) ? 1u : 0u) +
                 ((pb & 0xFF000000u) ? 1u : 0u);
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (i0 + j >= 0 && i0 + j < n && pred[i0 + j]) count++;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) count += __shfl_xor(count, off);
    if (lane == 0 && count) atomicAdd(tileCounts + tile, count);
  }
}

// Stable in-place compaction of the index vector and of every RecordID vector by an existing
// predicate vector (keep = byte != 0): the tail of the two-phase filter, for callers that compute
// their predicate elsewhere (geo intersection).
int compact_by_predicate(const uint8_t *pred, uint32_t *indexVector, RecordID **recordIDVectors, int numForeignTables,
                         int n, hipStream_t stream) {
  if (n <= 0) return 0;
  {
    const int device = current_device();
    mem_note_write(device, indexVector, 4ull * static_cast<size_t>(n));
    for (int t = 0; t < numForeignTables; t++) mem_note_write(device, recordIDVectors[t], 8ull * static_cast<size_t>(n));
  }
  const int pad = static_cast<int>(reinterpret_cast<uintptr_t>(pred) & 3);
  const int64_t numQuads = (static_cast<int64_t>(n) + pad + 3) / 4;
  const int tiles = static_cast<int>((numQuads + kBlock * kPQ - 1) / (kBlock * kPQ));
  const int passes = 1 + numForeignTables;
  const size_t head = 64;
  const size_t words = static_cast<size_t>(tiles) * (2 + passes) + 1;
  StreamBuffer wsBuf(head + 4 * words, stream);
  uint32_t *w = wsBuf.as<uint32_t>();
  uint32_t *total = w, *error = w + 1;
  unsigned int *tickets = w + 2;
  uint32_t *tileCounts = w + 16, *tileOffsets = tileCounts + tiles, *loaded = tileOffsets + tiles + 1;
  hip_check(hipMemsetAsync(w, 0, head + 4 * words, stream), "hipMemsetAsync");
  ARES_LAUNCH("pred_count_kernel", pred_count_kernel, capped_grid(tiles, 256 * 16), kBlock, stream, pred, tileCounts, pad, n, tiles);
  ARES_LAUNCH("filter_scan_kernel", filter_scan_kernel, 1, 1024, stream, tileCounts, tileOffsets, tiles, total);
  const int cgrid = capped_grid((tiles + kTilesPerTicket - 1) / kTilesPerTicket, 256 * 8);
  for (int pass = 0; pass < passes; pass++) {
    CompactWorkspace cw;
    cw.ticket = tickets + pass;
    cw.error = error;
    cw.tileOffsets = tileOffsets;
    cw.loaded = loaded + static_cast<size_t>(tiles) * pass;
    if (pass == 0)
      ARES_LAUNCH("filter_compact_kernel", (filter_compact_kernel<uint32_t, false>), cgrid, kBlock, stream, pred, indexVector, 0u,
                  pad, cw, n, tiles);
    else
      ARES_LAUNCH("filter_compact_kernel<rid>", (filter_compact_kernel<uint64_t, false>), cgrid, kBlock, stream, pred,
                  reinterpret_cast<uint64_t *>(recordIDVectors[pass - 1]), 0u, pad, cw, n, tiles);
  }
  uint32_t result[2] = {0, 0};  // {survivors, error}
  read_back_u32(total, result, 2, stream);
  if (result[1]) throw AlgorithmError("ERROR: filter: compaction wait timed out");
  return static_cast<int>(result[0]);
}

}  // namespace ares

using namespace ares;

extern "C" {

size_t AresStreamEvents(int device, void *stream) {
  size_t n = profiler_stream_events(reinterpret_cast<hipStream_t>(stream));
  if (device < 0 || device >= kMaxDevices) return n;
  DeferLock lock(device);
  for (const ErrorCheck &c : t_state->errorChecks) n += c.device == device && c.stream == reinterpret_cast<hipStream_t>(stream);
  return n;
}

void AresTempStats(size_t *handedOutBytes, size_t *cachedBytes) { temp_stats(handedOutBytes, cachedBytes); }

void AresFlushDeferred(int device) {
  int current = 0;
  if (hipGetDevice(&current) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  try {
    if (current != device) hip_check(hipSetDevice(device), "hipSetDevice");
    flush_deferred(device);
    if (current != device) (void)hipSetDevice(current);
  } catch (std::exception &e) {
    fprintf(stderr, "Exception happened when flushing deferred transforms: %s\n", e.what());
  }
}

CGoCallResHandle InitIndexVector(uint32_t *indexVector, uint32_t start, int indexVectorLength, void *cudaStream,
                                 int device) {
  ARES_ABI_BEGIN_NOFLUSH(device)
  hipStream_t stream = reinterpret_cast<hipStream_t>(cudaStream);
  begin_batch(device, stream, indexVector, start, indexVectorLength);
  if (!defer_iota(device, stream, indexVector, start, indexVectorLength)) {
    flush_deferred(device);
    DeferLock lock(device);
    launch_init_index(indexVector, start, indexVectorLength, stream);
  }
  ARES_ABI_END("InitIndexVector")
}

CGoCallResHandle UnaryTransform(InputVector input, OutputVector output, uint32_t *indexVector,
                                int indexVectorLength, uint32_t *baseCounts, uint32_t startCount,
                                enum UnaryFunctorType functorType, void *cudaStream, int device) {
  ARES_ABI_BEGIN_NOFLUSH(device)
  resHandle.res = int_result(run_transform(&input, 1, output, indexVector, indexVectorLength, baseCounts,
                                           startCount, functorType, reinterpret_cast<hipStream_t>(cudaStream), device));
  ARES_ABI_END("UnaryTransform")
}

CGoCallResHandle BinaryTransform(InputVector lhs, InputVector rhs, OutputVector output, uint32_t *indexVector,
                                 int indexVectorLength, uint32_t *baseCounts, uint32_t startCount,
                                 enum BinaryFunctorType functorType, void *cudaStream, int device) {
  ARES_ABI_BEGIN_NOFLUSH(device)
  InputVector ins[2] = {lhs, rhs};
  resHandle.res = int_result(run_transform(ins, 2, output, indexVector, indexVectorLength, baseCounts, startCount,
                                           functorType, reinterpret_cast<hipStream_t>(cudaStream), device));
  ARES_ABI_END("BinaryTransform")
}

CGoCallResHandle UnaryFilter(InputVector input, uint32_t *indexVector, uint8_t *predicateVector,
                             int indexVectorLength, RecordID **recordIDVectors, int numForeignTables,
                             uint32_t *baseCounts, uint32_t startCount, enum UnaryFunctorType functorType,
                             void *cudaStream, int device) {
  ARES_ABI_BEGIN_NOFLUSH(device)
  resHandle.res = int_result(run_filter(&input, 1, indexVector, predicateVector, indexVectorLength, recordIDVectors,
                                        numForeignTables, baseCounts, startCount, functorType,
                                        reinterpret_cast<hipStream_t>(cudaStream), device));
  ARES_ABI_END("UnaryFilter")
}

CGoCallResHandle BinaryFilter(InputVector lhs, InputVector rhs, uint32_t *indexVector, uint8_t *predicateVector,
                              int indexVectorLength, RecordID **recordIDVectors, int numForeignTables,
                              uint32_t *baseCounts, uint32_t startCount, enum BinaryFunctorType functorType,
                              void *cudaStream, int device) {
  ARES_ABI_BEGIN_NOFLUSH(device)
  InputVector ins[2] = {lhs, rhs};
  resHandle.res = int_result(run_filter(ins, 2, indexVector, predicateVector, indexVectorLength, recordIDVectors,
                                        numForeignTables, baseCounts, startCount, functorType,
                                        reinterpret_cast<hipStream_t>(cudaStream), device));
  ARES_ABI_END("BinaryFilter")
}

}  // extern "C"
