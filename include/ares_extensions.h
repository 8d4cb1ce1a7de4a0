/* ares_extensions.h — entry points of the MI355X libalgorithm.so that are NOT part of the
 * reference ABI.  A host that only knows query/time_series_aggregate.h never needs them.
 */
#ifndef ARES_EXTENSIONS_H_
#define ARES_EXTENSIONS_H_

#include <stddef.h>
#include <stdint.h>

#include "ares_algorithm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-kernel timing with HIP events recorded on each launch's own stream.  Enable(1) clears the
 * log and starts recording, Enable(0) stops.  Report() resolves the events (the caller has already
 * synchronised its streams) and writes one "kernel launches total_ms" line per kernel name into
 * buf; it returns the buffer size needed.  The Go host's counterpart is the per-stage wall timing
 * of query/stats.go:160-169, which must synchronise the stream after every stage instead. */
void AresProfilerEnable(int on);
size_t AresProfilerReport(char *buf, size_t len);

/* Behaviour switches of libalgorithm.so that are read from the environment (ARES_HASH_REDUCE=global,
 * ARES_GROUPED=0, ARES_LEAN_MIN_GROUPS=n) are parsed once and kept; AresReloadEnv() makes every one of them
 * read its variable again on next use.  For tests that flip a switch inside one process. */
void AresReloadEnv(void);

/* The scan and merge kernels of a high- or low-cardinality HashReduce are compiled for the query's SHAPE at run time
 * (hiprtc; comparison and + - x constants are kernel arguments, so a new time range reuses the kernel) on a background
 * thread — a query never waits for the compiler, it runs the generic kernels until its kernels are loaded.
 * AresRtcWait() blocks until nothing is being compiled and returns the number of kernels in the cache; counters
 * (may be NULL) receives {hiprtc compilations, code objects loaded from the on-disk cache, kernels evicted}.
 * Environment: ARES_RTC=0 (off), ARES_RTC_ASYNC=0 (compile on the calling thread), ARES_RTC_CACHE_DIR (on-disk
 * cache of code objects; default ~/.cache/aresdb_amd/rtc; "off" disables), ARES_RTC_CACHE_ENTRIES (default 256). */
size_t AresRtcWait(long *counters);

/* Cross-call fusion inside the unchanged ABI.  Root transforms of the hot shape (a 4-byte column,
 * optionally combined with a constant, written to a dimension vector or a measure vector) are not
 * launched one by one: libalgorithm.so keeps up to 8 of them per (device, stream) and runs them as
 * ONE kernel that reads the index vector once.  Nothing observable changes: every other
 * libalgorithm.so entry point, and every libmem.so entry point through which the host could read,
 * overwrite or free device memory (WaitForCudaStream, AsyncCopy*, DeviceFree, DestroyCudaStream,
 * CudaProfilerStop), first launches what is pending.  libalgorithm.so finds the sibling libmem.so
 * (same directory) at run time and registers AresFlushDeferred through AresMemSetFlushHook; when
 * that is not possible (another allocator library in use) or ARES_DEFER=0 is set, every transform
 * is launched immediately as before. */
void AresFlushDeferred(int device);                      /* exported by libalgorithm.so */
void AresMemSetFlushHook(void (*hook)(int device));      /* exported by libmem.so */

/* Cross-call fusion, second stage: the pending dimension / measure transforms of a batch are never
 * launched when the HashReduce call that follows can evaluate them on the fly (the kernels of
 * AresFusedFilterHashReduce below): the dimension and measure vectors of the batch's rows are then
 * neither written nor read.  For that the pending work must survive the two things the Go host does
 * between project() and reduce() (query/aql_batchexecutor.go:213-216): WaitForCudaStream, and
 * DeviceFree of the batch's columns and of the index vector.  libmem.so therefore reports those
 * events instead of flushing blindly:
 *   on_wait   — the host waits for `stream`; libalgorithm.so keeps work pending only if nothing but
 *               a later libalgorithm.so call can observe its results;
 *   on_free   — returns a non-zero tag (it names the stream whose work it is) when not-yet-launched work
 *               still reads the block: libmem.so keeps the block aside (not reusable) until
 *               AresMemReleaseHeld is called with that tag; frees of a pending OUTPUT launch the work
 *               first;
 *   on_access — a copy is about to touch [ptr, ptr + bytes): work whose inputs or outputs overlap
 *               is launched first (also work that HashReduce had skipped: the skipped transforms
 *               stay launchable until the next batch begins);
 *   flush     — everything pending on the device is launched (all other entry points).
 * A HashReduce that cannot use the pending work as it is (other aggregate, layout, joins, ...) simply
 * launches it and proceeds as before.  ARES_FUSE=0 switches this stage off, ARES_DEFER=0 both.
 * The same bookkeeping lets a fast filter return its survivor count without compacting the index
 * vector yet (the predicate vector is written): the compaction runs as soon as anything reads the
 * vector — a second filter, transforms that are launched after all, a copy, any other entry point —
 * and never when HashReduce re-derives the survivors itself.  ARES_LAZY_COMPACT=0 compacts at once. */
typedef struct {
  void (*flush)(int device);
  void (*on_wait)(int device, void *stream);
  uintptr_t (*on_free)(int device, void *ptr, size_t bytes);
  void (*on_access)(int device, const void *ptr, size_t bytes);
} AresDeferralHooks;
void AresMemSetDeferralHooks(const AresDeferralHooks *hooks); /* exported by libmem.so */
void AresMemReleaseHeld(int device, uintptr_t tag);           /* exported by libmem.so; tag 0 = every held block */

/* Further notifications from libmem.so to libalgorithm.so (optional; `size` = sizeof of the caller's
 * struct, members past it are taken as absent):
 *   on_write          — a copy / memset is about to WRITE [ptr, ptr + bytes) (on_access is called as
 *                       well): results whose layout libalgorithm.so remembers (partition-grouped
 *                       HashReduce outputs) are no longer trusted;
 *   on_stream_destroy — DestroyCudaStream, after the stream has been synchronised: per-stream caches
 *                       and bookkeeping of libalgorithm.so are released;
 *   trim              — libmem.so could not allocate: libalgorithm.so gives its cached temporaries back.
 * AresMemTrimCache is the opposite direction: libalgorithm.so could not allocate, libmem.so releases its
 * parked blocks of the device. */
typedef struct {
  size_t size;
  void (*on_write)(int device, const void *ptr, size_t bytes);
  void (*on_stream_destroy)(int device, void *stream);
  void (*trim)(int device);
} AresMemAuxHooks;
void AresMemSetAuxHooks(const AresMemAuxHooks *hooks); /* exported by libmem.so */
void AresMemTrimCache(int device);                     /* exported by libmem.so */
/* Write tracking.  DeviceAllocate hands out cleared memory; libmem.so clears a block when it is freed, and
 * only the byte ranges written since the block was last cleared.  libmem.so sees its own copies and fills;
 * libalgorithm.so reports the outputs of its kernels with AresMemNoteWrite and switches tracking on with
 * AresMemEnableWriteTracking (never called: a freed block counts as written all over).  A host component
 * that lets anything else write into DeviceAllocate memory (a collective library receiving into it) reports
 * those writes with AresMemNoteWrite too. */
void AresMemNoteWrite(int device, const void *ptr, size_t bytes);
void AresMemEnableWriteTracking(void);

/* Shared fences.  DeviceFree fences a block with one event per stream; frees that follow one another with NOTHING submitted
 * to the device in between (the Go host frees a batch's columns, index and predicate vector in a row) share the events of
 * the first.  "Nothing submitted" must be known: libalgorithm.so reports every entry point and every kernel launch with
 * AresMemNoteActivity and switches sharing on with AresMemEnableActivityTracking (never called: every free records its
 * own events).  A host component that submits work to the query's streams itself (the driver's RCCL calls) reports it
 * the same way. */
void AresMemNoteActivity(void);
void AresMemEnableActivityTracking(void);

/* Books of libmem.so for one device: bytes / number of blocks the host holds (DeviceAllocate, deviceMalloc;
 * held blocks were freed by the host but are kept aside for deferred work and are NOT counted as live),
 * blocks kept aside, bytes parked in the cache.  Any pointer may be NULL. */
void AresMemStats(int device, size_t *liveBytes, size_t *liveBlocks, size_t *heldBlocks, size_t *parkedBytes);
/* Driver calls libmem.so's block cache could not avoid on `device` since the process started (hipMalloc, hipFree,
 * cache trims): after warm-up a steady workload adds none.  Any pointer may be NULL. */
void AresMemDriverCalls(int device, size_t *mallocs, size_t *frees, size_t *trims);
/* Events of the block cache (fences of parked blocks, recycled events) whose last record was on `stream`.  An event must
 * not be touched once its stream is destroyed (the ROCm 7.x runtime reaches into the stream object: use after free), so
 * DestroyCudaStream retires them while the stream exists: 0 right after it, for tests. */
size_t AresMemStreamEvents(int device, void *stream);
/* the same for libalgorithm.so: error-word watches of lazily launched compactions and unresolved profiler events */
size_t AresStreamEvents(int device, void *stream);
/* libalgorithm.so's stream temporaries (workspaces, decoded run-length columns, row hashes kept beside results): bytes handed
 * out and not yet released, and bytes sitting in its cache.  A steady workload keeps the first bounded (tests: nothing may
 * pile up batch after batch).  Either pointer may be NULL. */
void AresTempStats(size_t *handedOutBytes, size_t *cachedBytes);

/* Fused batch execution: filter -> dimension / measure projection -> hash reduction of ONE batch in
 * a single pass over the source columns, without the index / predicate / dimension vectors the
 * one-call-per-AST-node ABI materialises in between (~145 B/row of HBM traffic on
 * BASELINE config C3; this path moves ~53 B/row).  It replaces, for one batch, the call sequence
 *   InitIndexVector, BinaryFilter x numFilters, Unary/BinaryTransform x (numDims + 1), HashReduce
 * of query/aql_batchexecutor.go:103-273 and produces the same groups and aggregates.
 *
 * Shapes accepted (anything else returns an error string starting with "not fusable", and the host
 * simply runs the ordinary sequence): every expression is a main-table VectorPartyInput of a
 * 4-byte type (modes 1/2), bare (arity 1, Noop) or combined with a ConstantInput by a binary
 * functor (arity 2); filters are comparison functors (ANDed); dimensions are stored as
 * Int32 / Uint32 / Float32; the aggregate is one HashReduce supports with native LDS atomics.
 * prevKeys / prevValues hold the prevSize groups accumulated so far (the reference keeps them as the
 * first rows of the input vectors, query/aql_processor.go:752-774); outKeys.VectorCapacity must be
 * at least prevSize + batchRows.  res = number of groups written to outKeys / outValues. */
typedef struct {
  InputVector lhs, rhs; /* rhs is read only when arity == 2 */
  int arity;
  int functor; /* Unary/BinaryFunctorType */
  enum DataType outType;
} AresFusedExpr;

typedef struct {
  int numFilters;
  AresFusedExpr filters[4];
  int numDims;
  AresFusedExpr dims[4];
  AresFusedExpr measure; /* outType = data type of the measure vector */
  enum AggregateFunction aggFunc;
} AresFusedQuery;

CGoCallResHandle AresFusedFilterHashReduce(const AresFusedQuery *query, int batchRows, DimensionVector prevKeys,
                                           uint8_t *prevValues, int prevSize, DimensionVector outKeys,
                                           uint8_t *outValues, void *cudaStream, int device);

#ifdef __cplusplus
}
#endif
#endif /* ARES_EXTENSIONS_H_ */
