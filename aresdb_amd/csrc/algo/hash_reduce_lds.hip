// HashReduce, partitioned: kernels and host side (device code: hr_kernels.hpp).
#include <hip/hip_runtime.h>

#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>

#include <memory>
#include <mutex>
#include <vector>

#include "aggregate.hpp"
#include "common.hpp"
#include "device_model.hpp"
#include "ares_extensions.h"
#include "dim_layout.hpp"
#include "fast_eval.hpp"
#include "hash_reduce_lds.hpp"
#include "hr_kernels.hpp"
#include "hr_rtc.hpp"
#include "sort_reduce_fused.hpp"

namespace ares {

namespace {
using namespace hr;

__global__ __launch_bounds__(kThreads) void hr_partition_kernel(const uint8_t *dimValues, DimLayoutD L, size_t capacity,
                                                                const uint8_t *inputValues, AggSpec a, int length,
                                                                Workspace ws) {
  partition_generic_body(dimValues, L, capacity, inputValues, a, length, ws);
}

template <int ND, int VW>
__global__ __launch_bounds__(kThreads) void hr_partition4_kernel(const uint8_t *dimValues, size_t capacity,
                                                                 const uint8_t *inputValues, uint32_t rowStart, AggSpec a,
                                                                 int length, Workspace ws, int allowDirect) {
  DimVectorSource<ND, VW> src{dimValues, capacity, inputValues, rowStart};
  partition_body(src, a, length, ws, allowDirect);
}

template <int ND>
__global__ __launch_bounds__(kThreads) void hr_fused_scan_kernel(FusedPlanD plan, uint32_t rowBase, AggSpec a, int length,
                                                                 Workspace ws) {
  FusedSource<ND> src(plan, rowBase);
  partition_body(src, a, length, ws, 1);
}

template <int ND4, int RWB>
__global__ __launch_bounds__(kThreads) void hr_merge_kernel(const uint8_t *__restrict__ dimIn,
                                                            const uint8_t *__restrict__ inValues,
                                                            uint8_t *__restrict__ dimOut, DimLayoutD L, size_t capacity,
                                                            uint8_t *__restrict__ outputValues, AggSpec a, Workspace ws,
                                                            uint32_t prevSize) {
  merge_body<ND4, false, RWB>(dimIn, capacity, inValues, dimOut, L, capacity, outputValues, a, ws, nullptr, prevSize);
}

template <int ND, int RWB>
__global__ __launch_bounds__(kThreads) void hr_fused_merge_kernel(FusedPlanD plan, const uint8_t *__restrict__ prevDims,
                                                                  size_t prevCapacity,
                                                                  const uint8_t *__restrict__ prevValues, uint32_t prevSize,
                                                                  uint8_t *__restrict__ dimOut, size_t outCapacity,
                                                                  uint8_t *__restrict__ outputValues, AggSpec a, Workspace ws) {
  DimLayoutD L;  // unused by the all-4-byte emission
  L.numDims = ND;
  merge_body<ND, true, RWB>(prevDims, prevCapacity, prevValues, dimOut, L, outCapacity, outputValues, a, ws, &plan, prevSize);
}

// ---- partition-grouped results ---------------------------------------------------------------------
// A merge emits every partition's groups as a few contiguous row ranges of the output vectors and
// records them.  When the host feeds those vectors back as the first rows of the next HashReduce
// input (query/aql_processor.go:743-776: previous results first, the batch's rows behind them), the
// next merge reads each partition's previous groups straight from its ranges: they are neither
// re-hashed into records nor re-partitioned, so a batch costs O(its rows + groups streamed once)
// instead of O(rows + groups x (partition + merge + gather)).  The ranges are only trusted while
// nothing has written to the vectors: frees, copies into them and every libalgorithm.so writer drop
// the entry (grouped_note_write), the merge itself checks that every row it is handed hashes into
// its partition (a mismatch makes the host redo the call the long way), and the feature is off unless
// the sibling libmem.so reports frees and copies (deferral_hooks_active()).  ARES_GROUPED=0: off.
// free range buffers per device.  Declared before the state table: the table's buffers return here when it is destroyed
// at exit.  The mutex guards the free list only: taken by the buffers' deleter, possibly under g_groupedMutex, never
// the other way round.
std::mutex g_rangesMutex;
std::vector<std::pair<int, uint32_t *>> g_freeRanges;

// dimension slots of a vector the grouped state (and the generated kernels) can describe: at most kFusedDims dimensions of
// 4, 2 or 1 bytes (dim_layout.hpp: descending width order)
struct SlotWidths {
  int nd = 0, dimBytes = 0;  // dimBytes: value bytes of one row
  uint8_t width[kFusedDims] = {}, off[kFusedDims] = {};
  bool same(const SlotWidths &o) const { return nd == o.nd && memcmp(width, o.width, sizeof(width)) == 0; }
  size_t row_bytes() const { return static_cast<size_t>(dimBytes + nd); }
};
bool slot_widths(const uint8_t numDimsPerDimWidth[NUM_DIM_WIDTH], SlotWidths *out) {
  SlotWidths w;
  if (numDimsPerDimWidth[0] || numDimsPerDimWidth[1]) return false;
  for (int k = 2; k < NUM_DIM_WIDTH; k++)
    for (int j = 0; j < numDimsPerDimWidth[k]; j++) {
      if (w.nd >= kFusedDims) return false;
      w.width[w.nd] = static_cast<uint8_t>(1 << (NUM_DIM_WIDTH - 1 - k));
      w.off[w.nd] = static_cast<uint8_t>(w.dimBytes);
      w.dimBytes += w.width[w.nd];
      w.nd++;
    }
  *out = w;
  return w.nd >= 1;
}

struct GroupedState {
  int device;
  const uint8_t *dims;
  const uint8_t *values;
  size_t capacity;
  SlotWidths slots;
  int valueBytes, size, partBits;
  // kMaxPartitions x kRangeWords words of device memory.  Shared: a call that looked the state up keeps the buffer
  // alive while its merge reads it, whatever another thread's eviction or overwrite does to the table meanwhile; the
  // buffer goes back to the free list with the last reference.  Null: the rows are not grouped by partition (results of
  // image-mode merges are in order of first appearance).
  std::shared_ptr<uint32_t> ranges;
  // ---- table image (see "table images" below): the partitions' LDS tables as the merge that produced these vectors
  // left them — what the query's next HashReduce starts from instead of re-hashing and re-inserting every group
  std::shared_ptr<uint8_t> image;
  uint64_t lineage = 0;     // the query (chain of HashReduce calls) the image belongs to
  // Ancestry inside a lineage: every image-mode result gets a generation; a result that STARTED from an image records the
  // generation of the state it started from.  Only a state's parent holds, in its leading rows, exactly the groups the
  // state began with — same lineage alone does not say that (two results forked from one input share it).
  uint64_t gen = 0, parentGen = 0;
  bool lazyValues = false;  // rows [0, size) of `values` are NOT written: they are what the image's value plane holds
};
std::mutex g_groupedMutex;
std::vector<GroupedState> g_grouped;
constexpr size_t kRangesBytes = sizeof(uint32_t) * kMaxPartitions * kRangeWords;

bool grouped_enabled() {
  static EnvSwitch<bool> on("ARES_GROUPED", [](const char *e) { return !(e && e[0] == '0'); });
  return on.get() && deferral_hooks_active();
}

std::shared_ptr<uint32_t> take_ranges(int device) {
  uint32_t *p = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_rangesMutex);
    for (size_t i = 0; i < g_freeRanges.size(); i++)
      if (g_freeRanges[i].first == device) {
        p = g_freeRanges[i].second;
        g_freeRanges[i] = g_freeRanges.back();
        g_freeRanges.pop_back();
        break;
      }
  }
  if (!p) {
    void *fresh = nullptr;
    hip_check(hipMalloc(&fresh, kRangesBytes), "hipMalloc");
    p = static_cast<uint32_t *>(fresh);
  }
  return std::shared_ptr<uint32_t>(p, [device](uint32_t *q) {
    std::lock_guard<std::mutex> lock(g_rangesMutex);
    g_freeRanges.emplace_back(device, q);
  });
}

// ---- table images ------------------------------------------------------------------------------------------------
// What a live batch (2 Mi rows against a result of millions of groups) used to pay per HashReduce: every partition re-hashed
// and re-inserted its ~4.5 k previous groups from the previous output vectors and wrote all of them out again — twice the
// work of the batch's own records.  Now the merge of a DIRECT-mode batch leaves each partition's LDS table in HBM (keys,
// the output position of every group, values: kSlots x 16 B per partition, 64 MB for 512 partitions, coalesced) and the
// next call's merge starts from it (hr_rtc.hip generate_merge image = 2): previous groups cost one coalesced load, only
// groups the batch sees for the first time are emitted — appended, so a group keeps its position for the life of the query
// — and the measure vector is not written at all: it is DEFINED by the image's value plane ("lazy values") and written by
// hr_image_values_kernel when somebody reads it (a copy, an entry point that is handed the vector, an overwrite that leaves
// part of it) — at the latest when the host fetches the result.  Both result buffers of a query (the Go host ping-pongs
// them) carry their own image: a merge reads the input buffer's and writes the output buffer's, so either buffer can be
// materialised at any time.  The dimension rows of the output buffer are kept complete: the merge copies over the rows
// the output buffer has not seen (those added while it was the input's partner: a handful per batch in steady state).
// Everything is trusted only while nothing has written to the vectors (grouped_note_write), as the ranges are.
// ARES_IMAGE=0: off.
constexpr size_t kImagePartBytes = static_cast<size_t>(kSlots) * 16;
constexpr size_t kImageCountsOff = kImagePartBytes * kMaxPartitions;
constexpr size_t kImageBytes = kImageCountsOff + sizeof(uint32_t) * kMaxPartitions;
std::vector<std::pair<int, uint8_t *>> g_freeImages;  // (guarded by g_rangesMutex)
std::atomic<uint64_t> g_lineage{0};
std::atomic<uint64_t> g_generation{0};

bool image_enabled() {
  static EnvSwitch<bool> on("ARES_IMAGE", [](const char *e) { return !(e && e[0] == '0'); });
  return on.get() && grouped_enabled();
}

std::shared_ptr<uint8_t> take_image(int device) {
  uint8_t *p = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_rangesMutex);
    for (size_t i = 0; i < g_freeImages.size(); i++)
      if (g_freeImages[i].first == device) {
        p = g_freeImages[i].second;
        g_freeImages[i] = g_freeImages.back();
        g_freeImages.pop_back();
        break;
      }
  }
  if (!p) {  // images are an optimisation: out of memory means "no image this time" (null), not a failed HashReduce
    void *fresh = nullptr;
    if (hipMalloc(&fresh, kImageBytes) != hipSuccess) {
      (void)hipGetLastError();
      grouped_trim(device);
      if (g_memTrimCache) g_memTrimCache(device);
      fresh = nullptr;
      if (hipMalloc(&fresh, kImageBytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
      }
    }
    p = static_cast<uint8_t *>(fresh);
  }
  return std::shared_ptr<uint8_t>(p, [device](uint8_t *q) {
    std::lock_guard<std::mutex> lock(g_rangesMutex);
    g_freeImages.emplace_back(device, q);
  });
}
uint32_t *image_counts(uint8_t *image) { return reinterpret_cast<uint32_t *>(image + kImageCountsOff); }

// values[pos] = value for every group of the image: the measure vector an image-mode merge left unwritten
template <int VW>
__global__ __launch_bounds__(kThreads) void hr_image_values_kernel(const uint4 *image, uint8_t *values, uint32_t size) {
  const uint4 *img = image + static_cast<size_t>(blockIdx.x) * kSlots;
  const uint32_t *keys = reinterpret_cast<const uint32_t *>(img), *pos = keys + kSlots;
  const uint64_t *vals = reinterpret_cast<const uint64_t *>(img + kSlots / 2);
  for (int s = threadIdx.x; s < kSlots; s += kThreads) {
    if (keys[s] == 0u) continue;
    const uint32_t at = pos[s];
    if (at >= size) continue;  // (cannot happen: positions are handed out below the result's size)
    if (VW == 8) reinterpret_cast<uint64_t *>(values)[at] = vals[s];
    else reinterpret_cast<uint32_t *>(values)[at] = static_cast<uint32_t>(vals[s]);
  }
}

hipStream_t image_stream(int device) {  // materialisations run here, synchronously: rare (once per query, at its fetch)
  static std::mutex mu;
  static std::map<int, hipStream_t> streams;
  std::lock_guard<std::mutex> lock(mu);
  auto it = streams.find(device);
  if (it != streams.end()) return it->second;
  hipStream_t s = nullptr;
  hip_check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
  streams[device] = s;
  return s;
}

// caller holds g_groupedMutex (or owns `s` outright).  The image's producer has completed: the HashReduce that wrote it
// waited for its stream before it returned.
void materialize_values(GroupedState &s) {
  if (!s.lazyValues || !s.image) return;
  int current = 0;
  const bool have = hipGetDevice(&current) == hipSuccess;
  if (!have || current != s.device) hip_check(hipSetDevice(s.device), "hipSetDevice");
  hipStream_t stream = image_stream(s.device);
  uint8_t *values = const_cast<uint8_t *>(s.values);
  if (s.valueBytes == 8)
    ARES_LAUNCH("hr_image_values_kernel", hr_image_values_kernel<8>, 1 << s.partBits, kThreads, stream,
                reinterpret_cast<const uint4 *>(s.image.get()), values, static_cast<uint32_t>(s.size));
  else
    ARES_LAUNCH("hr_image_values_kernel", hr_image_values_kernel<4>, 1 << s.partBits, kThreads, stream,
                reinterpret_cast<const uint4 *>(s.image.get()), values, static_cast<uint32_t>(s.size));
  hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
  mem_note_write(s.device, values, static_cast<size_t>(s.valueBytes) * s.size);
  s.lazyValues = false;
  if (have && current != s.device) (void)hipSetDevice(current);
}

bool state_overlaps(const GroupedState &s, const uint8_t *lo, const uint8_t *hi) {
  auto hit = [&](const uint8_t *a, size_t bytes) { return a < hi && lo < a + bytes; };
  for (int d = 0; d < s.slots.nd; d++) {
    if (hit(s.dims + s.capacity * s.slots.off[d], static_cast<size_t>(s.slots.width[d]) * s.size)) return true;
    if (hit(s.dims + s.capacity * s.slots.dimBytes + s.capacity * d, static_cast<size_t>(s.size))) return true;
  }
  return hit(s.values, static_cast<size_t>(s.valueBytes) * s.size);
}

bool grouped_lookup(int device, const uint8_t *dims, const uint8_t *values, size_t capacity, const SlotWidths &slots, int valueBytes,
                    GroupedState *out) {
  if (!grouped_enabled()) return false;
  std::lock_guard<std::mutex> lock(g_groupedMutex);
  for (const GroupedState &s : g_grouped)
    if (s.device == device && s.dims == dims && s.values == values && s.capacity == capacity && s.slots.same(slots) &&
        s.valueBytes == valueBytes) {
      *out = s;
      return true;
    }
  return false;
}

void grouped_register(const GroupedState &s) {
  std::lock_guard<std::mutex> lock(g_groupedMutex);
  if (g_grouped.size() >= 64) {  // a host that never frees: forget the oldest (its unwritten values are written first)
    materialize_values(g_grouped.front());
    g_grouped.erase(g_grouped.begin());
  }
  g_grouped.push_back(s);
}

struct Regions {
  Workspace ws;
  std::unique_ptr<StreamBuffer> buf;
  size_t headBytes;
};

// previous groups from which a query is taken to be high-cardinality: beyond ~3/4 of an LDS table (8192 slots) a
// workgroup's table no longer holds the query's groups and rows would spill one by one (ARES_LEAN_MIN_GROUPS
// overrides: 0 sends every fusable batch to the specialised DIRECT-mode kernels — tests)
int lean_min_groups() {
  static EnvSwitch<int> v("ARES_LEAN_MIN_GROUPS", [](const char *e) { return e ? atoi(e) : 6000; });
  return v.get();
}

// groups the first batch of a plan shape produced the last time (cardinality feedback for first batches)
std::mutex g_shapeMutex;
std::unordered_map<size_t, int> g_firstBatchGroups;

int part_bits_for(int64_t length) {
  // ARES_MIN_PART_BITS (tests): at least that many partition bits — with 2, inputs of a few thousand rows take the generated
  // merges (whose 32-bit table keys need two spare hash bits), hence the table images, like production-sized ones
  static EnvSwitch<int> minBits("ARES_MIN_PART_BITS", [](const char *e) { return e ? atoi(e) : 0; });
  int partBits = minBits.get() > 0 ? (minBits.get() < 9 ? minBits.get() : 9) : 0;
  while ((8192ll << partBits) < length && (1 << partBits) < kMaxPartitions) partBits++;
  return partBits;
}

int grid_for(int64_t rows) {
  const int64_t tiles = (rows + kQuadTile - 1) / kQuadTile;
  return static_cast<int>(tiles < kMaxStreams ? (tiles < 1 ? 1 : tiles) : kMaxStreams);
}

// compact lines (8-byte records, hr::Workspace::lineRecords == 14) for the plan-sourced DIRECT scan; ARES_COMPACT=0:
// 16-byte records everywhere
bool compact_enabled() {
  static EnvSwitch<bool> on("ARES_COMPACT", [](const char *e) { return !(e && e[0] == '0'); });
  return on.get();
}

// rowsA: rows that may end up as region-A records (TABLE-mode flushes, ungrouped previous groups);
// rowsB / streams / rwB: rows, workgroups and record width (words) of the launch that may write region B;
// lineRecords: 0 = scattered records, 8 = 16-byte records in whole lines, 14 = compact lines (capB counts lines)
void make_regions(Regions &r, int partBits, int64_t rowsA, int64_t rowsB, int streams, int rwB, hipStream_t stream,
                  int lineRecords = 0) {
  const int numParts = 1 << partBits;
  Workspace &ws = r.ws;
  memset(&ws, 0, sizeof(ws));
  ws.partBits = partBits;
  ws.streams = streams;
  // region stride = capA * 16 B; keep it off large powers of two (cap = 17 mod 64 records) so that the
  // merge workgroups, which stream their regions in lockstep, do not camp on the same HBM channels
  ws.capA = ((2ull * (static_cast<uint64_t>(rowsA) / numParts) + 2 * kSlots) | 63ull) + 18;
  ws.capB = 0;
  ws.lineRecords = lineRecords;
  size_t bBytes = 0;
  if (streams > 0) {
    const uint64_t mean = (static_cast<uint64_t>(rowsB) / (static_cast<uint64_t>(numParts) * streams)) << record_stream_slack();
    if (lineRecords == static_cast<int>(kCompactLineRecords)) {  // whole 128-byte lines of 14 records; an odd number of lines per stream
      ws.capB = static_cast<uint32_t>(((2 * mean + 64) / kCompactLineRecords + 2) | 1ull);
      bBytes = 128ull * ws.capB * numParts * streams;
    } else if (lineRecords) {  // whole 128-byte lines of 8 records; an odd number of lines per stream keeps the strides off powers of two
      ws.capB = static_cast<uint32_t>(((2 * mean + 64 + 7) / 8 * 8) | 8ull);
      bBytes = sizeof(uint32_t) * rwB * static_cast<size_t>(ws.capB) * numParts * streams;
    } else {
      ws.capB = static_cast<uint32_t>(((2 * mean + 64) | 15ull) + 6);
      bBytes = sizeof(uint32_t) * rwB * static_cast<size_t>(ws.capB) * numParts * streams;
    }
  }
  bBytes = (bBytes + 255) / 256 * 256;
  const size_t headBytes = (sizeof(uint32_t) * (numParts + 8) + 255) / 256 * 256;  // cursors, outCount[4], the result ticket
  const size_t countsBytes = (sizeof(uint32_t) * static_cast<size_t>(numParts) * (streams > 0 ? streams : 1) + 255) / 256 * 256;
  const size_t aBytes = (sizeof(uint4) * ws.capA * numParts + 255) / 256 * 256;
  r.buf.reset(new StreamBuffer(headBytes + countsBytes + aBytes + bBytes + 256, stream));
  uint8_t *base = r.buf->as<uint8_t>();
  ws.cursorsA = reinterpret_cast<uint32_t *>(base);
  ws.outCount = ws.cursorsA + numParts;
  ws.countsB = reinterpret_cast<uint32_t *>(base + headBytes);
  ws.recA = reinterpret_cast<uint4 *>(base + headBytes + countsBytes);
  ws.recB = reinterpret_cast<uint32_t *>(base + headBytes + countsBytes + aBytes);
  r.headBytes = headBytes;
  hip_check(hipMemsetAsync(base, 0, headBytes, stream), "hipMemsetAsync");
}

struct MergeResult {
  uint32_t groups, overflow, stale, needGeneric;
};
MergeResult read_result(const Workspace &ws, hipStream_t stream) {
  uint32_t w[4] = {0, 0, 0, 0};
  read_back_u32(ws.outCount, w, 4, stream);
  return MergeResult{w[0], w[1], w[2], w[3]};
}
// ... when the merge was generated at run time: its last workgroup has written the words into the thread's pinned slot
// Off by default: measured SLOWER than the 16-byte copy command behind the kernel (live-batch leg 69.8 / 69.3 ms with it,
// 63.2 / 63.6 ms without, alternating runs on one box: profiles/r5_evidence_ab.txt — the ticket, the two fences and the
// system-scope stores at the tail of a 512-workgroup kernel cost more than the copy they replace).  ARES_RESULT_PINNED=1: on.
uint32_t *result_slot() {
  static EnvSwitch<bool> on("ARES_RESULT_PINNED", [](const char *e) { return e && e[0] == '1'; });
  return on.get() ? reinterpret_cast<uint32_t *>(pinned_words()) : nullptr;
}
MergeResult read_result_pinned(hipStream_t stream) {
  hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
  const volatile uint32_t *w = result_slot();
  return MergeResult{w[0], w[1], w[2], w[3]};
}

}  // namespace

void grouped_note_write(int device, const void *ptr, size_t bytes) {
  sorted_state_note_write(device, ptr, bytes);  // (sort_reduce_fused.hip: the row hashes kept beside a Sort + Reduce result)
  const uint8_t *lo = static_cast<const uint8_t *>(ptr);
  const uint8_t *hi = lo + (bytes ? bytes : 1);
  std::lock_guard<std::mutex> lock(g_groupedMutex);
  for (size_t i = 0; i < g_grouped.size();) {
    GroupedState &st = g_grouped[i];
    if (st.device == device && state_overlaps(st, lo, hi)) {
      // values that were never written are written now — unless the write (or free) takes all of them with it
      const uint8_t *vlo = st.values, *vhi = st.values + static_cast<size_t>(st.valueBytes) * st.size;
      if (st.lazyValues && !(lo <= vlo && vhi <= hi)) materialize_values(st);
      g_grouped.erase(g_grouped.begin() + i);
    } else {
      i++;
    }
  }
}

// [ptr, ptr + bytes) is about to be read (a copy, an entry point that is handed the vector): measure rows an image-mode
// merge left unwritten are written first.  The state stays: its rows are still what the merge produced.
void grouped_materialize_for_read(int device, const void *ptr, size_t bytes) {
  if (!ptr) return;
  const uint8_t *lo = static_cast<const uint8_t *>(ptr);
  const uint8_t *hi = lo + (bytes ? bytes : 1);
  std::lock_guard<std::mutex> lock(g_groupedMutex);
  for (GroupedState &st : g_grouped) {
    if (st.device != device || !st.lazyValues) continue;
    const uint8_t *vlo = st.values, *vhi = st.values + static_cast<size_t>(st.valueBytes) * st.size;
    if (vlo < hi && lo < vhi) materialize_values(st);
  }
}

void grouped_trim(int device) {
  std::vector<void *> idle;
  {
    std::lock_guard<std::mutex> lock(g_rangesMutex);
    for (size_t i = 0; i < g_freeRanges.size();)
      if (g_freeRanges[i].first == device) {
        idle.push_back(g_freeRanges[i].second);
        g_freeRanges[i] = g_freeRanges.back();
        g_freeRanges.pop_back();
      } else {
        i++;
      }
    for (size_t i = 0; i < g_freeImages.size();)
      if (g_freeImages[i].first == device) {
        idle.push_back(g_freeImages[i].second);
        g_freeImages[i] = g_freeImages.back();
        g_freeImages.pop_back();
      } else {
        i++;
      }
  }
  for (void *p : idle) (void)hipFree(p);  // (a buffer is on a free list only after the last call that used it has waited for its stream)
}

void grouped_note_write(int device, const DimensionVector &v) {
  size_t rowBytes = 0;
  for (int w = 0; w < NUM_DIM_WIDTH; w++) rowBytes += static_cast<size_t>(v.NumDimsPerDimWidth[w]) * ((1u << (NUM_DIM_WIDTH - 1 - w)) + 1);
  if (v.DimValues && v.VectorCapacity > 0) grouped_note_write(device, v.DimValues, rowBytes * static_cast<size_t>(v.VectorCapacity));
}

namespace {
std::atomic<int> g_recordStreamSlack{0};
}
int record_stream_slack() { return g_recordStreamSlack.load(std::memory_order_relaxed); }
bool grow_record_stream_slack() {
  int s = g_recordStreamSlack.load(std::memory_order_relaxed);
  while (s < 2)
    if (g_recordStreamSlack.compare_exchange_weak(s, s + 1)) return true;
  return false;
}

bool hash_reduce_lds_supported(const AggSpec &a) {
  if (a.op == OP_AVG) return false;
  if (a.vtype == V_F32 && a.op != OP_SUM) return false;  // float min/max keep the reference's compare form
  return true;
}

int hash_reduce_lds(int device, const DimensionVector &inputKeys, const uint8_t *inputValues,
                    const DimensionVector &outputKeys, uint8_t *outputValues, const AggSpec &a, int length,
                    hipStream_t stream) {
  const DimLayoutD L = make_dim_layout(inputKeys.NumDimsPerDimWidth);
  const size_t capacity = static_cast<size_t>(inputKeys.VectorCapacity);
  bool all4 = L.numDims >= 1 && L.numDims <= 4;  // beyond 4 dims the double-buffered quads spill
  for (int d = 0; d < L.numDims; d++) all4 = all4 && L.width[d] == 4;
  // a layout whose partition-grouped result a later FUSED batch can start from (narrow slots: the generated merge only,
  // which needs two spare hash bits in its 32-bit table keys: at least four partitions)
  SlotWidths slots;
  const bool describable = slot_widths(inputKeys.NumDimsPerDimWidth, &slots);
  const int partBits = std::max(part_bits_for(length), (describable && !all4) ? 2 : 0);
  const int numParts = 1 << partBits;
  // the output vectors are about to be rewritten: whatever was known about them is void
  grouped_note_write(device, outputValues, static_cast<size_t>(a.width) * capacity);  // (values first: unwritten ones die whole)
  grouped_note_write(device, outputKeys.DimValues, static_cast<size_t>(L.rowBytes) * capacity);
  GroupedState prev;
  bool grouped = all4 && grouped_lookup(device, inputKeys.DimValues, inputValues, capacity, slots, a.width, &prev) &&
                 prev.ranges && prev.partBits == partBits && prev.size > 0 && prev.size <= length;
  // (the reference goes with this call unless the result is registered below: nothing leaks when a launch throws)
  const std::shared_ptr<uint32_t> outRangesRef = (describable && grouped_enabled()) ? take_ranges(device) : nullptr;
  uint32_t *outRanges = outRangesRef.get();
  MergeResult res{0, 0, 0, 0};
  for (;;) {
    const int start = grouped ? prev.size : 0;
    const int rows = length - start;
    // a query that already has more groups than an LDS table holds: the write-combining scan, generated
    // for this vector shape (hr_rtc.hip), instead of the adaptive kernel's scattered DIRECT-mode stores
    RtcKernel lean;
    if (all4 && grouped && rows > 0 && start >= lean_min_groups() && (a.width == 4 || a.width == 8) && rtc_scan_available())
      lean = rtc_vector_scan_lookup(device, L.numDims, a.width, partBits);
    const int streams = (all4 && rows > 0) ? (lean ? rtc_scan_grid(rows) : grid_for(rows)) : 0;
    const int rwB = (a.width == 8 || lean) ? 4 : 3;
    Regions r;
    make_regions(r, partBits, rows, rows, streams, rwB, stream, lean ? 8 : 0);
    Workspace &ws = r.ws;
    if (lean && a.width == 8) ws.widen.mode = 2;  // line records carry the whole 8-byte value
    ws.prevRanges = grouped ? prev.ranges.get() : nullptr;
    ws.outRanges = outRanges;
    // the specialised merge for these records (it writes every partition's range entry itself)
    RtcKernel leanMerge = lean ? rtc_vector_merge_lookup(device, L.numDims, a.width, partBits, a) : nullptr;
    if (outRanges && !leanMerge) hip_check(hipMemsetAsync(outRanges, 0, kRangesBytes, stream), "hipMemsetAsync");
    if (lean) {
      rtc_vector_scan_launch(lean, inputKeys.DimValues, capacity, inputValues, L.numDims, a.width, static_cast<uint32_t>(start), rows,
                             ws, stream);
    } else if (all4) {
      if (rows > 0) {
#define ARES_HR_CASE(ND)                                                                                             \
  case ND:                                                                                                           \
    if (a.width == 8)                                                                                                \
      ARES_LAUNCH("hr_partition4_kernel", (hr_partition4_kernel<ND, 8>), streams, kThreads, stream, inputKeys.DimValues, \
                  capacity, inputValues, static_cast<uint32_t>(start), a, rows, ws, 1);                             \
    else                                                                                                             \
      ARES_LAUNCH("hr_partition4_kernel", (hr_partition4_kernel<ND, 4>), streams, kThreads, stream, inputKeys.DimValues, \
                  capacity, inputValues, static_cast<uint32_t>(start), a, rows, ws, 1);                             \
    break;
        switch (L.numDims) {
          ARES_HR_CASE(1) ARES_HR_CASE(2) ARES_HR_CASE(3) ARES_HR_CASE(4)
        }
#undef ARES_HR_CASE
      }
    } else {
      const int64_t tiles = (static_cast<int64_t>(length) + kTileRows - 1) / kTileRows;
      const int grid = static_cast<int>(tiles < 256 ? tiles : 256);
      ARES_LAUNCH("hr_partition_kernel", hr_partition_kernel, grid, kThreads, stream, inputKeys.DimValues, L, capacity,
                  inputValues, a, length, ws);
    }
#define ARES_HR_MERGE(ND, RWB)                                                                                       \
  ARES_LAUNCH("hr_merge_kernel", (hr_merge_kernel<ND, RWB>), numParts, kThreads, stream, inputKeys.DimValues, inputValues, \
              outputKeys.DimValues, L, capacity, outputValues, a, ws, static_cast<uint32_t>(start))
#define ARES_HR_MERGE_ND(ND)                       \
  if (rwB == 4) ARES_HR_MERGE(ND, 4);              \
  else ARES_HR_MERGE(ND, 3);
    auto generic_merge = [&] {
      switch (all4 ? L.numDims : 0) {
        case 1: ARES_HR_MERGE_ND(1) break;
        case 2: ARES_HR_MERGE_ND(2) break;
        case 3: ARES_HR_MERGE_ND(3) break;
        case 4: ARES_HR_MERGE_ND(4) break;
        default: ARES_HR_MERGE(0, 3); break;
      }
    };
    if (leanMerge) {
      FusedPlanD none;
      memset(&none, 0, sizeof(none));
      rtc_merge_launch(leanMerge, none, inputKeys.DimValues, capacity, inputValues, static_cast<uint32_t>(start), outputKeys.DimValues,
                       capacity, outputValues, ws, stream, nullptr, result_slot());
    } else {
      generic_merge();
    }
    res = (leanMerge && result_slot()) ? read_result_pinned(stream) : read_result(ws, stream);
    mem_note_dim_rows(device, outputKeys, 0, res.groups);  // what this attempt emitted
    mem_note_write(device, outputValues, static_cast<size_t>(a.width) * res.groups);
    if (leanMerge && res.needGeneric && !res.overflow && !(grouped && res.stale)) {
      // a partition holds more groups than one LDS table: the generic multi-round merge over the same records
      hip_check(hipMemsetAsync(ws.outCount, 0, 4 * sizeof(uint32_t), stream), "hipMemsetAsync");
      if (outRanges) hip_check(hipMemsetAsync(outRanges, 0, kRangesBytes, stream), "hipMemsetAsync");
      generic_merge();
      res = read_result(ws, stream);
      mem_note_dim_rows(device, outputKeys, 0, res.groups);  // what this attempt emitted
      mem_note_write(device, outputValues, static_cast<size_t>(a.width) * res.groups);
    }
#undef ARES_HR_MERGE_ND
#undef ARES_HR_MERGE
    static const bool trace = getenv("ARES_HR_TRACE") != nullptr;  // diagnostics
    if (trace)
      fprintf(stderr, "hash_reduce_lds: length %d start %d rows %d streams %d capA %llu capB %u partBits %d -> groups %u overflow %u stale %u\n",
              length, start, rows, streams, static_cast<unsigned long long>(ws.capA), ws.capB, partBits, res.groups, res.overflow, res.stale);
    r.buf->mark_idle();  // read_result waited for the stream behind the last kernel that touches the workspace
    if (grouped && res.stale) {  // the input vectors are not what the previous merge wrote: the long way
      grouped_note_write(device, inputKeys.DimValues, static_cast<size_t>(L.rowBytes) * capacity);
      grouped = false;
      continue;
    }
    break;
  }
  if (res.overflow) {
    static bool told = false;
    if (!told) {
      told = true;
      fprintf(stderr, "libalgorithm: a hash-partition region overflowed (skewed hashes?): HashReduce falls back to the global table\n");
    }
    return -1;
  }
  if (outRanges) {
    GroupedState s{device, outputKeys.DimValues, outputValues, capacity, slots, a.width, static_cast<int>(res.groups),
                   partBits, outRangesRef};
    if (res.groups > 0) grouped_register(s);
  }
  return static_cast<int>(res.groups);
}


// The fused pipeline for an already built plan (shared by the extension entry point and by the
// in-ABI fusion of pending transforms into HashReduce, transform.hip).  Returns the number of groups
// or -1 when a region overflowed.
static void fused_note_first_batch(size_t shape, int groups) {
  if (!shape || groups < 0) return;
  std::lock_guard<std::mutex> lock(g_shapeMutex);
  if (g_firstBatchGroups.size() > 4096) g_firstBatchGroups.clear();
  g_firstBatchGroups[shape] = groups;
}

int fused_hash_reduce_run(int device, const FusedPlanD &plan, int batchRows, const DimensionVector &prevKeys,
                          const uint8_t *prevValues, int prevSize, const DimensionVector &outKeys, uint8_t *outValues,
                          const AggSpec &a, hipStream_t stream) {
  SlotWidths slots;
  if (!slot_widths(outKeys.NumDimsPerDimWidth, &slots)) return kFusedUnavailable;
  const int nd = slots.nd;
  // Narrow plans (a 1- / 2-byte dimension slot or source column) run on the kernels generated for their shape ONLY: the
  // precompiled generic scan and merge read 4-byte columns and write 4-byte slots.  Whenever a generated kernel is not
  // loaded yet (it is being compiled in the background) or the call needs something only the generic kernels do, the
  // call is declined BEFORE anything is launched and the caller takes the unfused sequence for this batch.
  const bool narrow = fused_plan_narrow(plan, nd);
  const int mw = plan.measureWidth;
  const int64_t length = static_cast<int64_t>(batchRows) + prevSize;
  if (length == 0) return 0;
  const int partBits = std::max(part_bits_for(length), narrow ? 2 : 0);  // (the generated merge needs >= 4 partitions)
  const int numParts = 1 << partBits;
  const size_t prevCapacity = static_cast<size_t>(prevKeys.VectorCapacity);
  const size_t outCapacity = static_cast<size_t>(outKeys.VectorCapacity);
  // what is known about the OUTPUT vectors before this call rewrites them: when they belong to this query (the Go host
  // ping-pongs two result buffers) their leading dimension rows already hold the query's groups, and their table image is
  // the one to write into
  // (declined before anything is touched: what is known about the output vectors stays known)
  if (narrow && !(batchRows > 0 && rtc_scan_available())) return kFusedUnavailable;
  GroupedState outOld;
  const bool outFound = image_enabled() && grouped_lookup(device, outKeys.DimValues, outValues, outCapacity, slots, mw, &outOld);
  grouped_note_write(device, outValues, static_cast<size_t>(mw) * outCapacity);  // (values first: unwritten ones die whole)
  grouped_note_write(device, outKeys.DimValues, slots.row_bytes() * outCapacity);
  GroupedState prev;
  const bool found = prevSize > 0 && grouped_lookup(device, prevKeys.DimValues, prevValues, prevCapacity, slots, mw, &prev) &&
                     prev.partBits == partBits && prev.size == prevSize;
  bool grouped = found && prev.ranges != nullptr;
  const std::shared_ptr<uint32_t> outRangesRef = grouped_enabled() ? take_ranges(device) : nullptr;
  uint32_t *outRanges = outRangesRef.get();
  MergeResult res{0, 0, 0, 0};
  // Which scan.  A query that already has more groups than an LDS table serves well goes straight to DIRECT
  // mode: every surviving row becomes a record (compact lines when the batch's chunks fit the row field), with
  // scan and merge compiled for this plan's shape (hr_rtc.hip).  Fewer groups: the TABLE-mode scan compiled for
  // the shape — LDS aggregation, one record per group and workgroup.  The adaptive generic kernel takes the first
  // batch of a shape that was never seen (nothing known yet), every plan the generator does not cover, and every
  // call whose specialised kernel is not loaded yet (it is compiled in the background, never inside a query).
  // The first batch of a query has no previous groups to judge by: it goes by what the first batch of
  // the same plan shape (expressions and divisors, not columns or comparison constants) produced the last time.
  hr::Widen widen;
  widen.mode = mw == 8 ? 1 : 0;
  widen.rk = plan.measure.f.rk;
  widen.dtype = plan.measureDtype;
  size_t shape = 0;
  int expected = prevSize;
  bool known = prevSize > 0;
  const bool rtc = batchRows > 0 && rtc_scan_available();
  if (rtc && prevSize == 0) {
    shape = std::hash<std::string>()(rtc_scan_source(plan, nd, 0));
    std::lock_guard<std::mutex> lock(g_shapeMutex);
    auto it = g_firstBatchGroups.find(shape);
    if (it != g_firstBatchGroups.end()) {
      expected = it->second;
      known = true;
    }
  }
  const int chunkTiles = compact_enabled() ? rtc_compact_chunk_tiles(batchRows, partBits) : 0;
  const bool compact = chunkTiles > 0;
  RtcKernel lean, table, tableMerge, narrowMerge;
  SlowScope slowWhole("fused_hash_reduce_run");
  // (ARES_LEAN_MIN_GROUPS=0 — tests — sends every batch, known shape or not, to the DIRECT kernels)
  const bool wantLean = rtc && expected >= lean_min_groups() && (known || lean_min_groups() <= 0);
  // (a narrow shape nobody has seen has no adaptive generic kernel to find its cardinality out with: the TABLE scan takes
  // it — right for any number of groups, rows that find the table full travel alone — and the next first batch knows)
  const bool wantTable = rtc && !wantLean && (known || narrow);
  if (wantLean) lean = rtc_scan_lookup(device, plan, nd, partBits, compact);
  else if (wantTable) table = rtc_table_scan_lookup(device, plan, nd, partBits, a, widen);
  // ---- table image (see "table images" above): 2 = the merge starts from the image the previous call left with the input
  // vectors; 1 = the ordinary specialised merge, which leaves an image for the next call.  DIRECT-mode batches only.
  int imageMode = 0;
  RtcKernel imageMerge;
  std::shared_ptr<uint8_t> imageOut;
  uint32_t knownOut = 0;
  if (image_enabled() && wantLean && lean) {
    if (found && prev.image) imageMode = 2;
    else if (prevSize == 0 || grouped) imageMode = 1;
    if (imageMode) imageMerge = rtc_merge_lookup(device, plan, nd, partBits, a, widen, compact, false, false, imageMode);
    if (!imageMerge) imageMode = 0;  // (being compiled in the background: the ordinary kernels this time)
  }
  if (imageMode == 2) {
    // The output buffer's leading rows are trusted as "the first outOld.size groups of prev" only when the state found there
    // is the very state prev was derived from (the Go host's ping-pong: the call before last wrote this buffer).  A state of
    // the same lineage that is NOT prev's parent — two results forked from one input, then one reduced into the other's
    // buffer — holds other groups behind the common prefix.
    const bool mine = outFound && outOld.lineage == prev.lineage && outOld.image && outOld.image != prev.image &&
                      outOld.partBits == partBits && outOld.size <= prevSize && prev.parentGen != 0 && outOld.gen == prev.parentGen;
    knownOut = mine ? static_cast<uint32_t>(outOld.size) : 0u;
    imageOut = mine ? outOld.image : take_image(device);
    // a partition's key and position planes are rewritten unless the output's image already holds this very set (equal
    // counts within one lineage); an image of unknown content starts with counts no partition can have
    if (imageOut && !mine) hip_check(hipMemsetAsync(image_counts(imageOut.get()), 0xFF, sizeof(uint32_t) * kMaxPartitions, stream), "hipMemsetAsync");
  } else {
    if (imageMode == 1) imageOut = take_image(device);
    // the previous result's measure rows are read by everything but an image-mode merge: write them if they are still
    // only defined by their image
    if (prevSize > 0) grouped_materialize_for_read(device, prevValues, static_cast<size_t>(mw) * prevSize);
  }
  if (imageMode && !imageOut) {  // no memory for an image: the ordinary merge (which reads the previous result's measure rows)
    imageMode = 0;
    knownOut = 0;
    if (prevSize > 0) grouped_materialize_for_read(device, prevValues, static_cast<size_t>(mw) * prevSize);
  }
  {
    static const bool trace = getenv("ARES_HR_TRACE") != nullptr;  // diagnostics
    if (trace)
      fprintf(stderr, "fused_hash_reduce_run: batch %d prev %d partBits %d | found %d (state: size %d partBits %d ranges %d image %d lazy %d) grouped %d "
                      "| lean %d table %d compact %d narrow %d -> image mode %d knownOut %u\n",
              batchRows, prevSize, partBits, found ? 1 : 0, found ? prev.size : -1, found ? prev.partBits : -1, found && prev.ranges ? 1 : 0,
              found && prev.image ? 1 : 0, found && prev.lazyValues ? 1 : 0, grouped ? 1 : 0, lean ? 1 : 0, table ? 1 : 0, compact ? 1 : 0,
              narrow ? 1 : 0, imageMode, knownOut);
  }
  if (narrow && !imageMode) {  // (both kernels are asked for before the call is declined: one round of background compilation, not two)
    narrowMerge = rtc_merge_lookup(device, plan, nd, partBits, a, widen, wantLean && compact, false,
                                   /*regionA=*/wantTable || (prevSize > 0 && !grouped));
    if (!narrowMerge) return kFusedUnavailable;  // still being compiled (or a shape the generator declines)
  }
  if (narrow && !(wantLean ? lean : table)) return kFusedUnavailable;
  bool launched = false;
  for (;;) {
    // narrow plans: the generated merge, which also takes region A — what the TABLE scan emits, and previous groups that are
    // not grouped by partition (re-partitioned below by the layout-generic kernel)
    const bool prevToA = narrow && prevSize > 0 && !grouped && imageMode != 2;
    if (narrow && !imageMode) {
      narrowMerge = rtc_merge_lookup(device, plan, nd, partBits, a, widen, lean && compact, false, /*regionA=*/table || prevToA);
      if (!narrowMerge) return launched ? -1 : kFusedUnavailable;
      if (table) tableMerge = narrowMerge;
    }
    launched = true;
    const int streams = batchRows > 0 ? (lean ? rtc_scan_grid(batchRows) : table ? 0 : grid_for(batchRows)) : 0;
    Regions r;
    {
      SlowScope slow("make_regions");
      // DIRECT mode writes region A only with previous groups that are not grouped by partition yet; sizing it for
      // the batch as well made the first DIRECT batch of a process allocate (and the driver clear) 2.7 GB more
      const int64_t rowsA = !lean ? length : (prevSize > 0 && !grouped && imageMode != 2) ? prevSize : 0;
      make_regions(r, partBits, rowsA, batchRows, streams, lean ? 4 : 3, stream,
                   !lean ? 0 : compact ? static_cast<int>(kCompactLineRecords) : 8);
    }
    Workspace &ws = r.ws;
    ws.widen = widen;
    ws.chunkRows = static_cast<uint32_t>(chunkTiles) * 4096u;
    ws.rowBase = static_cast<uint32_t>(prevSize);
    ws.prevRanges = (grouped && imageMode != 2) ? prev.ranges.get() : nullptr;  // (an image-mode merge has them in its image)
    ws.outRanges = outRanges;
    Workspace wsPrev = ws;  // previous groups that are not grouped by partition: TABLE-mode pass into region A
    wsPrev.streams = 0;
    // the specialised merge reads region B and grouped previous results only
    RtcKernel leanMerge = imageMode ? imageMerge
                          : narrow  ? (lean ? narrowMerge : nullptr)
                          : (lean && (prevSize == 0 || grouped)) ? rtc_merge_lookup(device, plan, nd, partBits, a, widen, compact) : nullptr;
    RtcImageArgs imageArgs{nullptr, nullptr, nullptr, nullptr, 0u};
    if (imageMode) {
      imageArgs.in = imageMode == 2 ? prev.image.get() : nullptr;
      imageArgs.inCount = imageMode == 2 ? image_counts(prev.image.get()) : nullptr;
      imageArgs.out = imageOut.get();
      imageArgs.outCount = image_counts(imageOut.get());
      imageArgs.knownOut = knownOut;
    }
    const RtcImageArgs *imagePtr = imageMode ? &imageArgs : nullptr;
    if (prevToA) {
      const DimLayoutD prevLayout = make_dim_layout(prevKeys.NumDimsPerDimWidth);
      const int64_t tiles = (static_cast<int64_t>(prevSize) + kTileRows - 1) / kTileRows;
      ARES_LAUNCH("hr_partition_kernel", hr_partition_kernel, static_cast<int>(tiles < 256 ? tiles : 256), kThreads, stream,
                  prevKeys.DimValues, prevLayout, prevCapacity, prevValues, a, prevSize, wsPrev);
    }
    // (the specialised merge writes every partition's range entry itself)
    if (outRanges && !leanMerge && !tableMerge) hip_check(hipMemsetAsync(outRanges, 0, kRangesBytes, stream), "hipMemsetAsync");
#define ARES_FUSED_CASE(ND)                                                                                            \
  case ND:                                                                                                             \
    if (prevSize > 0 && !grouped && !narrow && imageMode != 2) {                                                       \
      if (mw == 8)                                                                                                     \
        ARES_LAUNCH("hr_partition4_kernel", (hr_partition4_kernel<ND, 8>), grid_for(prevSize), kThreads, stream,        \
                    prevKeys.DimValues, prevCapacity, prevValues, 0u, a, prevSize, wsPrev, 0);                         \
      else                                                                                                             \
        ARES_LAUNCH("hr_partition4_kernel", (hr_partition4_kernel<ND, 4>), grid_for(prevSize), kThreads, stream,        \
                    prevKeys.DimValues, prevCapacity, prevValues, 0u, a, prevSize, wsPrev, 0);                         \
    }                                                                                                                  \
    if (batchRows > 0 && lean)                                                                                         \
      rtc_scan_launch(lean, plan, static_cast<uint32_t>(prevSize), batchRows, ws, stream);                             \
    else if (batchRows > 0 && table)                                                                                   \
      rtc_table_scan_launch(table, plan, static_cast<uint32_t>(prevSize), batchRows, ws, stream);                      \
    else if (batchRows > 0)                                                                                            \
      ARES_LAUNCH("hr_fused_scan_kernel", hr_fused_scan_kernel<ND>, streams, kThreads, stream, plan,                   \
                  static_cast<uint32_t>(prevSize), a, batchRows, ws);                                                  \
    if (leanMerge || tableMerge)                                                                                       \
      rtc_merge_launch(leanMerge ? leanMerge : tableMerge, plan, prevKeys.DimValues, prevCapacity, prevValues,         \
                       static_cast<uint32_t>(prevSize), outKeys.DimValues, outCapacity, outValues, ws, stream, imagePtr, \
                       result_slot());                                                                                 \
    else if (lean)                                                                                                     \
      ARES_LAUNCH("hr_fused_merge_kernel", (hr_fused_merge_kernel<ND, 4>), numParts, kThreads, stream, plan,            \
                  prevKeys.DimValues, prevCapacity, prevValues, static_cast<uint32_t>(prevSize), outKeys.DimValues,    \
                  outCapacity, outValues, a, ws);                                                                      \
    else                                                                                                               \
      ARES_LAUNCH("hr_fused_merge_kernel", (hr_fused_merge_kernel<ND, 3>), numParts, kThreads, stream, plan,            \
                  prevKeys.DimValues, prevCapacity, prevValues, static_cast<uint32_t>(prevSize), outKeys.DimValues,    \
                  outCapacity, outValues, a, ws);                                                                      \
    break;
    switch (nd) {
      ARES_FUSED_CASE(1) ARES_FUSED_CASE(2) ARES_FUSED_CASE(3) ARES_FUSED_CASE(4)
      default:  // five to eight dimensions: the generated kernels only (the plan counts as narrow: both were looked up above)
        if (batchRows > 0 && lean) rtc_scan_launch(lean, plan, static_cast<uint32_t>(prevSize), batchRows, ws, stream);
        else if (batchRows > 0 && table) rtc_table_scan_launch(table, plan, static_cast<uint32_t>(prevSize), batchRows, ws, stream);
        rtc_merge_launch(leanMerge ? leanMerge : tableMerge, plan, prevKeys.DimValues, prevCapacity, prevValues, static_cast<uint32_t>(prevSize),
                         outKeys.DimValues, outCapacity, outValues, ws, stream, imagePtr, result_slot());
        break;
    }
    {
      SlowScope slow("read_result");
      res = ((leanMerge || tableMerge) && result_slot()) ? read_result_pinned(stream) : read_result(ws, stream);
    }
    if (imageMode == 2) res.groups += static_cast<uint32_t>(prevSize);  // (the kernel counted the groups it appended)
    if (imageMode == 2) {  // new groups' dimension rows and the rows copied over; the measure rows stay unwritten
      if (res.groups > knownOut) mem_note_dim_rows(device, outKeys, knownOut, res.groups - knownOut);
    } else {
      mem_note_dim_rows(device, outKeys, 0, res.groups);  // what this attempt emitted
      mem_note_write(device, outValues, static_cast<size_t>(mw) * res.groups);
    }
    if ((narrow || imageMode == 2) && res.needGeneric) {
      // more groups in a partition than one table: only the generic merge takes rounds (the caller runs the unfused
      // sequence, which reads the previous result's vectors: complete, whatever this attempt wrote into the output)
      r.buf->mark_idle();
      return -1;
    }
    if (imageMode == 1 && res.needGeneric) imageMode = 0;  // (the generic merge below leaves no image)
    if (leanMerge && res.needGeneric && !res.overflow && !(grouped && res.stale)) {
      // a partition holds more groups than one LDS table: the generic multi-round merge over the same records
      hip_check(hipMemsetAsync(ws.outCount, 0, 4 * sizeof(uint32_t), stream), "hipMemsetAsync");
      if (outRanges) hip_check(hipMemsetAsync(outRanges, 0, kRangesBytes, stream), "hipMemsetAsync");
      switch (nd) {
#define ARES_FUSED_REMERGE(ND)                                                                                         \
  case ND:                                                                                                             \
    ARES_LAUNCH("hr_fused_merge_kernel", (hr_fused_merge_kernel<ND, 4>), numParts, kThreads, stream, plan,              \
                prevKeys.DimValues, prevCapacity, prevValues, static_cast<uint32_t>(prevSize), outKeys.DimValues,      \
                outCapacity, outValues, a, ws);                                                                        \
    break;
        ARES_FUSED_REMERGE(1) ARES_FUSED_REMERGE(2) ARES_FUSED_REMERGE(3) ARES_FUSED_REMERGE(4)
#undef ARES_FUSED_REMERGE
      }
      res = read_result(ws, stream);
      mem_note_dim_rows(device, outKeys, 0, res.groups);  // what this attempt emitted
      mem_note_write(device, outValues, static_cast<size_t>(mw) * res.groups);
    }
#undef ARES_FUSED_CASE
    r.buf->mark_idle();  // read_result waited for the stream behind the last kernel that touches the workspace
    if (grouped && res.stale) {
      grouped_note_write(device, prevKeys.DimValues, slots.row_bytes() * prevCapacity);
      grouped = false;
      imageMode = 0;
      continue;
    }
    // a record stream overflowed (rows that arrive sorted fill the partitions of a workgroup's chunk unevenly): more room, again
    if (res.overflow && lean && grow_record_stream_slack()) continue;
    break;
  }
  if (res.overflow) {
    return -1;
  }
  fused_note_first_batch(shape, static_cast<int>(res.groups));
  if (shape && !lean && !table) {
    // the next first batch of this shape takes the specialised kernels: have them built now (in the background)
    if (static_cast<int>(res.groups) >= lean_min_groups()) {
      (void)rtc_scan_lookup(device, plan, nd, partBits, compact);
      (void)rtc_merge_lookup(device, plan, nd, partBits, a, widen, compact);
    } else {
      (void)rtc_table_scan_lookup(device, plan, nd, partBits, a, widen);
    }
  }
  if (outRanges) {
    GroupedState s{device, outKeys.DimValues, outValues, outCapacity, slots, mw, static_cast<int>(res.groups), partBits, outRangesRef};
    if (imageMode == 2) {  // rows in order of first appearance: no ranges; the measure rows are the image's value plane
      s.ranges = nullptr;
      s.image = imageOut;
      s.lineage = prev.lineage;
      s.gen = ++g_generation;
      s.parentGen = prev.gen;
      s.lazyValues = true;
    } else if (imageMode == 1) {
      s.image = imageOut;
      s.lineage = ++g_lineage;
      s.gen = ++g_generation;
    }
    if (res.groups > 0) grouped_register(s);
  }
  return static_cast<int>(res.groups);
}


// ---- fused extension: host side ---------------------------------------------------------------------
namespace {

struct NotFusable : std::runtime_error {
  explicit NotFusable(const std::string &why) : std::runtime_error("not fusable: " + why) {}
};

// Column slots: dimension d -> slot d, measure -> slot ND (a column used twice is simply loaded
// twice; the second load hits L1).  Filters reuse a slot that already holds their column, or take
// the one spare slot ND + 1.
int fused_column(FusedPlanD &plan, const FastOperands &f, int maxCols, bool reuse) {
  if (reuse)
    for (int c = 0; c < plan.numCols; c++)
      if (plan.cols[c].vals == f.vals && plan.cols[c].nulls == f.nulls && plan.cols[c].bitOff == f.bitOff &&
          plan.cols[c].step == static_cast<uint32_t>(f.step ? f.step : 4))
        return c;
  if (plan.numCols >= maxCols) throw NotFusable("too many distinct columns");
  FusedColumn &col = plan.cols[plan.numCols];
  col.vals = f.vals;
  col.nulls = f.nulls;
  col.bitOff = f.bitOff;
  col.step = static_cast<uint32_t>(f.step ? f.step : 4);
  return plan.numCols++;
}

void fused_expr(const AresFusedExpr &e, bool compareOnly, int batchRows, hipStream_t stream, FusedPlanD &plan, int maxCols,
                FusedExpr &out) {
  if (e.arity != 1 && e.arity != 2) throw NotFusable("arity");
  if (e.lhs.Type != VectorPartyInput || (e.arity == 2 && e.rhs.Type != ConstantInput))
    throw NotFusable("operands must be a main-table column and a constant");
  if (static_cast<int64_t>(e.lhs.Vector.VP.Length) < batchRows) throw NotFusable("column shorter than the batch");
  InputVector ins[2] = {e.lhs, e.arity == 2 ? e.rhs : e.lhs};
  EvalParams p;
  CallTemps temps;
  build_params(ins, e.arity, stream, nullptr, nullptr, 0, e.functor, p, temps);
  FastOperands f;
  if (!fast_operands(p, f, compareOnly)) throw NotFusable("expression shape");
  out.col = fused_column(plan, f, maxCols, compareOnly);
  out.f = f;
  out.f.vals = nullptr;
  out.f.nulls = nullptr;
  out.outKind = e.outType == Int32 ? K_I32 : e.outType == Uint32 ? K_U32 : K_F32;
}

int fused_filter_hash_reduce(int device, const AresFusedQuery &q, int batchRows, const DimensionVector &prevKeys,
                             uint8_t *prevValues, int prevSize, const DimensionVector &outKeys, uint8_t *outValues,
                             hipStream_t stream) {
  if (q.numDims < 1 || q.numDims > kGenericFusedDims) throw NotFusable("1..4 dimensions");
  if (q.numFilters < 0 || q.numFilters > kExtensionFilters) throw NotFusable("at most 4 filters");
  const int nd = q.numDims;
  for (int k = 0; k < NUM_DIM_WIDTH; k++)
    if (outKeys.NumDimsPerDimWidth[k] != (k == 2 ? nd : 0) || (prevSize > 0 && prevKeys.NumDimsPerDimWidth[k] != (k == 2 ? nd : 0)))
      throw NotFusable("dimension vector layout");
  const int mt = q.measure.outType;
  if (!(mt == Int32 || mt == Uint32 || mt == Float32 || mt == Int64 || mt == Float64)) throw NotFusable("measure type");
  const int mw = (mt == Int64 || mt == Float64) ? 8 : 4;
  const AggSpec a = make_agg_spec(q.aggFunc, mw);
  if (!hash_reduce_lds_supported(a)) throw NotFusable("aggregate");
  if (batchRows < 0 || prevSize < 0 || static_cast<int64_t>(batchRows) + prevSize > INT32_MAX) throw NotFusable("size");
  if (static_cast<int64_t>(outKeys.VectorCapacity) < static_cast<int64_t>(batchRows) + prevSize)
    throw std::invalid_argument("outKeys.VectorCapacity < prevSize + batchRows");

  FusedPlanD plan;
  memset(&plan, 0, sizeof(plan));
  const int maxCols = nd + 2;
  plan.numFilters = q.numFilters;
  for (int d = 0; d < nd; d++) {
    const int t = q.dims[d].outType;
    if (!(t == Int32 || t == Uint32 || t == Float32)) throw NotFusable("dimension type");
    fused_expr(q.dims[d], false, batchRows, stream, plan, maxCols, plan.dims[d]);
  }
  fused_expr(q.measure, false, batchRows, stream, plan, maxCols, plan.measure);
  for (int k = 0; k < q.numFilters; k++) fused_expr(q.filters[k], true, batchRows, stream, plan, maxCols, plan.filters[k]);
  plan.measureDtype = mt;
  plan.measureWidth = mw;
  plan.identity = identity_bits(q.aggFunc, mt);
  if (mw == 8 && plan.identity != 0) throw NotFusable("8-byte aggregate with a non-zero identity");

  const int groups = fused_hash_reduce_run(device, plan, batchRows, prevKeys, prevValues, prevSize, outKeys, outValues, a, stream);
  if (groups == kFusedUnavailable)
    throw NotFusable("the kernels generated for this plan's shape are not loaded yet (compiled in the background), or the shape is outside what they cover; "
                     "run the unfused sequence for this batch");
  if (groups < 0) throw NotFusable("a hash partition overflowed (skewed hashes); run the unfused sequence");
  return groups;
}

}  // namespace

}  // namespace ares

extern "C" CGoCallResHandle AresFusedFilterHashReduce(const AresFusedQuery *query, int batchRows, DimensionVector prevKeys,
                                                      uint8_t *prevValues, int prevSize, DimensionVector outKeys,
                                                      uint8_t *outValues, void *cudaStream, int device) {
  ARES_ABI_BEGIN(device)
  if (!query) throw std::invalid_argument("null query");
  resHandle.res = ares::int_result(ares::fused_filter_hash_reduce(device, *query, batchRows, prevKeys, prevValues, prevSize, outKeys,
                                                                  outValues, reinterpret_cast<hipStream_t>(cudaStream)));
  ARES_ABI_END("AresFusedFilterHashReduce")
}
