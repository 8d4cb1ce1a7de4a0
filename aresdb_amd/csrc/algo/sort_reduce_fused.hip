// Sort + Reduce, hash-keyed: the group-by of the reference's DEFAULT aggregation path without sorting rows.
//
// Reference: query/sort_reduce.cu:118-133 (hash every row of the dimension vector with murmur3_x64_128, stable sort of the
// index vector by the 64-bit hash), :135-249 (reduce_by_key over runs of equal hashes; the representative of a run is its
// first index; dimensions of the representatives gathered into the output vector) — what the Go host runs for every
// aggregate but SUM_SIGNED / SUM_FLOAT, and for those too unless enable_hash_reduction is set
// (query/aql_context.go:426-434, config/ares.yaml:11).  COUNT(*) — SUM over the literal 1 — always goes this way.
//
// What a caller can observe of Sort + Reduce over rows [0, prev) (the previous result) + [prev, prev + n) (the batch):
//   * the output groups in ASCENDING order of the 64-bit row hash, one group per distinct hash;
//   * each group's dimension row = that of its lowest-indexed row (stable sort of an iota index vector);
//   * each group's value = op over its rows' values (integer aggregates: independent of the order).
// Sorting 40-60 M surviving ROWS per batch to find 70 k - 2 M groups is what made this path 6-7 x slower than HashReduce
// (4 radix passes + hash + reduce + compaction + materialised dimension rows: 4.4 ms per 64 Mi-row batch).  Here the
// GROUPS are ordered instead:
//   1. sr_prev_kernel      hashes the previous result's rows (region A records, one cursor per partition);
//   2. sr_scan_rtc         (hr_rtc.hip, SCAN_SORT64) the fused scan over the batch's source columns — filters replayed from
//                          the journal, dimensions and measure evaluated on the fly — writes one 16-byte record
//                          {row, hash >> 32, carried measure, (u32)hash} per surviving row into the stream of the partition
//                          the TOP bits of the hash select: a partition is a contiguous range of the sorted order;
//   3. sr_merge_kernel     one workgroup per partition aggregates its records in an LDS table keyed by the 64-bit hash whose
//                          home slots are MONOTONE in the key (home = the key's bits below the partition's, scaled to the
//                          table) with linear probing and no wrap-around: every cluster (run of occupied slots) holds
//                          exactly the keys whose homes lie in it, all clusters before it hold smaller keys.  Ranking an
//                          entry = occupied slots before its cluster + keys of the cluster below its own — no sort pass;
//   4. sr_emit_kernel      prefix over the partitions' group counts, dimension rows of the representatives gathered (previous
//                          result) or re-evaluated from the source columns (batch rows), values stored: ascending hash order.
// Float sums keep the real sort (their order of additions is observable), as does anything this path declines: more
// groups than the partitions' tables hold, a skewed partition stream, a record whose hash equals the table's "empty" word.
//
// The same over rows that EXIST (a joined column, a generic expression, an eager host — or what the scan-fed path above
// declined, after its transforms were launched): fused_sort_reduce_vectors at the end of this file, the "wide layout" —
// sr_vector_scan_rtc into <= 512 level-1 partitions, sr_count_kernel / sr_prefix_kernel / sr_split_kernel into up to 2^17
// partitions of about a thousand entries in runs of exactly their size, sr_merge_kernel<WIDE> (2048-slot tables, four
// workgroups of 256 lanes per CU), sr_prefix_kernel, sr_emit_kernel<WIDE>.  50 M groups per call at C4's size.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <mutex>
#include <vector>

#include "aggregate.hpp"
#include "common.hpp"
#include "device_model.hpp"
#include "dim_layout.hpp"
#include "fast_eval.hpp"
#include "hash_reduce_lds.hpp"
#include "hr_kernels.hpp"
#include "hr_rtc.hpp"
#include "sort_reduce_fused.hpp"

namespace ares {

namespace {
using hr::kMaxPartitions;
using hr::kMaxStreams;
using hr::kThreads;

constexpr uint64_t kEmptyKey = ~0ull;
constexpr uint32_t kNoRow = 0xFFFFFFFFu;
// table slots per partition: 16 bytes per slot with 4-byte values (128 KB), 20 with 8-byte values (135 KB) — what a
// workgroup's 160 KB of LDS leave beside the per-wavefront queues of records that miss their home bucket.  The last kTail
// slots are overflow room for the clusters at the table's end (no wrap-around: order is position)
constexpr uint32_t kQueueCap = 80, kQueueDrain = 16;  // per wavefront: drained from kQueueDrain entries on (a push adds <= 64)
// WIDE (Sort + Reduce over materialised vectors, fused_sort_reduce_vectors below: results of tens of millions of groups):
// many small partitions instead — 2048 slots (37 KB with the queues: four workgroups per CU), what a partition's fixed
// costs (clearing the table, the ordering pass over its slots) are proportional to
template <int VW, bool WIDE = false>
struct Table {
  static constexpr int kSlots = WIDE ? 2048 : (VW == 4 ? 8192 : 6912);
  static constexpr int kTail = WIDE ? 128 : 256;
  static constexpr int kHomes = kSlots - kTail;
  static constexpr int kLanes = WIDE ? 256 : kThreads;  // workgroup size (wide: four wavefronts, so that a CU holds four workgroups)
  static constexpr int kPerLane = (kSlots + kLanes - 1) / kLanes;
  static constexpr int kMaxGroups = kHomes * 13 / 16;  // beyond ~0.8 the clusters (and the ranking walks) grow quickly
  static constexpr int kStage = WIDE ? (kMaxGroups + 7) / 8 * 8 : kSlots;  // staging entries per partition
};
constexpr int kWideMaxPartBits = 17, kWideEntries = 1024;  // partitions of the wide layout: entries / 1024 (up to 2^17)

struct SrArgs {
  // records
  const uint4 *recA;
  uint32_t *cursorsA;
  uint64_t capA;
  const uint4 *recB;
  const uint32_t *countsB;
  uint32_t capB;
  int streams, partBits;
  // previous result + batch
  const uint8_t *dimIn;
  const uint8_t *inValues;
  size_t inCapacity;
  uint32_t prevSize;
  // how a batch record's carried 4 bytes become the value: widen (hr::Widen) or the constant
  hr::Widen widen;
  int constMeasure;
  uint64_t constBits;
  AggSpec agg;
  // staging: per partition Table::kSlots entries {row, value lo, value hi, 0}, ordered; group count per partition
  uint4 *staging;
  uint32_t *partCount;
  uint32_t *flags;  // [0] groups (emit), [1] a stream / region / table overflowed, [2] a hash equals the empty word
  uint32_t maxGroups;  // groups a partition's table takes (Table::kMaxGroups; ARES_SR_MAX_GROUPS lowers it: tests)
  // emission
  uint8_t *dimOut;
  uint8_t *outValues;
  // the groups' keys: stageKeys [partition][Table::kSlots] in the merge's order; keysOut [group] as emitted (what the next
  // call of the query reads as prevKeys: the previous result's row hashes, ascending — null: not known, sr_prev_kernel)
  uint64_t *stageKeys;
  uint64_t *keysOut;
  const uint64_t *prevKeys;
  // wide layout (fused_sort_reduce_vectors): region B holds ONE run per partition (recB + offsetsB[p], countsB[p] records,
  // written by sr_split_kernel); prevBounds[p .. p + 1] = the partition's range of prevKeys; partBase[p] = groups before
  // partition p (sr_prefix_kernel); copyAll: every representative's dimension row is copied from dimIn (no plan)
  const uint32_t *prevBounds;
  const uint32_t *offsetsB;
  const uint32_t *partBase;
  int copyAll;
  int gatherValues;  // (wide, 8-byte values: a record has no room for one — the merge reads inValues[row])
  uint64_t *phases;  // ARES_HR_PHASES=1 (diagnostics): six time stamps per partition (100 MHz clock), else null
};

__device__ __forceinline__ uint32_t sr_partition(uint64_t key, int pb) { return pb ? static_cast<uint32_t>(key >> (64 - pb)) : 0u; }

// ---- 1. previous result -> region A ---------------------------------------------------------------------------------
// The rows are (in the Go host's flow) the previous Reduce's output, ascending by hash: consecutive lanes mostly share a
// partition, so each wavefront reserves once per distinct partition it holds.
__global__ __launch_bounds__(256) void sr_prev_kernel(SrArgs m, DimLayoutD L, uint4 *recA) {
  const int lane = threadIdx.x & 63;
  const int64_t n = m.prevSize;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * 256; base < n; base += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t i = base + threadIdx.x;
    const bool have = i < n;
    uint64_t key = 0;
    if (have) {
      Murmur128Stream ms(0);
      hash_dim_row(ms, m.dimIn, L, m.inCapacity, static_cast<uint32_t>(i));
      key = ms.finish();
    }
    const uint32_t p = sr_partition(key, m.partBits);
    uint64_t todo = __ballot(have);
    uint32_t at = 0;
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const uint32_t lp = __shfl(p, leader);
      const uint64_t same = __ballot(have && p == lp) & todo;
      uint32_t start = 0;
      if (lane == leader) start = atomicAdd(m.cursorsA + lp, static_cast<uint32_t>(__popcll(same)));
      start = __shfl(start, leader);
      if ((same >> lane) & 1ull) at = start + static_cast<uint32_t>(__popcll(same & ((1ull << lane) - 1)));
      todo &= ~same;
    }
    if (have) {
      if (at < m.capA) recA[static_cast<uint64_t>(p) * m.capA + at] = make_uint4(static_cast<uint32_t>(i), static_cast<uint32_t>(key >> 32), 0u, static_cast<uint32_t>(key));
      else m.flags[1] = 1u;
      if (key == kEmptyKey) m.flags[2] = 1u;
    }
  }
}

// ---- 3. per-partition aggregation + ordering ------------------------------------------------------------------------
template <int VW>
struct Slots {
  using V = typename std::conditional<VW == 4, uint32_t, uint64_t>::type;
};

template <int VW>
__device__ __forceinline__ void sr_aggregate(typename Slots<VW>::V *slot, uint64_t bits, const AggSpec &a) {
  if constexpr (VW == 8) {
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(slot), static_cast<unsigned long long>(bits), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
    if (a.vtype == V_U32) {
      const uint32_t x = static_cast<uint32_t>(bits);
      if (a.op == OP_SUM) __hip_atomic_fetch_add(slot, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (a.op == OP_MIN) __hip_atomic_fetch_min(slot, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_max(slot, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      int32_t *ps = reinterpret_cast<int32_t *>(slot);
      const int32_t x = static_cast<int32_t>(static_cast<uint32_t>(bits));
      if (a.op == OP_SUM) __hip_atomic_fetch_add(ps, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (a.op == OP_MIN) __hip_atomic_fetch_min(ps, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_max(ps, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

template <int VW, bool WIDE>
__global__ __launch_bounds__((Table<VW, WIDE>::kLanes)) void sr_merge_kernel(SrArgs m) {
  using T = Table<VW, WIDE>;
  using V = typename Slots<VW>::V;
  __shared__ __attribute__((aligned(16))) uint64_t sKeys[T::kSlots];
  __shared__ uint32_t sRows[T::kSlots + 1];  // (+ 1: the spare slot records that miss their home bucket aim their atomics at)
  __shared__ V sVals[T::kSlots + 1];
  __shared__ uint32_t sRun[WIDE ? 1 : kMaxStreams];
  __shared__ uint4 sQueue[T::kLanes / 64][kQueueCap];
  __shared__ uint32_t sWave[T::kLanes / 64];
  __shared__ uint32_t sClaims, sBad;
  const int p = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pb = m.partBits;
  const int numParts = 1 << pb;
  const AggSpec a = m.agg;
  if (m.phases && tid == 0) m.phases[static_cast<size_t>(p) * 8 + 5] = wall_clock64();
  // wide layout: a partition is a thousand entries — its life is a chain of memory latencies (bounds -> hashes and values of
  // the previous groups; count and offset -> records).  Everything a lane needs first is requested here, before the table is
  // cleared: two previous groups per lane, the wavefront's first chunk of records.
  uint32_t wFrom = 0, wTo = 0, wCnt = 0;
  const uint4 *wRun = m.recB;
  uint64_t wKey[2] = {0, 0}, wVal[2] = {0, 0};
  uint4 wRec[4];
  if constexpr (WIDE) {
    if (m.prevBounds) {
      wFrom = m.prevBounds[p];
      wTo = m.prevBounds[p + 1];
    }
    wCnt = m.countsB[p];
    wRun = m.recB + m.offsetsB[p];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint32_t i = wFrom + static_cast<uint32_t>(k) * T::kLanes + tid;
      if (i < wTo) {
        wKey[k] = m.prevKeys[i];
        wVal[k] = load_value_bits(m.inValues, a, i);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t i = static_cast<uint32_t>(wave) * 256u + static_cast<uint32_t>(k) * 64u + lane;
      wRec[k] = *(i < wCnt ? wRun + i : m.recB);
      wRec[k].x = i < wCnt ? wRec[k].x : kNoRow;
    }
  }
  for (int s = tid; s < T::kSlots; s += T::kLanes) {
    sKeys[s] = kEmptyKey;
    sRows[s] = kNoRow;
    sVals[s] = static_cast<V>(a.identity);
  }
  if (tid == 0) {
    sRows[T::kSlots] = kNoRow;
    sVals[T::kSlots] = static_cast<V>(a.identity);
  }
  if constexpr (!WIDE) {
    if (tid < m.streams) sRun[tid] = m.countsB[static_cast<uint64_t>(tid) * numParts + p];
  }
  if (tid == 0) { sClaims = 0; sBad = 0; }
  auto stamp = [&](int k) {
    if (m.phases && tid == 0) m.phases[static_cast<size_t>(p) * 8 + k] = wall_clock64();
  };
  stamp(0);
  __syncthreads();
  stamp(1);

  // The table is probed by BUCKETS of four keys (32 bytes: two 16-byte LDS reads): home bucket = the 32 key bits right below
  // the partition's, scaled to the home buckets — monotone in the key; a key sits in the first free slot at or behind its
  // home bucket's first (claims only ever turn the LOWEST empty slot of a bucket into a key, buckets are walked upwards, no
  // wrap-around), so every slot between a key's home and its place is occupied: what the ordering below goes by.
  // With one key per probe the longest probe sequence among 64 lanes paced every wavefront (253 of a partition's 300 us at
  // 4.5 k groups); a record meets its group in its home bucket ~93 % of the time.
  constexpr uint32_t kHomeBuckets = T::kHomes / 4, kBuckets = T::kSlots / 4;
  auto home_bucket = [&](uint32_t hi, uint32_t lo) { return __umulhi(pb ? ((hi << pb) | (lo >> (32 - pb))) : hi, kHomeBuckets); };
  // One probe of bucket b for key (hi, lo): returns the key's slot, or -1 (the caller looks at bucket `b` again — it may have
  // been advanced, or a claim was lost to another lane, maybe for this very key), or -2 (ran off the table / table full).
  auto probe = [&](uint32_t &b, uint32_t hi, uint32_t lo) -> int {
    const uint4 *bk = reinterpret_cast<const uint4 *>(sKeys + 4u * b);
    const uint4 u = bk[0], v = bk[1];
    const bool m0 = u.x == lo && u.y == hi, m1 = u.z == lo && u.w == hi, m2 = v.x == lo && v.y == hi, m3 = v.z == lo && v.w == hi;
    if (m0 || m1 || m2 || m3) return static_cast<int>(4u * b + (m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u));
    const bool e0 = (u.x & u.y) == 0xFFFFFFFFu, e1 = (u.z & u.w) == 0xFFFFFFFFu, e2 = (v.x & v.y) == 0xFFFFFFFFu, e3 = (v.z & v.w) == 0xFFFFFFFFu;
    if (e0 || e1 || e2 || e3) {  // a group that is new in this partition: rare once the groups exist
      const uint32_t slot = 4u * b + (e0 ? 0u : e1 ? 1u : e2 ? 2u : 3u);
      unsigned long long expected = kEmptyKey;
      if (__hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned long long *>(sKeys + slot), &expected,
                                               (static_cast<unsigned long long>(hi) << 32) | lo, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP)) {
        if (__hip_atomic_fetch_add(&sClaims, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= m.maxGroups)
          __hip_atomic_store(&sBad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return static_cast<int>(slot);
      }
      return -1;  // lost the slot: the same bucket again (the winner may be this very key)
    }
    if (++b >= kBuckets) {  // ran off the tail: more groups than the table orders
      __hip_atomic_store(&sBad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return -2;
    }
    return -1;
  };
  auto settle = [&](int slot, uint32_t row, uint64_t value) {
    __hip_atomic_fetch_min(sRows + slot, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    sr_aggregate<VW>(sVals + slot, value, a);
  };
  auto insert = [&](uint32_t row, uint32_t hi, uint32_t lo, uint64_t value) {  // the general loop, one record
    if ((hi & lo) == 0xFFFFFFFFu) {  // the table's empty word: the caller takes the real sort
      m.flags[2] = 1u;
      return;
    }
    uint32_t b = home_bucket(hi, lo);
    int slot = -1;
    while (slot == -1) slot = probe(b, hi, lo);
    if (slot >= 0) settle(slot, row, value);
  };

  // ---- previous groups.  Their row hashes are known (the query's previous Reduce left them, ascending, beside its result):
  // the partition's rows are the range of hashes that start with its bits — two binary searches, no hashing, no region A
  if (WIDE) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint32_t i = wFrom + static_cast<uint32_t>(k) * T::kLanes + tid;
      if (i < wTo) insert(i, static_cast<uint32_t>(wKey[k] >> 32), static_cast<uint32_t>(wKey[k]), wVal[k]);
    }
    for (uint32_t i = wFrom + 2u * T::kLanes + tid; i < wTo; i += T::kLanes) {
      const uint64_t key = m.prevKeys[i];
      insert(i, static_cast<uint32_t>(key >> 32), static_cast<uint32_t>(key), load_value_bits(m.inValues, a, i));
    }
  } else if (m.prevKeys) {
    auto lower = [&](uint64_t bound) {
      uint32_t lo = 0, hi = m.prevSize;
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (m.prevKeys[mid] < bound) lo = mid + 1; else hi = mid;
      }
      return lo;
    };
    // (wide layout: 2^17 partitions would each walk the array with two chains of dependent loads — sr_bounds_kernel did)
    const uint32_t from = m.prevBounds ? m.prevBounds[p] : pb ? lower(static_cast<uint64_t>(p) << (64 - pb)) : 0u;
    const uint32_t to = m.prevBounds ? m.prevBounds[p + 1] : (pb && p + 1 < numParts) ? lower(static_cast<uint64_t>(p + 1) << (64 - pb)) : m.prevSize;
    for (uint32_t i = from + tid; i < to; i += T::kLanes) {
      const uint64_t key = m.prevKeys[i];
      insert(i, static_cast<uint32_t>(key >> 32), static_cast<uint32_t>(key), load_value_bits(m.inValues, a, i));
    }
  } else if (m.recA)
  // ---- ... or not (region A, written by sr_prev_kernel): the value is read from the previous result's measure vector
  {
    const uint32_t cursor = m.cursorsA[p];
    const uint32_t nA = cursor < m.capA ? cursor : static_cast<uint32_t>(m.capA);
    const uint4 *recA = m.recA + static_cast<uint64_t>(p) * m.capA;
    for (uint32_t i = tid; i < nA; i += T::kLanes) {
      const uint4 r = recA[i];
      insert(r.x, r.y, r.w, load_value_bits(m.inValues, a, r.x));
    }
  }
  if (m.phases) __syncthreads();
  stamp(2);
  // ---- the batch's records (region B): every wavefront streams whole runs, four records per lane in flight.  Round one
  // looks at every record's home bucket with straight-line code (no claim, no advance); what is left — the group lives
  // further on, or is new: a few lanes per segment — goes through the probe loop.
  {
    const uint4 *pad = m.recB;  // (any readable address: lanes past a chunk's end load it and ignore it)
    struct Chunk {
      const uint4 *ptr;
      uint32_t n;  // records (0: no chunk left)
    };
    int g = wave;
    uint32_t off = 0;
    auto next = [&]() -> Chunk {
      if constexpr (WIDE) return Chunk{pad, 0u};
      while (g < m.streams) {
        const uint32_t cnt = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(sRun[g])));
        if (off < cnt) {
          Chunk c{m.recB + (static_cast<uint64_t>(g) * numParts + p) * m.capB + off, cnt - off < 256u ? cnt - off : 256u};
          off += 256u;
          return c;
        }
        g += T::kLanes / 64;
        off = 0;
      }
      return Chunk{pad, 0u};
    };
    auto load = [&](uint4(&r)[4], const Chunk &c) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t i = static_cast<uint32_t>(k) * 64u + lane;
        r[k] = *(i < c.n ? c.ptr + i : pad);
        r[k].x = i < c.n ? r[k].x : kNoRow;  // (past the chunk's end: no record)
      }
    };
    uint4 *queue = sQueue[wave];
    uint32_t qn = 0;  // (wave-uniform)
    auto drain = [&](uint32_t first, uint32_t count) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (static_cast<uint32_t>(lane) < count) {
        const uint4 q = queue[first + lane];
        insert(q.x, q.y, q.w, m.constMeasure ? m.constBits : m.gatherValues ? load_value_bits(m.inValues, a, q.x) : hr::widen_value(m.widen, q.z));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto consume = [&](uint4(&r)[4]) {
      uint32_t pend = 0, bkt[4];
      // all four home buckets are read before the first compare (eight independent 16-byte LDS reads in flight), and the
      // atomics are issued unconditionally — a record that does not meet its group at home aims them at a spare slot behind
      // the table (identity / no-row operands: the slot's contents never matter) — so that no branch separates the four
      // records' LDS round trips: the waves were parked on them half of the time
      uint4 u[4], v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        bkt[k] = home_bucket(r[k].y, r[k].w);
        const uint4 *bk = reinterpret_cast<const uint4 *>(sKeys + 4u * bkt[k]);
        u[k] = bk[0];
        v[k] = bk[1];
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool valid = r[k].x != kNoRow;  // (row ~0: the padding of a stream's last line, or no record at all)
        const uint32_t hi = r[k].y, lo = r[k].w;
        const bool m0 = u[k].x == lo && u[k].y == hi, m1 = u[k].z == lo && u[k].w == hi, m2 = v[k].x == lo && v[k].y == hi, m3 = v[k].z == lo && v[k].w == hi;
        const bool hit = valid && (m0 || m1 || m2 || m3) && (hi & lo) != 0xFFFFFFFFu;
        const uint32_t slot = hit ? 4u * bkt[k] + (m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u) : static_cast<uint32_t>(T::kSlots);
        const uint64_t value = m.constMeasure ? m.constBits : (WIDE && m.gatherValues) ? (valid ? load_value_bits(m.inValues, a, r[k].x) : 0ull) : hr::widen_value(m.widen, r[k].z);
        // (the group's lowest row is settled after its first few records: a plain read tells the rest they need no atomic)
        if (hit && r[k].x < sRows[slot]) __hip_atomic_fetch_min(sRows + slot, r[k].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        sr_aggregate<VW>(sVals + slot, hit ? value : a.identity, a);
        pend |= (valid && !hit ? 1u : 0u) << k;
      }
      // Records that did not meet their group at home (it lives further on, or is new: a few lanes per segment) are queued
      // per wavefront in LDS and taken through the general probe loop up to 64 at a time, every lane busy: run where they
      // occur, that loop executed for a handful of lanes after nearly every segment — two thirds of the kernel's vector
      // instructions (the kernel is bound by their issue: 181 per 64 records, measured with SQ_INSTS_VALU).
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool pk = (pend >> k) & 1u;
        const uint64_t pm = __ballot(pk);
        if (pm) {
          if (pk) queue[qn + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(pm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(pm), 0u))] = r[k];
          qn += static_cast<uint32_t>(__popcll(pm));
          if (qn >= kQueueDrain) {
            const uint32_t take = qn < 64u ? qn : 64u;
            qn -= take;
            drain(qn, take);
          }
        }
      }
    };
    // Small batches (live batches: 2 Mi rows = 512 tiles, two per scanning workgroup) leave every partition a few hundred
    // runs of a dozen records — one or two lines each.  Walked run by run, sixteen runs per wavefront, that is a chain of
    // dependent loads (113 us of a partition's 155 at 2 Mi rows): when no run is longer than four lines, the first L lines of
    // EVERY run are fetched at once instead (L = lines of the longest run), four 16-byte loads per lane in flight.
    uint32_t maxRun = 0;
    for (int gg = lane; !WIDE && gg < m.streams; gg += 64) maxRun = sRun[gg] > maxRun ? sRun[gg] : maxRun;
#pragma unroll
    for (int off2 = 32; off2 > 0; off2 >>= 1) {
      const uint32_t t = static_cast<uint32_t>(__shfl_xor(static_cast<int>(maxRun), off2));
      maxRun = t > maxRun ? t : maxRun;
    }
    const uint32_t lpr = (maxRun + 7u) / 8u;  // lines of the longest run
    uint4 ra[4], rb[4];
    if constexpr (WIDE) {  // one run of a thousand records: chunks of 256 dealt to the four wavefronts (the CU's other workgroups cover the loads)
      if (static_cast<uint32_t>(wave) * 256u < wCnt) consume(wRec);  // (requested at the kernel's start)
      for (uint32_t c = static_cast<uint32_t>(wave) * 256u + T::kLanes / 64 * 256u; c < wCnt; c += T::kLanes / 64 * 256u) {
        if (__hip_atomic_load(&sBad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        const Chunk ch{wRun + c, wCnt - c < 256u ? wCnt - c : 256u};
        load(ra, ch);
        consume(ra);
      }
    } else if (m.streams > 0 && lpr >= 1u && lpr <= 4u) {
      const uint32_t units = static_cast<uint32_t>(m.streams) * 8u * lpr;
      for (uint32_t base = 0; base < units; base += 4u * T::kLanes) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t unit = base + static_cast<uint32_t>(k) * T::kLanes + tid;  // lane (unit & 7) of line (unit >> 3) % lpr of run (unit >> 3) / lpr
          const uint32_t ul = unit < units ? unit >> 3 : 0u, gg = ul / lpr, idx = (ul - gg * lpr) * 8u + (unit & 7u);
          const bool in = unit < units && idx < sRun[gg];
          ra[k] = *(in ? m.recB + (static_cast<uint64_t>(gg) * numParts + p) * m.capB + idx : pad);
          ra[k].x = in ? ra[k].x : kNoRow;
        }
        consume(ra);
        if (__hip_atomic_load(&sBad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
      }
    } else {
      // two register stages: the next chunk's loads are in flight while the current one goes through the table
      Chunk ca = next();
      load(ra, ca);
      while (ca.n) {
        Chunk cb = next();
        load(rb, cb);
        consume(ra);
        if (!cb.n || __hip_atomic_load(&sBad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        ca = next();
        load(ra, ca);
        consume(rb);
        if (__hip_atomic_load(&sBad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
      }
    }
    while (qn) {
      const uint32_t take = qn < 64u ? qn : 64u;
      qn -= take;
      drain(qn, take);
    }
  }
  __syncthreads();
  stamp(3);
  if (sBad) {  // (uniform)
    if (tid == 0) {
      m.flags[WIDE ? 3 : 1] = 1u;  // (wide: told apart from a record stream's overflow — more room for records would not help)
      m.partCount[p] = 0u;
    }
    return;
  }
  // ---- order: lane t owns slots [t * kPerLane, (t + 1) * kPerLane)
  const int first = tid * T::kPerLane;
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < T::kPerLane; k++) mine += first + k < T::kSlots && sKeys[first + k] != kEmptyKey;
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == 63) sWave[wave] = incl;
  __syncthreads();
  uint32_t before = incl - mine;
  for (int w = 0; w < wave; w++) before += sWave[w];
  // (wide: packed — a partition emits at most its entries: records + previous groups, whose prefixes are at hand)
  const uint64_t stageAt = WIDE ? static_cast<uint64_t>(m.offsetsB[p]) + (m.prevBounds ? m.prevBounds[p] : 0u) : static_cast<uint64_t>(p) * T::kStage;
  uint4 *stage = m.staging + stageAt;
  if constexpr (T::kPerLane == 8 && T::kSlots % 8 == 0) {
    // The lane's eight keys in registers (four 16-byte LDS reads), compared pair by pair where nothing but occupied slots lies
    // between them; a cluster that runs on to the left of the lane's first slot or to the right of its last is walked ONCE for
    // all of the lane's keys in it.  (Walked key by key — two chains of dependent LDS reads each — this phase was half of a
    // wide partition's life: 8.1 of 15.4 us.)
    uint64_t own[8];
    {
      const uint4 *q = reinterpret_cast<const uint4 *>(sKeys + first);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint4 t = q[k];
        own[2 * k] = (static_cast<uint64_t>(t.y) << 32) | t.x;
        own[2 * k + 1] = (static_cast<uint64_t>(t.w) << 32) | t.z;
      }
    }
    uint32_t occ = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) occ |= (own[k] != kEmptyKey ? 1u : 0u) << k;
    uint32_t smaller[8], start[8];  // keys of the cluster below own[k] (so far); first slot of its cluster within the lane
#pragma unroll
    for (int k = 0; k < 8; k++) {
      smaller[k] = 0;
      start[k] = 0;
    }
#pragma unroll
    for (int k = 1; k < 8; k++)
#pragma unroll
      for (int j = 0; j < k; j++) {
        const uint32_t span = ((1u << (k - j + 1)) - 1u) << j;  // slots j .. k
        const bool joined = (occ & span) == span;
        smaller[k] += joined && own[j] < own[k];
        smaller[j] += joined && own[k] < own[j];
      }
#pragma unroll
    for (int k = 1; k < 8; k++) start[k] = ((occ >> (k - 1)) & 1u) ? start[k - 1] : static_cast<uint32_t>(k);
    const uint32_t head = static_cast<uint32_t>(__builtin_ctz(~occ & 0x1FFu));         // own slots 0 .. head - 1 are one run from slot 0
    const uint32_t tail = static_cast<uint32_t>(__builtin_clz((~occ & 0xFFu) << 24 | 0x800000u));  // own slots 8 - tail .. 7 one run up to slot 7
    uint32_t left = 0;  // occupied slots right before the lane's first
    if (head) {
      for (int b = first - 1; b >= 0; b--) {
        const uint64_t kb = sKeys[b];
        if (kb == kEmptyKey) break;
#pragma unroll
        for (int k = 0; k < 8; k++) smaller[k] += static_cast<uint32_t>(k) < head && kb < own[k];
        left++;
      }
    }
    if (tail) {
      for (int f = first + 8; f < T::kSlots; f++) {
        const uint64_t kf = sKeys[f];
        if (kf == kEmptyKey) break;
#pragma unroll
        for (int k = 0; k < 8; k++) smaller[k] += static_cast<uint32_t>(k) >= 8u - tail && kf < own[k];
      }
    }
    uint32_t seen = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (!((occ >> k) & 1u)) continue;
      const int s = first + k;
      // occupied slots before the cluster's first = occupied slots before s - (slots of the cluster before s)
      const uint32_t inCluster = static_cast<uint32_t>(k) - start[k] + (start[k] == 0 ? left : 0u);
      const uint32_t rank = before + seen - inCluster + smaller[k];
      const uint64_t v = static_cast<uint64_t>(sVals[s]);
      stage[rank] = make_uint4(sRows[s], static_cast<uint32_t>(v), static_cast<uint32_t>(v >> 32), 0u);
      m.stageKeys[stageAt + rank] = own[k];
      seen++;
    }
  } else {
    uint32_t seen = 0;  // occupied slots of this lane before the current one
#pragma unroll
    for (int k = 0; k < T::kPerLane; k++) {
      const int s = first + k;
      if (s >= T::kSlots) break;
      const uint64_t key = sKeys[s];
      if (key == kEmptyKey) continue;
      uint32_t smaller = 0;
      int b = s - 1;
      while (b >= 0) {
        const uint64_t kb = sKeys[b];
        if (kb == kEmptyKey) break;
        smaller += kb < key;
        b--;
      }
      for (int f = s + 1; f < T::kSlots; f++) {
        const uint64_t kf = sKeys[f];
        if (kf == kEmptyKey) break;
        smaller += kf < key;
      }
      // occupied slots before the cluster's first = occupied slots before s - (s - first slot of the cluster)
      const uint32_t rank = before + seen - static_cast<uint32_t>(s - (b + 1)) + smaller;
      const uint64_t v = static_cast<uint64_t>(sVals[s]);
      stage[rank] = make_uint4(sRows[s], static_cast<uint32_t>(v), static_cast<uint32_t>(v >> 32), 0u);
      m.stageKeys[stageAt + rank] = key;
      seen++;
    }
  }
  if (tid == T::kLanes - 1) m.partCount[p] = before + mine;
  if (m.phases) __syncthreads();
  stamp(4);
}

// ---- 4. emission in partition order ---------------------------------------------------------------------------------
// dimension d of batch row r, as the transform would have stored it in the dimension vector (fused_eval_row of
// hr_kernels.hpp for columns of 1 / 2 / 4 bytes)
__device__ __forceinline__ void sr_eval_dim(const FusedPlanD &plan, int d, uint32_t r, uint32_t *bits, uint32_t *ok) {
  const FusedExpr &e = plan.dims[d];
  const FusedColumn col = plan.cols[e.col];
  const int step = col.step ? static_cast<int>(col.step) : 4;
  const uint8_t *base = reinterpret_cast<const uint8_t *>(col.vals);
  const bool sgn = e.f.akind == K_I32;
  uint32_t raw;
  if (step == 4) raw = col.vals[r];
  else if (step == 2) raw = sgn ? static_cast<uint32_t>(static_cast<int32_t>(reinterpret_cast<const int16_t *>(base)[r])) : reinterpret_cast<const uint16_t *>(base)[r];
  else raw = sgn ? static_cast<uint32_t>(static_cast<int32_t>(reinterpret_cast<const int8_t *>(base)[r])) : base[r];
  const uint32_t rok = col.nulls ? get_bit(col.nulls, r + col.bitOff) : 1u;
  const hr::FusedConst c = hr::fused_const(e.f);
  const DVal x = eval_fast(e.f, raw, rok, c.y, c.fd);
  *bits = cvt32(x, e.f.rk, e.outKind).bits;
  *ok = x.ok ? 1u : 0u;
}

template <int VW, bool WIDE>
__global__ __launch_bounds__((Table<VW, WIDE>::kLanes)) void sr_emit_kernel(SrArgs m, FusedPlanD plan, DimLayoutD L) {
  using T = Table<VW, WIDE>;
  __shared__ uint32_t sBase;
  const int p = blockIdx.x, tid = threadIdx.x;
  if constexpr (!WIDE) {
    if (tid == 0) sBase = 0;
    __syncthreads();
    if (tid < p) {
      const uint32_t c = m.partCount[tid];
      if (c) atomicAdd(&sBase, c);
    }
    __syncthreads();
  }
  const uint32_t count = m.partCount[p];
  const uint32_t base = WIDE ? m.partBase[p] : sBase;
  if (!WIDE && p == (1 << m.partBits) - 1 && tid == 0) m.flags[0] = base + count;  // (wide: sr_prefix_kernel)
  const uint64_t stageAt = WIDE ? static_cast<uint64_t>(m.offsetsB[p]) + (m.prevBounds ? m.prevBounds[p] : 0u) : static_cast<uint64_t>(p) * T::kStage;
  const uint4 *stage = m.staging + stageAt;
  const size_t cap = m.inCapacity;  // (the reference strides BOTH vectors by inputKeys.VectorCapacity: sort_reduce.cu:234-239)
  uint8_t *nullsOut = m.dimOut + static_cast<size_t>(L.valueBytes) * cap;
  for (uint32_t i = tid; i < count; i += T::kLanes) {
    const uint4 e = stage[i];
    const uint32_t at = base + i;
    m.keysOut[at] = m.stageKeys[stageAt + i];
    if (m.copyAll || e.x < m.prevSize) {
      copy_dim_row(m.dimIn, cap, m.dimOut, cap, L, e.x, at);
    } else {
      const uint32_t r = e.x - m.prevSize;
      for (int d = 0; d < L.numDims; d++) {
        uint32_t bits, ok;
        sr_eval_dim(plan, d, r, &bits, &ok);
        uint8_t *q = m.dimOut + static_cast<size_t>(L.valueOff[d]) * cap + static_cast<size_t>(L.width[d]) * at;
        if (L.width[d] == 4) *reinterpret_cast<uint32_t *>(q) = bits;
        else if (L.width[d] == 2) *reinterpret_cast<uint16_t *>(q) = static_cast<uint16_t>(bits);
        else *q = static_cast<uint8_t>(bits);
        nullsOut[static_cast<size_t>(d) * cap + at] = static_cast<uint8_t>(ok);
      }
    }
    if (VW == 8) reinterpret_cast<uint64_t *>(m.outValues)[at] = (static_cast<uint64_t>(e.z) << 32) | e.y;
    else reinterpret_cast<uint32_t *>(m.outValues)[at] = e.y;
  }
}

// ---- the wide layout's extra steps ------------------------------------------------------------------------------------------
// where each partition's range of the previous result's (ascending) row hashes begins: bounds[p] = first index whose key's
// top bits are >= p; bounds[numParts] = prevSize
__global__ __launch_bounds__(256) void sr_bounds_kernel(const uint64_t *prevKeys, uint32_t prevSize, int pb, uint32_t *bounds) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x, numParts = 1u << pb;
  if (p > numParts) return;
  uint32_t lo = 0, hi = prevSize;
  if (p == numParts || pb == 0) {
    lo = p == 0 ? 0u : prevSize;
  } else {
    const uint64_t bound = static_cast<uint64_t>(p) << (64 - pb);
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (prevKeys[mid] < bound) lo = mid + 1; else hi = mid;
    }
  }
  bounds[p] = lo;
}

// The scan's records lie in [workgroup][level-1 partition] streams (up to 512 partitions: what the scan's line staging
// holds); each level-1 partition is dealt out to its 2^(pb - pb1) partitions by the hash's top bits, into runs of EXACTLY
// the partition's size (a key that a million rows share fills one partition: no capacity to guess).  One workgroup per
// (level-1 partition, group of streams), twice: sr_count_kernel adds its records per partition (LDS histogram, one global
// atomic per partition it meets); after the prefix over the counts sr_split_kernel reserves its share of each run with one
// atomic per partition and writes the records there.  Order within a partition does not matter to the merge (lowest row by atomic
// min, integer aggregates).
struct SplitArgs {
  const uint4 *rec1;
  const uint32_t *counts1;
  uint32_t cap1;
  int streams, group, pb1, pb, spread;
  uint4 *rec2;
  uint32_t *counts2;         // records per partition (sr_count_kernel)
  const uint32_t *offsets2;  // their exclusive prefix
  uint32_t *cursors2;        // records placed so far
};
// the `sub`-th partition of level-1 partition p1.  spread: the scan filed a hash under (low pb1 bits of its partition index) XOR
// scramble(leading bits) — generate_vector, hr_rtc.hip —, `sub` = the leading bits
__device__ __forceinline__ uint32_t split_partition(uint32_t sub, uint32_t p1, const SplitArgs &s) {
  if (!s.spread) return (p1 << (s.pb - s.pb1)) + sub;  // (level 1 = the hash's top bits, `sub` the bits below)
  const uint32_t mask = (1u << s.pb1) - 1u;
  return (sub << s.pb1) | ((p1 ^ ((sub * 0x9E3779B1u) >> 23)) & mask);
}
__device__ __forceinline__ uint32_t split_sub(const uint4 &r, const SplitArgs &s) {
  const int sb = s.pb - s.pb1;
  return !sb ? 0u : s.spread ? r.y >> (32 - sb) : (r.y << s.pb1) >> (32 - sb);
}
__global__ __launch_bounds__(256) void sr_count_kernel(SplitArgs s) {
  __shared__ uint32_t sHist[256];
  const int tid = threadIdx.x;
  const int units = (s.streams + s.group - 1) / s.group;
  const int p1 = blockIdx.x / units, unit = blockIdx.x - p1 * units;
  const int numParts1 = 1 << s.pb1, sb = s.pb - s.pb1, fan = 1 << sb;
  const int g0 = unit * s.group, g1 = g0 + s.group < s.streams ? g0 + s.group : s.streams;
  auto sub_of = [&](const uint4 &r) { return split_sub(r, s); };
  sHist[tid] = 0;
  __syncthreads();
  for (int g = g0; g < g1; g++) {
    const uint32_t stored = s.counts1[static_cast<uint64_t>(g) * numParts1 + p1], cnt = stored < s.cap1 ? stored : s.cap1;
    const uint4 *run = s.rec1 + (static_cast<uint64_t>(g) * numParts1 + p1) * s.cap1;
    for (uint32_t i = tid; i < cnt; i += 256) {
      const uint4 r = run[i];
      if (r.x != kNoRow) __hip_atomic_fetch_add(sHist + sub_of(r), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  if (tid < fan && sHist[tid]) atomicAdd(s.counts2 + split_partition(static_cast<uint32_t>(tid), static_cast<uint32_t>(p1), s), sHist[tid]);
}

// ... the second launch: tiles of 2048 records (the unit's runs taken as one sequence) are ordered by partition in LDS —
// histogram, prefix, one reservation per partition met — and written out in that order: a partition's share of the tile is
// one contiguous piece (written record by record in arrival order, 16 bytes here and 16 there, the same data took twice as long)
constexpr int kSplitTile = 2048, kSplitPerLane = kSplitTile / 256;
__global__ __launch_bounds__(256) void sr_split_kernel(SplitArgs s) {
  __shared__ uint4 sTile[kSplitTile];
  __shared__ uint32_t sHist[256], sStart[256], sBase[256], sRunStart[257], sWaveSum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int units = (s.streams + s.group - 1) / s.group;
  const int p1 = blockIdx.x / units, unit = blockIdx.x - p1 * units;
  const int numParts1 = 1 << s.pb1, sb = s.pb - s.pb1;
  const int g0 = unit * s.group, g1 = g0 + s.group < s.streams ? g0 + s.group : s.streams, runs = g1 - g0;
  auto sub_of = [&](const uint4 &r) { return split_sub(r, s); };
  // the unit's runs as one sequence: sRunStart[k] = records before run k (runs <= 256: one per lane, scanned by the workgroup)
  {
    uint32_t c = 0;
    if (tid < runs) {
      const uint32_t stored = s.counts1[static_cast<uint64_t>(g0 + tid) * numParts1 + p1];
      c = stored < s.cap1 ? stored : s.cap1;
    }
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) sWaveSum[wave] = incl;
    __syncthreads();
    uint32_t before = incl - c;
    for (int w = 0; w < wave; w++) before += sWaveSum[w];
    sRunStart[tid] = before;
    if (tid == 255) sRunStart[256] = before + c;
    __syncthreads();
  }
  const uint32_t total = sRunStart[runs < 256 ? runs : 256];
  for (uint32_t tile = 0; tile < total; tile += kSplitTile) {
    sHist[tid] = 0;
    __syncthreads();
    uint4 r[kSplitPerLane];
    uint32_t rank[kSplitPerLane];
#pragma unroll
    for (int k = 0; k < kSplitPerLane; k++) {
      const uint32_t j = tile + static_cast<uint32_t>(k) * 256u + tid;
      r[k].x = kNoRow;
      if (j < total) {
        int lo = 0, hi = runs - 1;  // the run that holds record j: last k with sRunStart[k] <= j
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (sRunStart[mid] <= j) lo = mid; else hi = mid - 1;
        }
        r[k] = s.rec1[(static_cast<uint64_t>(g0 + lo) * numParts1 + p1) * s.cap1 + (j - sRunStart[lo])];
      }
    }
#pragma unroll
    for (int k = 0; k < kSplitPerLane; k++)
      rank[k] = r[k].x != kNoRow ? __hip_atomic_fetch_add(sHist + sub_of(r[k]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
    __syncthreads();
    {  // where each partition's piece starts in the tile, and in the partition's run
      const uint32_t c = sHist[tid];
      uint32_t incl = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
      }
      if (lane == 63) sWaveSum[wave] = incl;
      __syncthreads();
      uint32_t before = incl - c;
      for (int w = 0; w < wave; w++) before += sWaveSum[w];
      sStart[tid] = before;
      const uint32_t part = split_partition(static_cast<uint32_t>(tid), static_cast<uint32_t>(p1), s);
      sBase[tid] = c ? s.offsets2[part] + atomicAdd(s.cursors2 + part, c) : 0u;
    }
    __syncthreads();
    uint32_t kept = 0;
#pragma unroll
    for (int k = 0; k < kSplitPerLane; k++)
      if (r[k].x != kNoRow) sTile[sStart[sub_of(r[k])] + rank[k]] = r[k];
    kept = sStart[255] + sHist[255];
    __syncthreads();
    for (uint32_t j = tid; j < kept; j += 256) {
      const uint4 q = sTile[j];
      const uint32_t sub = sub_of(q);
      s.rec2[sBase[sub] + (j - sStart[sub])] = q;
    }
    __syncthreads();
  }
}

// exclusive prefix of a per-partition count (one workgroup of 1024 lanes walks tiles of 4096 counts; numParts <= 2^17):
// records / groups before each partition, the total to *total
__global__ __launch_bounds__(1024) void sr_prefix_kernel(const uint32_t *partCount, int numParts, uint32_t *partBase, uint32_t *total) {
  __shared__ uint32_t sWave[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t carry = 0;
  int flip = 0;
  for (int base = 0; base < numParts; base += 4096, flip ^= 1) {
    const int i = base + 4 * tid;
    uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = i + k < numParts ? partCount[i + k] : 0u;
    const uint32_t mine = c[0] + c[1] + c[2] + c[3];
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) sWave[flip][wave] = incl;
    __syncthreads();
    uint32_t before = carry + incl - mine, all = 0;
    for (int w = 0; w < 16; w++) {
      const uint32_t t = sWave[flip][w];
      before += w < wave ? t : 0u;
      all += t;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (i + k < numParts) partBase[i + k] = before;
      before += c[k];
    }
    carry += all;
  }
  if (tid == 0) *total = carry;
}

// ---- the row hashes of a result, kept beside it ---------------------------------------------------------------------------
// A fused Sort + Reduce leaves, in a block of its own, the 64-bit row hash of every group it emitted — ascending, like the
// groups.  When the host feeds those vectors back as the first rows of the query's next Reduce (query/aql_processor.go:718-724),
// the merge reads each partition's previous groups as a RANGE of that array: sr_prev_kernel (hashing 2.3 M rows and scattering
// them into region A: 0.085 ms per batch) is not run.  The array is trusted only while nothing has written to the vectors
// (every writer of a result vector reports to grouped_note_write, which calls sorted_state_note_write) — the contract the
// partition-grouped ranges and table images of HashReduce live by.  ARES_SORT_STATE=0: off.
struct SortedState {
  int device;
  const uint8_t *dims;
  const uint8_t *values;
  size_t capacity;
  uint8_t ndw[NUM_DIM_WIDTH];
  int valueBytes, size;
  std::shared_ptr<uint64_t> keys;  // (a caller that looked the state up keeps the block alive while its kernels read it)
};
std::mutex g_sortedMutex;
std::vector<SortedState> g_sorted;

bool sorted_state_enabled() {
  static EnvSwitch<bool> on("ARES_SORT_STATE", [](const char *e) { return !(e && e[0] == '0'); });
  return on.get() && deferral_hooks_active();
}

// a block of the temporaries' cache that outlives the call: released as idle memory (its last user has been waited for)
std::shared_ptr<uint64_t> take_key_block(int device, size_t groups, hipStream_t stream) {
  void *p = stream_alloc(sizeof(uint64_t) * (groups ? groups : 1), stream);
  return std::shared_ptr<uint64_t>(static_cast<uint64_t *>(p), [device](uint64_t *q) {
    int current = 0;
    const bool have = hipGetDevice(&current) == hipSuccess;
    if (have && current != device) (void)hipSetDevice(device);
    stream_release_idle(q);
    if (have && current != device) (void)hipSetDevice(current);
  });
}

size_t dim_row_bytes(const uint8_t ndw[NUM_DIM_WIDTH]) {
  size_t rowBytes = 0;
  for (int w = 0; w < NUM_DIM_WIDTH; w++) rowBytes += static_cast<size_t>(ndw[w]) * ((1u << (NUM_DIM_WIDTH - 1 - w)) + 1);
  return rowBytes;
}

std::shared_ptr<uint64_t> sorted_state_lookup(int device, const DimensionVector &v, const uint8_t *values, int valueBytes, int size) {
  if (!sorted_state_enabled() || size <= 0) return nullptr;
  std::lock_guard<std::mutex> lock(g_sortedMutex);
  for (const SortedState &st : g_sorted)
    if (st.device == device && st.dims == v.DimValues && st.values == values && st.capacity == static_cast<size_t>(v.VectorCapacity) &&
        memcmp(st.ndw, v.NumDimsPerDimWidth, sizeof(st.ndw)) == 0 && st.valueBytes == valueBytes && st.size == size)
      return st.keys;
  return nullptr;
}

// ... whatever its size (a Reduce over materialised vectors is not told where the previous result ends): the state whose rows
// are the first *size <= maxSize rows of these vectors
std::shared_ptr<uint64_t> sorted_state_find(int device, const DimensionVector &v, const uint8_t *values, int valueBytes, int maxSize, int *size) {
  if (!sorted_state_enabled() || maxSize <= 0) return nullptr;
  std::lock_guard<std::mutex> lock(g_sortedMutex);
  for (const SortedState &st : g_sorted)
    if (st.device == device && st.dims == v.DimValues && st.values == values && st.capacity == static_cast<size_t>(v.VectorCapacity) &&
        memcmp(st.ndw, v.NumDimsPerDimWidth, sizeof(st.ndw)) == 0 && st.valueBytes == valueBytes && st.size <= maxSize) {
      *size = st.size;
      return st.keys;
    }
  return nullptr;
}

void sorted_state_register(int device, const DimensionVector &v, const uint8_t *values, size_t capacity, int valueBytes, int size,
                           std::shared_ptr<uint64_t> keys) {
  if (!sorted_state_enabled() || size <= 0) return;
  SortedState st{device, v.DimValues, values, capacity, {}, valueBytes, size, std::move(keys)};
  memcpy(st.ndw, v.NumDimsPerDimWidth, sizeof(st.ndw));
  std::lock_guard<std::mutex> lock(g_sortedMutex);
  if (g_sorted.size() >= 64) g_sorted.erase(g_sorted.begin());  // (a host that never frees: forget the oldest)
  g_sorted.push_back(std::move(st));
}

int sr_part_bits(int64_t length) {
  // ARES_MIN_PART_BITS (tests): small inputs take several partitions like production-sized ones
  static EnvSwitch<int> minBits("ARES_MIN_PART_BITS", [](const char *e) { return e ? atoi(e) : 0; });
  int partBits = minBits.get() > 0 ? (minBits.get() < 9 ? minBits.get() : 9) : 0;
  while ((4096ll << partBits) < length && (1 << partBits) < kMaxPartitions) partBits++;
  return partBits;
}

}  // namespace

// something writes (or frees) [ptr, ptr + bytes): what is known about result vectors in there is void (hash_reduce_lds.hip:
// grouped_note_write passes every report on)
void sorted_state_note_write(int device, const void *ptr, size_t bytes) {
  const uint8_t *lo = static_cast<const uint8_t *>(ptr), *hi = lo + (bytes ? bytes : 1);
  std::lock_guard<std::mutex> lock(g_sortedMutex);
  for (size_t i = 0; i < g_sorted.size();) {
    const SortedState &st = g_sorted[i];
    // the rows the state describes: [0, size) of every dimension's values and validity bytes, and of the measure vector (the
    // next batch's transforms write rows BEHIND them into the same vectors)
    auto hit = [&](const uint8_t *a, size_t n) { return a < hi && lo < a + n; };
    bool touched = st.device == device && hit(st.values, static_cast<size_t>(st.valueBytes) * st.size);
    if (st.device == device && !touched) {
      size_t off = 0, valueBytes = 0;
      int nd = 0;
      for (int w = 0; w < NUM_DIM_WIDTH; w++) valueBytes += static_cast<size_t>(st.ndw[w]) << (NUM_DIM_WIDTH - 1 - w);
      for (int w = 0; w < NUM_DIM_WIDTH && !touched; w++) {
        const size_t width = static_cast<size_t>(1) << (NUM_DIM_WIDTH - 1 - w);
        for (int k = 0; k < st.ndw[w] && !touched; k++) {
          touched = hit(st.dims + off * st.capacity, width * st.size) || hit(st.dims + valueBytes * st.capacity + static_cast<size_t>(nd) * st.capacity, st.size);
          off += width;
          nd++;
        }
      }
    }
    if (touched) g_sorted.erase(g_sorted.begin() + i);
    else i++;
  }
}

bool fused_sort_reduce_enabled() {
  static EnvSwitch<bool> on("ARES_SORT_FUSE", [](const char *e) { return !(e && e[0] == '0'); });
  return on.get() && rtc_scan_available();
}

bool fused_sort_reduce_supported(const AggSpec &a) {
  if (a.vtype == V_U32 || a.vtype == V_I32) return a.op == OP_SUM || a.op == OP_MIN || a.op == OP_MAX;
  return (a.vtype == V_U64 || a.vtype == V_I64) && a.op == OP_SUM;
}

int fused_sort_reduce_run(int device, const FusedPlanD &plan, int nd, bool constMeasure, uint64_t constBits, int batchRows,
                          const DimensionVector &in, const uint8_t *inValues, int prevSize, const DimensionVector &out,
                          uint8_t *outValues, const AggSpec &a, hipStream_t stream) {
  if (!fused_sort_reduce_enabled() || !fused_sort_reduce_supported(a) || batchRows <= 0 || prevSize < 0) return kFusedUnavailable;
  // ARES_SR_SCAN_FED=0 (tests): this path declines everything — its callers go on to the wide layout over materialised rows
  static EnvSwitch<bool> scanFed("ARES_SR_SCAN_FED", [](const char *e) { return !(e && e[0] == '0'); });
  if (!scanFed.get()) return kFusedUnavailable;
  const int vw = a.width;
  static EnvSwitch<int> maxGroups("ARES_SR_MAX_GROUPS", [](const char *e) { return e ? atoi(e) : 0; });
  int tableGroups = vw == 4 ? Table<4>::kMaxGroups : Table<8>::kMaxGroups;
  if (maxGroups.get() > 0 && maxGroups.get() < tableGroups) tableGroups = maxGroups.get();
  const int64_t length = static_cast<int64_t>(batchRows) + prevSize;
  const int partBits = sr_part_bits(length);
  const int numParts = 1 << partBits;
  // the previous result alone must fit the tables with room for a partition's share to vary (hashes spread evenly: the
  // largest of 512 partitions of a 2.5 M-group result holds ~5 % more than the mean): beyond, the real sort
  if (static_cast<int64_t>(prevSize) > static_cast<int64_t>(numParts) * tableGroups * 9 / 10) return kFusedUnavailable;
  const DimLayoutD L = make_dim_layout(in.NumDimsPerDimWidth);
  if (L.numDims != nd) return kFusedUnavailable;
  RtcKernel scan = rtc_sort_scan_lookup(device, plan, nd, partBits);
  if (!scan) return kFusedUnavailable;  // being compiled in the background (or a shape the generator declines)

  // ---- workspace: [cursors A | flags][counts B][partition counts][region A][region B][staging]
  const int streams = rtc_scan_grid(batchRows);
  const uint64_t mean = (static_cast<uint64_t>(batchRows) / (static_cast<uint64_t>(numParts) * streams)) << record_stream_slack();
  const uint32_t capB = static_cast<uint32_t>(((2 * mean + 64 + 7) / 8 * 8) | 8ull);  // whole lines of 8; odd line count per stream
  const uint64_t capA = ((2ull * (static_cast<uint64_t>(prevSize) / numParts) + 1024) | 63ull) + 18;
  const size_t stageSlots = static_cast<size_t>(vw == 4 ? Table<4>::kSlots : Table<8>::kSlots);
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t headBytes = up(sizeof(uint32_t) * (numParts + 16));
  const size_t countsBytes = up(sizeof(uint32_t) * static_cast<size_t>(numParts) * streams);
  const size_t partBytes = up(sizeof(uint32_t) * numParts);
  const size_t aBytes = up(sizeof(uint4) * capA * numParts);
  const size_t bBytes = up(sizeof(uint4) * static_cast<size_t>(capB) * numParts * streams);
  const size_t stageBytes = up(sizeof(uint4) * stageSlots * numParts), stageKeyBytes = up(sizeof(uint64_t) * stageSlots * numParts);
  StreamBuffer buf(headBytes + countsBytes + partBytes + aBytes + bBytes + stageBytes + stageKeyBytes + 256, stream);
  uint8_t *base = buf.as<uint8_t>();
  hip_check(hipMemsetAsync(base, 0, headBytes, stream), "hipMemsetAsync");
  hr::Workspace ws;
  memset(&ws, 0, sizeof(ws));
  ws.cursorsA = reinterpret_cast<uint32_t *>(base);
  ws.outCount = ws.cursorsA + numParts;  // [0] groups, [1] overflow, [2] empty-key hash
  ws.countsB = reinterpret_cast<uint32_t *>(base + headBytes);
  uint32_t *partCount = reinterpret_cast<uint32_t *>(base + headBytes + countsBytes);
  ws.recA = reinterpret_cast<uint4 *>(base + headBytes + countsBytes + partBytes);
  ws.capA = capA;
  ws.recB = reinterpret_cast<uint32_t *>(base + headBytes + countsBytes + partBytes + aBytes);
  ws.capB = capB;
  ws.streams = streams;
  ws.partBits = partBits;
  ws.lineRecords = 8;
  ws.rowBase = static_cast<uint32_t>(prevSize);

  SrArgs m;
  memset(&m, 0, sizeof(m));
  m.recA = ws.recA;
  m.cursorsA = ws.cursorsA;
  m.capA = capA;
  m.recB = reinterpret_cast<const uint4 *>(ws.recB);
  m.countsB = ws.countsB;
  m.capB = capB;
  m.streams = streams;
  m.partBits = partBits;
  m.dimIn = in.DimValues;
  m.inValues = inValues;
  m.inCapacity = static_cast<size_t>(in.VectorCapacity);
  m.prevSize = static_cast<uint32_t>(prevSize);
  m.widen.mode = vw == 8 ? 1 : 0;
  m.widen.rk = plan.measure.f.rk;
  m.widen.dtype = plan.measureDtype;
  m.constMeasure = constMeasure ? 1 : 0;
  m.constBits = constBits;
  m.agg = a;
  m.staging = reinterpret_cast<uint4 *>(base + headBytes + countsBytes + partBytes + aBytes + bBytes);
  m.stageKeys = reinterpret_cast<uint64_t *>(base + headBytes + countsBytes + partBytes + aBytes + bBytes + stageBytes);
  m.partCount = partCount;
  m.flags = ws.outCount;
  m.maxGroups = static_cast<uint32_t>(tableGroups);
  m.dimOut = out.DimValues;
  m.outValues = outValues;

  // the previous result's row hashes, if the query's previous Reduce left them (and nothing has written to the vectors since)
  const std::shared_ptr<uint64_t> prevKeys = prevSize > 0 ? sorted_state_lookup(device, in, inValues, vw, prevSize) : nullptr;
  const std::shared_ptr<uint64_t> keysOut = take_key_block(device, static_cast<size_t>(length), stream);
  m.prevKeys = prevKeys.get();
  m.keysOut = keysOut.get();
  if (prevSize > 0 && !prevKeys) {
    const int grid = static_cast<int>(std::min<int64_t>((static_cast<int64_t>(prevSize) + 255) / 256, 256 * 8));
    ARES_LAUNCH("sr_prev_kernel", sr_prev_kernel, grid, 256, stream, m, L, ws.recA);
  }
  rtc_sort_scan_launch(scan, plan, static_cast<uint32_t>(prevSize), batchRows, ws, stream);
  static const bool phasesOn = [] {
    const char *e = getenv("ARES_HR_PHASES");
    return e && e[0] == '1';
  }();
  static uint64_t *phases = nullptr;
  if (phasesOn) {
    if (!phases) hip_check(hipMalloc(reinterpret_cast<void **>(&phases), sizeof(uint64_t) * 8 * kMaxPartitions), "hipMalloc");
    hip_check(hipMemsetAsync(phases, 0, sizeof(uint64_t) * 8 * kMaxPartitions, stream), "hipMemsetAsync");
    m.phases = phases;
  }
  if (vw == 8) {
    ARES_LAUNCH("sr_merge_kernel", (sr_merge_kernel<8, false>), numParts, kThreads, stream, m);
    ARES_LAUNCH("sr_emit_kernel", (sr_emit_kernel<8, false>), numParts, kThreads, stream, m, plan, L);
  } else {
    ARES_LAUNCH("sr_merge_kernel", (sr_merge_kernel<4, false>), numParts, kThreads, stream, m);
    ARES_LAUNCH("sr_emit_kernel", (sr_emit_kernel<4, false>), numParts, kThreads, stream, m, plan, L);
  }
  uint32_t w[3] = {0, 0, 0};
  read_back_u32(ws.outCount, w, 3, stream);
  if (phasesOn) {  // diagnostics: where a partition's time goes
    static int launches = 0;
    std::vector<uint64_t> h(static_cast<size_t>(8) * numParts);
    hip_check(hipMemcpy(h.data(), phases, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost), "hipMemcpy");
    if (++launches <= 4 || launches % 16 == 0) {
      double sum[4] = {0, 0, 0, 0};
      uint64_t first = ~0ull, last = 0;
      for (int p = 0; p < numParts; p++) {
        const uint64_t *t = &h[static_cast<size_t>(8) * p];
        if (t[0] < first) first = t[0];
        if (t[4] > last) last = t[4];
        for (int k = 0; k < 4; k++) sum[k] += static_cast<double>(t[k + 1] - t[k]) * 0.01;
      }
      fprintf(stderr, "sr_merge_kernel phases (launch %d, %d partitions, prev %d, batch %d): span %.1f us; per partition avg: init %.1f + previous groups %.1f + records %.1f + order %.1f us\n",
              launches, numParts, prevSize, batchRows, static_cast<double>(last - first) * 0.01, sum[0] / numParts, sum[1] / numParts, sum[2] / numParts, sum[3] / numParts);
    }
  }
  buf.mark_idle();
  static const bool trace = getenv("ARES_HR_TRACE") != nullptr;  // diagnostics
  if (trace)
    fprintf(stderr, "fused_sort_reduce_run: batch %d prev %d partBits %d streams %d capA %llu capB %u vw %d const %d -> groups %u overflow %u emptykey %u\n",
            batchRows, prevSize, partBits, streams, static_cast<unsigned long long>(capA), capB, vw, constMeasure ? 1 : 0, w[0], w[1], w[2]);
  // a record stream may have overflowed (sorted rows fill a chunk's partitions unevenly): more room — kept for the process — and again
  if (w[1] && !w[2] && grow_record_stream_slack())
    return fused_sort_reduce_run(device, plan, nd, constMeasure, constBits, batchRows, in, inValues, prevSize, out, outValues, a, stream);
  if (w[1] || w[2]) return -1;  // the outputs may be partly written: the caller runs the real Sort + Reduce over them
  // (the caller reported the output vectors as rewritten before this call: what is registered now describes the new rows)
  sorted_state_register(device, out, outValues, static_cast<size_t>(in.VectorCapacity), vw, static_cast<int>(w[0]), keysOut);
  return static_cast<int>(w[0]);
}

// Sort + Reduce over MATERIALISED vectors (rows [0, length) of `in` and `inValues` exist: the batch's dimensions came from a
// join, a generic expression — anything the fused scans do not evaluate): the same aggregation by 64-bit row hash, in the
// wide layout — the previous result is tens of millions of groups when this matters (50 M keys, 64 Mi rows per batch: four
// radix passes over 114 M entries + the hashing pass + a reduce that gathers every entry's dimension row, 10 ms):
//   sr_bounds_kernel   (previous result's row hashes known) each partition's range of them;
//   sr_scan_rtc        (hr_rtc.hip: generate_vector, sort64) hashes rows [prev, length) — all rows when the previous result's
//                      hashes are not known — into <= 512 level-1 partitions x streams;
//   sr_split_kernel    deals each level-1 partition's records out to its partitions (2^pb <= 2^17 in all, ~1 k entries each);
//   sr_merge_kernel    <WIDE>: 2048-slot tables, one run per partition;
//   sr_prefix_kernel   groups before each partition;
//   sr_emit_kernel     <WIDE>: the representatives' dimension rows copied from `in` (ascending: the previous result's rows come
//                      out in the order they lie in).
// 4-byte dimensions (up to eight) and 4-byte integer aggregates; returns like fused_sort_reduce_run.
static int sort_reduce_vectors_run(int device, int length, const DimensionVector &in, const uint8_t *inValues, const DimensionVector &out,
                                   uint8_t *outValues, const AggSpec &a, hipStream_t stream, int slack, bool spread) {
  static const bool trace = getenv("ARES_HR_TRACE") != nullptr;  // diagnostics
  auto decline = [&](const char *why) {
    if (trace) fprintf(stderr, "fused_sort_reduce_vectors: rows %d declined: %s\n", length, why);
    return kFusedUnavailable;
  };
  if (!fused_sort_reduce_enabled() || !fused_sort_reduce_supported(a) || (a.width != 4 && a.width != 8) || length <= 0) return decline("aggregate");
  const DimLayoutD L = make_dim_layout(in.NumDimsPerDimWidth);
  const int nd = L.numDims;
  if (nd < 1 || nd > kFusedDims || in.NumDimsPerDimWidth[0] || in.NumDimsPerDimWidth[1]) return decline("layout");  // (slots of 4 / 2 / 1 bytes)
  int widths[kFusedDims];
  for (int d = 0; d < nd; d++) widths[d] = L.width[d];
  using T = Table<4, true>;  // (the 8-byte tables have the same number of slots)
  static_assert(Table<8, true>::kSlots == T::kSlots && Table<8, true>::kMaxGroups == T::kMaxGroups, "wide tables differ by value width");
  const int vw = a.width;
  static EnvSwitch<int> maxGroups("ARES_SR_MAX_GROUPS", [](const char *e) { return e ? atoi(e) : 0; });
  int tableGroups = T::kMaxGroups;
  if (maxGroups.get() > 0 && maxGroups.get() < tableGroups) tableGroups = maxGroups.get();
  // ARES_SRV_PART_BITS (tests): small inputs take the partition counts of production-sized ones
  static EnvSwitch<int> forceBits("ARES_SRV_PART_BITS", [](const char *e) { return e ? atoi(e) : -1; });
  int partBits = 0;
  while ((static_cast<int64_t>(kWideEntries) << partBits) < length && partBits < kWideMaxPartBits) partBits++;
  if (forceBits.get() >= 0) partBits = forceBits.get() < kWideMaxPartBits ? forceBits.get() : kWideMaxPartBits;
  const int numParts = 1 << partBits;
  if (static_cast<int64_t>(length) > static_cast<int64_t>(numParts) * tableGroups * 7 / 10) return decline("more rows than the tables take");  // (all rows may be groups)
  const int pb1 = partBits < 9 ? partBits : 9, numParts1 = 1 << pb1;
  RtcKernel scan = rtc_sort_vector_scan_lookup(device, nd, widths, pb1);
  if (!scan) return decline("scan kernel not available (yet)");  // being compiled in the background

  int prevSize = 0;
  const std::shared_ptr<uint64_t> prevKeys = sorted_state_find(device, in, inValues, vw, length, &prevSize);
  if (!prevKeys) prevSize = 0;  // every row is hashed
  const int batchRows = length - prevSize;

  // ---- workspace: [head: flags][level-1 counts][cursors][partition counts][partition bases][bounds][level 1][level 2][staging][keys]
  const int streams = batchRows > 0 ? rtc_scan_grid(batchRows) : 1;
  // (`slack`: more room per level-1 stream after an overflow — for this call only: what overflows here is a previous result
  // whose row hashes are not known, hashed again in the ascending order it lies in)
  const uint64_t mean1 = (static_cast<uint64_t>(batchRows) / (static_cast<uint64_t>(numParts1) * streams)) << slack;
  const uint32_t cap1 = static_cast<uint32_t>(((2 * mean1 + 64 + 7) / 8 * 8) | 8ull);
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t headBytes = up(sizeof(uint32_t) * 16);
  const size_t counts1Bytes = up(sizeof(uint32_t) * static_cast<size_t>(numParts1) * streams);
  const size_t cursorBytes = up(sizeof(uint32_t) * numParts);  // (twice: counts, cursors)
  const size_t partBytes = up(sizeof(uint32_t) * (numParts + 1));
  const size_t l1Bytes = up(sizeof(uint4) * static_cast<size_t>(cap1) * numParts1 * streams);
  const size_t l2Bytes = up(sizeof(uint4) * (static_cast<size_t>(batchRows) + 16));
  // staging (packed: one entry per row at most) shares the level-1 region: the split has read it before the merge stages anything
  const size_t stageBytes = up(sizeof(uint4) * (static_cast<size_t>(length) + 16)), stageKeyBytes = up(sizeof(uint64_t) * (static_cast<size_t>(length) + 16));
  const size_t sharedBytes = l1Bytes > stageBytes + stageKeyBytes ? l1Bytes : stageBytes + stageKeyBytes;
  StreamBuffer buf(headBytes + counts1Bytes + 2 * cursorBytes + 4 * partBytes + sharedBytes + l2Bytes + 256, stream);
  uint8_t *at = buf.as<uint8_t>();
  auto take = [&](size_t bytes) {
    uint8_t *p = at;
    at += bytes;
    return p;
  };
  uint32_t *flags = reinterpret_cast<uint32_t *>(take(headBytes));
  uint32_t *counts1 = reinterpret_cast<uint32_t *>(take(counts1Bytes));
  uint32_t *counts2 = reinterpret_cast<uint32_t *>(take(cursorBytes));
  uint32_t *cursors2 = reinterpret_cast<uint32_t *>(take(cursorBytes));
  hip_check(hipMemsetAsync(flags, 0, headBytes + counts1Bytes + 2 * cursorBytes, stream), "hipMemsetAsync");
  uint32_t *offsets2 = reinterpret_cast<uint32_t *>(take(partBytes));
  uint32_t *partCount = reinterpret_cast<uint32_t *>(take(partBytes));
  uint32_t *partBase = reinterpret_cast<uint32_t *>(take(partBytes));
  uint32_t *bounds = reinterpret_cast<uint32_t *>(take(partBytes));
  uint8_t *shared = take(sharedBytes);
  uint4 *rec1 = reinterpret_cast<uint4 *>(shared);
  uint4 *rec2 = reinterpret_cast<uint4 *>(take(l2Bytes));

  SrArgs m;
  memset(&m, 0, sizeof(m));
  m.recB = rec2;
  m.countsB = counts2;
  m.offsetsB = offsets2;
  m.capB = 0;
  m.streams = 1;
  m.partBits = partBits;
  m.dimIn = in.DimValues;
  m.inValues = inValues;
  m.inCapacity = static_cast<size_t>(in.VectorCapacity);
  m.prevSize = static_cast<uint32_t>(prevSize);
  m.widen.mode = 0;
  m.widen.rk = a.vtype == V_I32 ? K_I32 : K_U32;
  m.widen.dtype = a.vtype == V_I32 ? Int32 : Uint32;
  m.gatherValues = vw == 8 ? 1 : 0;
  m.agg = a;
  m.staging = reinterpret_cast<uint4 *>(shared);
  m.stageKeys = reinterpret_cast<uint64_t *>(shared + stageBytes);
  m.partCount = partCount;
  m.flags = flags;
  m.maxGroups = static_cast<uint32_t>(tableGroups);
  m.dimOut = out.DimValues;
  m.outValues = outValues;
  m.prevKeys = prevSize > 0 ? prevKeys.get() : nullptr;
  m.prevBounds = prevSize > 0 ? bounds : nullptr;
  m.partBase = partBase;
  m.copyAll = 1;
  const std::shared_ptr<uint64_t> keysOut = take_key_block(device, static_cast<size_t>(length), stream);
  m.keysOut = keysOut.get();

  if (prevSize > 0)
    ARES_LAUNCH("sr_bounds_kernel", sr_bounds_kernel, (numParts + 1 + 255) / 256, 256, stream, prevKeys.get(), static_cast<uint32_t>(prevSize), partBits, bounds);
  if (batchRows > 0) {
    hr::Workspace ws;
    memset(&ws, 0, sizeof(ws));
    ws.outCount = flags;  // [0] groups, [1] overflow, [2] empty-key hash
    ws.countsB = counts1;
    ws.recB = reinterpret_cast<uint32_t *>(rec1);
    ws.capB = cap1;
    ws.streams = streams;
    ws.partBits = pb1;
    ws.lineRecords = 8;
    ws.rowBase = static_cast<uint32_t>(prevSize);
    rtc_sort_vector_scan_launch(scan, in.DimValues, static_cast<size_t>(in.VectorCapacity), inValues, nd, widths, static_cast<uint32_t>(prevSize), batchRows, partBits, spread, ws, stream);
    SplitArgs sp;
    memset(&sp, 0, sizeof(sp));
    sp.rec1 = rec1;
    sp.counts1 = counts1;
    sp.cap1 = cap1;
    sp.streams = streams;
    // ~4096 records per workgroup
    const uint64_t run = static_cast<uint64_t>(batchRows) / (static_cast<uint64_t>(numParts1) * streams) + 1;
    sp.group = static_cast<int>(std::max<uint64_t>(1, std::min<uint64_t>(static_cast<uint64_t>(streams), 4096 / run)));
    sp.pb1 = pb1;
    sp.pb = partBits;
    sp.spread = spread ? 1 : 0;
    sp.rec2 = rec2;
    sp.counts2 = counts2;
    sp.offsets2 = offsets2;
    sp.cursors2 = cursors2;
    const int units = (streams + sp.group - 1) / sp.group;
    ARES_LAUNCH("sr_count_kernel", sr_count_kernel, numParts1 * units, 256, stream, sp);
    ARES_LAUNCH("sr_prefix_kernel", sr_prefix_kernel, 1, 1024, stream, counts2, numParts, offsets2, flags + 4);
    ARES_LAUNCH("sr_split_kernel", sr_split_kernel, numParts1 * units, 256, stream, sp);
  } else {
    hip_check(hipMemsetAsync(offsets2, 0, partBytes, stream), "hipMemsetAsync");
  }
  static const bool phasesOn = [] {
    const char *e = getenv("ARES_HR_PHASES");
    return e && e[0] == '1';
  }();
  static uint64_t *phases = nullptr;
  if (phasesOn) {
    if (!phases) hip_check(hipMalloc(reinterpret_cast<void **>(&phases), sizeof(uint64_t) * 8 << kWideMaxPartBits), "hipMalloc");
    hip_check(hipMemsetAsync(phases, 0, sizeof(uint64_t) * 8 * numParts, stream), "hipMemsetAsync");
    m.phases = phases;
  }
  FusedPlanD noPlan;
  memset(&noPlan, 0, sizeof(noPlan));
  if (vw == 8) ARES_LAUNCH("sr_merge_kernel", (sr_merge_kernel<8, true>), numParts, T::kLanes, stream, m);
  else ARES_LAUNCH("sr_merge_kernel", (sr_merge_kernel<4, true>), numParts, T::kLanes, stream, m);
  ARES_LAUNCH("sr_prefix_kernel", sr_prefix_kernel, 1, 1024, stream, partCount, numParts, partBase, flags);  // (flags[0]: groups)
  if (vw == 8) ARES_LAUNCH("sr_emit_kernel", (sr_emit_kernel<8, true>), numParts, T::kLanes, stream, m, noPlan, L);
  else ARES_LAUNCH("sr_emit_kernel", (sr_emit_kernel<4, true>), numParts, T::kLanes, stream, m, noPlan, L);
  uint32_t w[4] = {0, 0, 0, 0};
  read_back_u32(flags, w, 4, stream);
  if (phasesOn) {  // diagnostics: where a partition's time goes
    static int launches = 0;
    std::vector<uint64_t> h(static_cast<size_t>(8) * numParts);
    hip_check(hipMemcpy(h.data(), phases, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost), "hipMemcpy");
    if (++launches <= 6 || launches % 16 == 0) {
      double sum[4] = {0, 0, 0, 0}, resident = 0;
      uint64_t first = ~0ull, last = 0;
      for (int p = 0; p < numParts; p++) {
        const uint64_t *t = &h[static_cast<size_t>(8) * p];
        if (t[5] < first) first = t[5];
        if (t[4] > last) last = t[4];
        for (int k = 0; k < 4; k++) sum[k] += static_cast<double>(t[k + 1] - t[k]) * 0.01;
        resident += static_cast<double>(t[4] - t[5]) * 0.01;
      }
      const double span = static_cast<double>(last - first) * 0.01;
      {
        std::vector<double> ends, lives;
        for (int p = 0; p < numParts; p++) {
          ends.push_back(static_cast<double>(h[static_cast<size_t>(8) * p + 4] - first) * 0.01);
          lives.push_back(static_cast<double>(h[static_cast<size_t>(8) * p + 4] - h[static_cast<size_t>(8) * p + 5]) * 0.01);
        }
        std::sort(ends.begin(), ends.end());
        std::sort(lives.begin(), lives.end());
        auto q = [&](const std::vector<double> &v, double f) { return v[static_cast<size_t>(f * (v.size() - 1))]; };
        fprintf(stderr, "  workgroups done by: 50%% %.0f us, 90%% %.0f, 99%% %.0f, 100%% %.0f; lifetimes: median %.1f us, 90%% %.1f, 99%% %.1f, max %.1f; first starts spread %.0f us\n",
                q(ends, 0.5), q(ends, 0.9), q(ends, 0.99), q(ends, 1.0), q(lives, 0.5), q(lives, 0.9), q(lives, 0.99), q(lives, 1.0),
                static_cast<double>(h[5 + 8 * static_cast<size_t>(numParts - 1)] - first) * 0.01);
      }
      fprintf(stderr, "sr_merge_kernel<wide> phases (launch %d, %d partitions, prev %d, batch %d): span %.1f us, %.0f workgroups resident on average; per partition avg: clear %.1f + init %.1f + previous groups %.1f + records %.1f + order %.1f us\n",
              launches, numParts, prevSize, batchRows, span, resident / span, resident / numParts - (sum[0] + sum[1] + sum[2] + sum[3]) / numParts, sum[0] / numParts,
              sum[1] / numParts, sum[2] / numParts, sum[3] / numParts);
    }
  }
  buf.mark_idle();
  if (trace)
    fprintf(stderr, "fused_sort_reduce_vectors: rows %d prev %d (hashes %s) partBits %d/%d%s streams %d cap1 %u -> groups %u stream overflow %u table overflow %u emptykey %u\n",
            length, prevSize, prevKeys ? "known" : "unknown", pb1, partBits, spread ? " spread" : "", streams, cap1, w[0], w[1], w[3], w[2]);
  // a level-1 stream overflowed.  Hashes of the previous result unknown: its rows lie in hash order, tile after tile into one
  // stream each — again with the level-1 partitions spread (and more room: a workgroup's tiles still meet by chance);
  // otherwise with more room
  if (w[1] && !w[2] && !w[3] && !prevKeys && !spread) return sort_reduce_vectors_run(device, length, in, inValues, out, outValues, a, stream, slack + 1, true);
  if (w[1] && !w[2] && !w[3] && slack < 3) return sort_reduce_vectors_run(device, length, in, inValues, out, outValues, a, stream, slack + 1, spread);
  if (w[1] || w[2] || w[3]) return -1;  // the outputs may be partly written: the caller runs the real Sort + Reduce over them
  sorted_state_register(device, out, outValues, static_cast<size_t>(in.VectorCapacity), vw, static_cast<int>(w[0]), keysOut);
  return static_cast<int>(w[0]);
}

int fused_sort_reduce_vectors(int device, int length, const DimensionVector &in, const uint8_t *inValues, const DimensionVector &out,
                              uint8_t *outValues, const AggSpec &a, hipStream_t stream) {
  return sort_reduce_vectors_run(device, length, in, inValues, out, outValues, a, stream, record_stream_slack(), false);
}

}  // namespace ares
