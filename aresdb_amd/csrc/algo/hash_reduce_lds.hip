// HashReduce, partitioned: group-by aggregation in LDS instead of global memory.
//
// Why (tools/ubench_atomics.hip, MI355X): read-modify-writes on global memory run at ~24 G/s for
// the whole chip whatever the table size or scope, LDS atomics at 1.2-1.7 T/s.  So the global hash
// table of hash_reduce.hip is replaced, for the aggregates LDS can do natively, by two kernels:
//
//   1. partition: every workgroup (1024 lanes, one per CU) owns an 8192-slot hash table in LDS
//      (key = hash32 << 32 | row, value 8 bytes).  Rows are hashed (murmur3_x86_32 of the packed
//      dimension row, as the reference: query/hash_reduction.cu:216-243) and aggregated into it
//      with LDS atomics.  Before the table can overflow it is FLUSHED: its entries become 16-byte
//      records {row, hash, value}, counting-sorted by the top bits of the hash and appended to that
//      partition's region of a global buffer (one cursor reservation per partition per flush).
//      Low-cardinality input never flushes until the end: a workgroup emits one record per group.
//   2. merge: one workgroup per partition aggregates the partition's records in an LDS table and
//      emits final groups (dimension row of the representative + value).  A partition with more
//      groups than the table holds is processed in rounds over disjoint hash sub-ranges chosen
//      between rounds only (range halves after an overflowing attempt, doubles after a sparse one),
//      so membership of a group in a round never depends on timing.
//
// Group identity, representative row (lowest row index, via 64-bit atomic min on the key) and the
// unspecified output order are exactly those of hash_reduce.hip / the reference.  A partition
// region that overflows (adversarial hash skew) makes the host fall back to the global-table path.
#include <hip/hip_runtime.h>

#include "aggregate.hpp"
#include "common.hpp"
#include "device_model.hpp"
#include "ares_extensions.h"
#include "dim_layout.hpp"
#include "fast_eval.hpp"
#include "hash_reduce_lds.hpp"

namespace ares {

namespace {

constexpr int kThreads = 1024;
constexpr int kSlots = 8192;              // LDS table slots (16 bytes each: 128 KiB)
constexpr int kSlotMask = kSlots - 1;
constexpr int kRowsPerLane = 2;           // rows inserted per lane between occupancy checks
constexpr int kTileRows = kThreads * kRowsPerLane;
constexpr int kFlushAt = kSlots * 3 / 4 - kTileRows;  // flush when more entries than this are held
constexpr int kMergeLimit = kSlots - kThreads - 128;   // groups per merge round (every lane may claim one more)
constexpr int kMaxPartitions = 512;
constexpr uint64_t kEmpty = ~0ull;

struct Workspace {
  uint4 *records;        // numPartitions regions of `cap` records
  uint32_t *cursors;     // records appended per partition
  uint32_t *outCount;    // groups emitted
  uint32_t *overflow;    // a partition region overflowed
  uint64_t cap;
  int partBits;
  int debug;  // ARES_HR_DEBUG: timing experiments only (results are wrong when set)
};

__device__ __forceinline__ void lds_aggregate(uint64_t *slot, uint64_t bits, const AggSpec &a) {
  switch (a.vtype) {
    case V_F64:
      __hip_atomic_fetch_add(reinterpret_cast<double *>(slot), __longlong_as_double(static_cast<long long>(bits)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    case V_U64: case V_I64:
      __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(slot), static_cast<unsigned long long>(bits),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    case V_F32:
      __hip_atomic_fetch_add(reinterpret_cast<float *>(slot), bits_f(static_cast<uint32_t>(bits)), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    case V_U32: {
      uint32_t *p = reinterpret_cast<uint32_t *>(slot);
      const uint32_t x = static_cast<uint32_t>(bits);
      if (a.op == OP_SUM) __hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (a.op == OP_MIN) __hip_atomic_fetch_min(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_max(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    }
    default: {  // V_I32
      int32_t *p = reinterpret_cast<int32_t *>(slot);
      const int32_t x = static_cast<int32_t>(static_cast<uint32_t>(bits));
      if (a.op == OP_SUM) __hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (a.op == OP_MIN) __hip_atomic_fetch_min(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_max(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    }
  }
}

// Finds or claims the slot of hash h (linear probing); lowers the key to min(key, h<<32|row).
// `claims` counts successful claims.  The caller guarantees the table cannot fill up.
__device__ __forceinline__ int lds_find_or_claim(uint64_t *keys, uint32_t h, uint32_t row, uint32_t *claims) {
  const uint64_t mine = (static_cast<uint64_t>(h) << 32) | row;
  int slot = static_cast<int>(h) & kSlotMask;
  for (;;) {
    uint64_t cur = keys[slot];
    if (cur == kEmpty) {
      unsigned long long expected = kEmpty;
      if (__hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned long long *>(keys + slot), &expected,
                                               static_cast<unsigned long long>(mine), __ATOMIC_RELAXED,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
        __hip_atomic_fetch_add(claims, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return slot;
      }
      cur = expected;
    }
    if (static_cast<uint32_t>(cur >> 32) == h) {
      if (mine < cur)
        __hip_atomic_fetch_min(reinterpret_cast<unsigned long long *>(keys + slot), static_cast<unsigned long long>(mine),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return slot;
    }
    slot = (slot + 1) & kSlotMask;
  }
}

__device__ __forceinline__ void clear_table(uint64_t *keys, uint64_t *vals, uint64_t identity) {
  for (int s = threadIdx.x; s < kSlots; s += kThreads) {
    keys[s] = kEmpty;
    vals[s] = identity;
  }
}

// ---- kernel 1: aggregate in LDS, spill hash-partitioned records ----------------------------------
__global__ __launch_bounds__(kThreads) void hr_partition_kernel(const uint8_t *dimValues, DimLayoutD L, size_t capacity,
                                                                const uint8_t *inputValues, AggSpec a, int length,
                                                                Workspace ws) {
  __shared__ uint64_t sKeys[kSlots];
  __shared__ uint64_t sVals[kSlots];
  __shared__ uint32_t sPartCount[kMaxPartitions];
  __shared__ uint32_t sPartBase[kMaxPartitions];
  __shared__ uint32_t sClaims;
  const int numParts = 1 << ws.partBits;
  clear_table(sKeys, sVals, a.identity);
  for (int p = threadIdx.x; p < numParts; p += kThreads) sPartCount[p] = 0;
  if (threadIdx.x == 0) sClaims = 0;
  __syncthreads();

  const int64_t numTiles = (static_cast<int64_t>(length) + kTileRows - 1) / kTileRows;
  for (int64_t tile = blockIdx.x;; tile += gridDim.x) {
    const bool more = tile < numTiles;
    if (more) {
#pragma unroll
      for (int k = 0; k < kRowsPerLane; k++) {
        const int64_t i = tile * kTileRows + static_cast<int64_t>(k) * kThreads + threadIdx.x;
        if (i < length) {
          const uint32_t row = static_cast<uint32_t>(i);
          Murmur32Stream ms(0);
          hash_dim_row(ms, dimValues, L, capacity, row);
          const uint32_t h = ms.finish();
          const uint64_t v = load_value_bits(inputValues, a, row);
          if (ws.debug & 1) {  // experiment: loads + hash only
            if (h == 0x12345678u && v == 0x9abcdef012345678ull) sClaims = 1;
          } else {
            const int slot = lds_find_or_claim(sKeys, h, row, &sClaims);
            lds_aggregate(sVals + slot, v, a);
          }
        }
      }
    }
    __syncthreads();
    const uint32_t held = sClaims;
    if (more && held <= static_cast<uint32_t>(kFlushAt)) continue;  // uniform: sClaims is read after the barrier
    if (held > 0) {
      // ---- flush: counting sort of the entries by partition, one cursor reservation each ----
      constexpr int kPerLane = kSlots / kThreads;
      uint32_t rank[kPerLane];
#pragma unroll
      for (int k = 0; k < kPerLane; k++) {
        const uint64_t key = sKeys[threadIdx.x + k * kThreads];
        rank[k] = 0;
        if (key != kEmpty) {
          const uint32_t p = ws.partBits ? static_cast<uint32_t>(key >> (64 - ws.partBits)) : 0u;
          rank[k] = __hip_atomic_fetch_add(&sPartCount[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      __syncthreads();
      for (int p = threadIdx.x; p < numParts; p += kThreads) {
        const uint32_t c = sPartCount[p];
        if (c) {
          sPartBase[p] = atomicAdd(ws.cursors + p, c);
          sPartCount[p] = 0;
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < kPerLane; k++) {
        const int s = threadIdx.x + k * kThreads;
        const uint64_t key = sKeys[s];
        if (key != kEmpty) {
          const uint32_t p = ws.partBits ? static_cast<uint32_t>(key >> (64 - ws.partBits)) : 0u;
          const uint64_t at = static_cast<uint64_t>(sPartBase[p]) + rank[k];
          const uint64_t v = sVals[s];
          if (ws.debug & 2) {  // experiment: no record stores
          } else if (at < ws.cap) {
            ws.records[static_cast<uint64_t>(p) * ws.cap + at] =
                make_uint4(static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32), static_cast<uint32_t>(v),
                           static_cast<uint32_t>(v >> 32));
          } else {
            *ws.overflow = 1u;
          }
          sKeys[s] = kEmpty;
          sVals[s] = a.identity;
        }
      }
      if (threadIdx.x == 0) sClaims = 0;
      __syncthreads();
    }
    if (!more) break;
  }
}


// ---- kernel 1, hot layout: every dimension 4 bytes wide --------------------------------------------
// Each lane owns 4 consecutive rows per tile: the ND value vectors are read 16 bytes per lane, the
// validity bytes 4 per lane, the measures 16/32 bytes per lane (byte-aligned vector accesses are
// native on gfx950), and the NEXT tile's loads are issued before the current tile is hashed and
// inserted, so HBM latency hides behind the LDS work.
//
// Two modes per workgroup, chosen from what the data does:
//   * TABLE  — aggregate into the LDS hash table, flush when it fills (as the generic kernel);
//   * DIRECT — when a flush shows (almost) no duplicates inside a tile's reach, aggregation in the
//     table is pure overhead: rows become records straight away, counting-sorted by partition in
//     LDS (the table's memory) and written back with coalesced 16-byte stores.
struct __attribute__((packed, aligned(1))) PU32x4 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) PU32 { uint32_t v; };
struct __attribute__((packed, aligned(1))) PU64x2 { uint64_t v[2]; };

template <int ND>
struct QuadRows {
  uint32_t dim[ND][4];
  uint32_t nul[ND];  // 4 validity bytes
  uint64_t val[4];
};

template <int ND, int VW>
__device__ __forceinline__ void load_quad(QuadRows<ND> &q, const uint8_t *dimValues, size_t capacity,
                                          const uint8_t *inputValues, int64_t i0, int length) {
  const uint8_t *nulls = dimValues + static_cast<size_t>(4 * ND) * capacity;
  if (i0 + 3 < length) {
#pragma unroll
    for (int d = 0; d < ND; d++) {
      const PU32x4 v = *reinterpret_cast<const PU32x4 *>(dimValues + static_cast<size_t>(4 * d) * capacity + 4 * i0);
#pragma unroll
      for (int j = 0; j < 4; j++) q.dim[d][j] = v.v[j];
      q.nul[d] = reinterpret_cast<const PU32 *>(nulls + static_cast<size_t>(d) * capacity + i0)->v;
    }
    if (VW == 8) {
      const PU64x2 a = *reinterpret_cast<const PU64x2 *>(inputValues + 8 * i0);
      const PU64x2 b = *reinterpret_cast<const PU64x2 *>(inputValues + 8 * i0 + 16);
      q.val[0] = a.v[0]; q.val[1] = a.v[1]; q.val[2] = b.v[0]; q.val[3] = b.v[1];
    } else {
      const PU32x4 v = *reinterpret_cast<const PU32x4 *>(inputValues + 4 * i0);
#pragma unroll
      for (int j = 0; j < 4; j++) q.val[j] = v.v[j];
    }
  } else {
#pragma unroll
    for (int d = 0; d < ND; d++) {
      q.nul[d] = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        q.dim[d][j] = 0;
        if (i0 + j < length) {
          q.dim[d][j] = *reinterpret_cast<const uint32_t *>(dimValues + static_cast<size_t>(4 * d) * capacity + 4 * (i0 + j));
          q.nul[d] |= static_cast<uint32_t>(nulls[static_cast<size_t>(d) * capacity + i0 + j]) << (8 * j);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      q.val[j] = 0;
      if (i0 + j < length)
        q.val[j] = VW == 8 ? *reinterpret_cast<const uint64_t *>(inputValues + 8 * (i0 + j))
                           : static_cast<uint64_t>(*reinterpret_cast<const uint32_t *>(inputValues + 4 * (i0 + j)));
    }
  }
}

template <int ND>
__device__ __forceinline__ uint32_t hash_quad_row(const QuadRows<ND> &q, int j) {
  Murmur32Stream ms(0);
#pragma unroll
  for (int d = 0; d < ND; d++) ms.push(q.dim[d][j], 4);
#pragma unroll
  for (int d = 0; d < ND; d++) ms.push((q.nul[d] >> (8 * j)) & 0xFFu, 1);
  return ms.finish();
}

constexpr int kQuadTile = kThreads * 4;                   // rows per tile of the hot-layout kernel
constexpr int kQuadFlushAt = kSlots * 3 / 4 - kQuadTile;  // = 2048

// Row source "dimension vector": the ABI's HashReduce input (values per dimension, validity bytes,
// measures), already projected by the transform calls.
template <int ND_, int VW>
struct DimVectorSource {
  static constexpr int ND = ND_;
  static constexpr bool kPairDirect = true;  // registers allow two tiles per partition sort
  using Raw = QuadRows<ND_>;
  const uint8_t *dimValues;
  size_t capacity;
  const uint8_t *inputValues;
  uint32_t rowBase;
  __device__ __forceinline__ void prepare() {}
  __device__ __forceinline__ void load(Raw &r, int64_t i0, int length) const {
    load_quad<ND_, VW>(r, dimValues, capacity, inputValues, i0, length);
  }
  // hash + measure bits of the quad's four rows; returns which of them take part (4-bit mask)
  __device__ __forceinline__ uint32_t rows(const Raw &r, int64_t i0, int length, uint32_t (&h)[4], uint64_t (&v)[4]) const {
    uint32_t alive = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      alive |= (i0 + j < length ? 1u : 0u) << j;
      h[j] = hash_quad_row<ND_>(r, j);
      v[j] = r.val[j];
    }
    return alive;
  }
};

// The partition kernel body, shared by the ABI path (DimVectorSource) and the fused scan
// (FusedSource: filter + projection evaluated on the fly from the source columns).
template <typename Source>
__device__ __forceinline__ void partition_body(Source &src, const AggSpec &a, int length, const Workspace &ws) {
  // TABLE mode: sKeys / sVals are the hash table.  DIRECT mode: the same 128 KiB stage up to 4096
  // sorted records (sKeys[k] = {row, hash}, sVals[k] = value).
  __shared__ uint64_t sKeys[kSlots];
  __shared__ uint64_t sVals[kSlots];
  __shared__ uint32_t sPartCount[kMaxPartitions];
  __shared__ uint32_t sPartBase[kMaxPartitions];   // global base of the partition's run
  __shared__ uint32_t sPartLocal[kMaxPartitions];  // DIRECT: first staged index of the partition
  __shared__ uint32_t sWaveSum[kThreads / 64];
  __shared__ uint32_t sClaims, sStaged;
  const int numParts = 1 << ws.partBits;
  const int pb = ws.partBits;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  clear_table(sKeys, sVals, a.identity);
  for (int p = threadIdx.x; p < numParts; p += kThreads) sPartCount[p] = 0;
  if (threadIdx.x == 0) sClaims = 0;
  __syncthreads();
  src.prepare();

  const int64_t numTiles = (static_cast<int64_t>(length) + kQuadTile - 1) / kQuadTile;
  bool direct = false;
  uint32_t rowsSinceFlush = 0;
  typename Source::Raw buf;
  int64_t tile = blockIdx.x;
  if (tile < numTiles) src.load(buf, tile * kQuadTile + 4 * threadIdx.x, length);

  // processes one tile (hash + measure bits of the lane's four rows); returns false when the
  // workgroup is done
  auto step = [&](const uint32_t (&h)[4], const uint64_t (&v)[4], const uint32_t alive, int64_t t) -> bool {
    const bool more = t < numTiles;
    const int64_t i0 = t * kQuadTile + 4 * static_cast<int64_t>(threadIdx.x);
    if (more && !direct) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if ((alive >> j) & 1u) {
          if (ws.debug & 1) {
            if (h[j] == 0x12345678u && v[j] == 0x9abcdef012345678ull) sClaims = 1;
          } else {
            const int slot = lds_find_or_claim(sKeys, h[j], src.rowBase + static_cast<uint32_t>(i0 + j), &sClaims);
            lds_aggregate(sVals + slot, v[j], a);
          }
        }
      }
      rowsSinceFlush += kQuadTile;
    }
    if (!direct) {
      __syncthreads();
      const uint32_t held = sClaims;
      if (more && held <= static_cast<uint32_t>(kQuadFlushAt)) return true;
      if (held > 0) {
        // ---- flush the table: counting sort by partition, one cursor reservation each ----
        constexpr int kPerLane = kSlots / kThreads;
        uint32_t rank[kPerLane];
#pragma unroll
        for (int k = 0; k < kPerLane; k++) {
          const uint64_t key = sKeys[threadIdx.x + k * kThreads];
          rank[k] = 0;
          if (key != kEmpty) {
            const uint32_t p = pb ? static_cast<uint32_t>(key >> (64 - pb)) : 0u;
            rank[k] = __hip_atomic_fetch_add(&sPartCount[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        __syncthreads();
        for (int p = threadIdx.x; p < numParts; p += kThreads) {
          const uint32_t c = sPartCount[p];
          if (c) {
            sPartBase[p] = atomicAdd(ws.cursors + p, c);
            sPartCount[p] = 0;
          }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPerLane; k++) {
          const int s = threadIdx.x + k * kThreads;
          const uint64_t key = sKeys[s];
          if (key != kEmpty) {
            const uint32_t p = pb ? static_cast<uint32_t>(key >> (64 - pb)) : 0u;
            const uint64_t at = static_cast<uint64_t>(sPartBase[p]) + rank[k];
            const uint64_t v = sVals[s];
            if (ws.debug & 2) {
            } else if (at < ws.cap) {
              ws.records[static_cast<uint64_t>(p) * ws.cap + at] =
                  make_uint4(static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32), static_cast<uint32_t>(v),
                             static_cast<uint32_t>(v >> 32));
            } else {
              *ws.overflow = 1u;
            }
            sKeys[s] = kEmpty;
            sVals[s] = a.identity;
          }
        }
        // (almost) every row became its own entry: stop aggregating, just partition
        if (more && !(ws.debug & 4) && static_cast<uint64_t>(held) * 5 > static_cast<uint64_t>(rowsSinceFlush) * 4)
          direct = true;
        rowsSinceFlush = 0;
        if (threadIdx.x == 0) sClaims = 0;
        __syncthreads();
      }
      return more;
    }
    return more;  // DIRECT tiles are handled in pairs by direct_pair below
  };

  // ---- DIRECT: rows -> records, counting-sorted by partition in LDS, coalesced write-back ----
  // Two tiles (8192 rows, the whole 128 KiB stage) per sort: one cursor reservation per partition
  // and one set of barriers per 8192 rows, runs of ~8192 / numParts records per partition.
  auto direct_pair = [&](const uint32_t (&h)[8], const uint64_t (&v)[8], const uint32_t alive, const uint32_t (&rowId)[8]) {
    uint32_t rank[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      rank[j] = 0;
      if ((alive >> j) & 1u) {
        const uint32_t p = pb ? h[j] >> (32 - pb) : 0u;
        rank[j] = __hip_atomic_fetch_add(&sPartCount[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
    uint32_t reservedBase = 0;
    bool reserved = false;
    {  // exclusive scan of the partition counts (numParts <= kThreads), global reservation
      const uint32_t c = threadIdx.x < static_cast<uint32_t>(numParts) ? sPartCount[threadIdx.x] : 0u;
      uint32_t incl = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t tmp = __shfl_up(incl, off);
        if (lane >= off) incl += tmp;
      }
      if (lane == 63) sWaveSum[wave] = incl;
      __syncthreads();
      uint32_t before = 0;
      for (int w = 0; w < wave; w++) before += sWaveSum[w];
      if (threadIdx.x < static_cast<uint32_t>(numParts)) {
        sPartLocal[threadIdx.x] = before + incl - c;
        // the reservation's round trip (a returning global atomic, ~2 us) overlaps the staging below:
        // its result is only published to LDS right before the write-back needs it
        if (c) reservedBase = (ws.debug & 32) ? 0u : atomicAdd(ws.cursors + threadIdx.x, c);
        reserved = c != 0;
        sPartCount[threadIdx.x] = 0;
        if (threadIdx.x == static_cast<uint32_t>(numParts) - 1) sStaged = before + incl;
      }
    }
    __syncthreads();
    if (ws.debug & 64) return;  // experiment: hash + count only
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if ((alive >> j) & 1u) {
        const uint32_t p = pb ? h[j] >> (32 - pb) : 0u;
        const uint32_t at = sPartLocal[p] + rank[j];
        sKeys[at] = (static_cast<uint64_t>(h[j]) << 32) | rowId[j];
        sVals[at] = v[j];
      }
    }
    if (reserved) sPartBase[threadIdx.x] = reservedBase;
    __syncthreads();
    {
      const uint32_t staged = sStaged;
      for (uint32_t k = threadIdx.x; k < staged; k += kThreads) {
        const uint64_t key = sKeys[k];
        const uint64_t val = sVals[k];
        const uint32_t p = pb ? static_cast<uint32_t>(key >> (64 - pb)) : 0u;
        const uint64_t at = static_cast<uint64_t>(sPartBase[p]) + (k - sPartLocal[p]);
        if (ws.debug & 2) {
        } else if (at < ws.cap) {
          ws.records[static_cast<uint64_t>(p) * ws.cap + at] =
              make_uint4(static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32), static_cast<uint32_t>(val),
                         static_cast<uint32_t>(val >> 32));
        } else {
          *ws.overflow = 1u;
        }
      }
    }
    __syncthreads();  // the stage is reused by the next pair
  };

  // TABLE mode: one register buffer is enough to overlap HBM latency with the LDS work — the tile's
  // rows are reduced to (hash, measure) pairs first, then the NEXT tile's loads are issued into the
  // same registers before the current tile goes through the table.
  for (;;) {
    uint32_t h[4];
    uint64_t v[4];
    const uint32_t alive = src.rows(buf, tile * kQuadTile + 4 * static_cast<int64_t>(threadIdx.x), length, h, v);  // 0 past the end
    const int64_t next = tile + gridDim.x;
    if (next < numTiles) src.load(buf, next * kQuadTile + 4 * threadIdx.x, length);
    const bool more = step(h, v, alive, tile);
    tile = next;
    if (!more) return;
    if (direct) break;
  }
  if constexpr (Source::kPairDirect) {
    // DIRECT mode: `buf` holds tile `tile` (when it exists); a second buffer takes its partner.
    typename Source::Raw buf2;
    {
      const int64_t partner = tile + gridDim.x;
      if (partner < numTiles) src.load(buf2, partner * kQuadTile + 4 * threadIdx.x, length);
    }
    while (tile < numTiles) {
      const int64_t tileB = tile + gridDim.x;
      const int64_t i0A = tile * kQuadTile + 4 * static_cast<int64_t>(threadIdx.x);
      const int64_t i0B = tileB * kQuadTile + 4 * static_cast<int64_t>(threadIdx.x);
      uint32_t h[8], rowId[8];
      uint64_t v[8];
      uint32_t alive;
      {
        uint32_t hA[4], hB[4];
        uint64_t vA[4], vB[4];
        const uint32_t aliveA = src.rows(buf, i0A, length, hA, vA);
        const uint32_t aliveB = tileB < numTiles ? src.rows(buf2, i0B, length, hB, vB) : 0u;
        alive = aliveA | (aliveB << 4);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          h[j] = hA[j]; v[j] = vA[j]; rowId[j] = src.rowBase + static_cast<uint32_t>(i0A + j);
          h[4 + j] = hB[j]; v[4 + j] = vB[j]; rowId[4 + j] = src.rowBase + static_cast<uint32_t>(i0B + j);
        }
      }
      const int64_t nextA = tileB + gridDim.x, nextB = nextA + gridDim.x;
      if (nextA < numTiles) src.load(buf, nextA * kQuadTile + 4 * threadIdx.x, length);
      if (nextB < numTiles) src.load(buf2, nextB * kQuadTile + 4 * threadIdx.x, length);
      direct_pair(h, v, alive, rowId);
      tile = nextA;
    }
  } else {
    while (tile < numTiles) {  // one tile per sort
      const int64_t i0 = tile * kQuadTile + 4 * static_cast<int64_t>(threadIdx.x);
      uint32_t h[8], rowId[8];
      uint64_t v[8];
      uint32_t alive;
      {
        uint32_t hA[4];
        uint64_t vA[4];
        alive = src.rows(buf, i0, length, hA, vA);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          h[j] = hA[j]; v[j] = vA[j]; rowId[j] = src.rowBase + static_cast<uint32_t>(i0 + j);
          h[4 + j] = 0; v[4 + j] = 0; rowId[4 + j] = 0;
        }
      }
      const int64_t next = tile + gridDim.x;
      if (next < numTiles) src.load(buf, next * kQuadTile + 4 * threadIdx.x, length);
      direct_pair(h, v, alive, rowId);
      tile = next;
    }
  }
}

template <int ND, int VW>
__global__ __launch_bounds__(kThreads) void hr_partition4_kernel(const uint8_t *dimValues, size_t capacity,
                                                                 const uint8_t *inputValues, AggSpec a, int length,
                                                                 Workspace ws) {
  DimVectorSource<ND, VW> src{dimValues, capacity, inputValues, 0u};
  partition_body(src, a, length, ws);
}


// ---- row source "fused scan": filter + projection evaluated from the source columns -------------
// (extension entry point AresFusedFilterHashReduce, include/ares_extensions.h).  The batch's columns
// are read once, 16 bytes per lane; the conjunction of comparison filters decides which rows take
// part; dimensions and the measure are evaluated with the very functions the transform kernels use
// (fast_eval.hpp), so the (value, validity) pairs that are hashed are bit-identical to what the
// UnaryTransform / BinaryTransform calls would have stored in the dimension vector.
struct FusedConst {
  DVal y;
  FastDivisor fd;
};
__device__ __forceinline__ FusedConst fused_const(const FastOperands &f) {
  FusedConst c;
  c.y.bits = f.bbits;
  c.y.ok = f.bok;
  c.y = cvt32(c.y, f.bkind, f.I);
  const uint32_t mag = (f.I == K_I32 && static_cast<int32_t>(c.y.bits) < 0) ? 0u - c.y.bits : c.y.bits;
  c.fd = make_fast_divisor(mag);
  return c;
}

// measure bits of an evaluated value (MeasureProxy, query/iterator.hpp:616-647, no run lengths)
__device__ __forceinline__ uint64_t fused_measure_bits(const FusedPlanD &p, DVal r) {
  const int rk = p.measure.f.rk;
  if (!r.ok) return p.identity;
  if (p.measureWidth == 8) {
    if (p.measureDtype == Float64) return static_cast<uint64_t>(__double_as_longlong(to_double32(r, rk)));
    return static_cast<uint64_t>(rk == K_F32 ? static_cast<int64_t>(bits_f(r.bits))
                                 : rk == K_I32 ? static_cast<int64_t>(static_cast<int32_t>(r.bits))
                                               : static_cast<int64_t>(r.bits));
  }
  return cvt32(r, rk, p.measureDtype == Int32 ? K_I32 : p.measureDtype == Uint32 ? K_U32 : K_F32).bits;
}

template <int ND_>
struct FusedSource {
  static constexpr int ND = ND_;
  static constexpr bool kPairDirect = false;  // expression evaluation needs the registers
  static constexpr int NC = ND_ + 2;  // distinct columns a plan of ND dimensions may touch
  struct Raw {
    uint32_t v[NC][4];
    uint32_t win[NC];  // 16-bit validity window starting at the byte of the quad's first row
  };
  const FusedPlanD &plan;
  uint32_t rowBase;
  FusedConst fc[kFusedFilters], dc[ND_], mc;

  __device__ __forceinline__ FusedSource(const FusedPlanD &p, uint32_t base) : plan(p), rowBase(base) {}
  __device__ __forceinline__ void prepare() {
#pragma unroll
    for (int k = 0; k < kFusedFilters; k++)
      if (k < plan.numFilters) fc[k] = fused_const(plan.filters[k].f);
#pragma unroll
    for (int d = 0; d < ND_; d++) dc[d] = fused_const(plan.dims[d].f);
    mc = fused_const(plan.measure.f);
  }
  __device__ __forceinline__ void load(Raw &r, int64_t i0, int length) const {
#pragma unroll
    for (int c = 0; c < NC; c++) {
      if (c >= plan.numCols) continue;
      const FusedColumn col = plan.cols[c];
      if (i0 + 3 < length) {
        const PU32x4 v = *reinterpret_cast<const PU32x4 *>(col.vals + i0);
#pragma unroll
        for (int j = 0; j < 4; j++) r.v[c][j] = v.v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) r.v[c][j] = i0 + j < length ? col.vals[i0 + j] : 0u;
      }
      r.win[c] = 0xFFFFu;
      if (col.nulls && i0 < length)
        r.win[c] = reinterpret_cast<const PU16 *>(col.nulls + ((static_cast<uint32_t>(i0) + col.bitOff) >> 3))->v;
    }
  }
  __device__ __forceinline__ uint32_t rows(const Raw &r, int64_t i0, int length, uint32_t (&h)[4], uint64_t (&v)[4]) const {
    uint32_t ok[NC];  // validity nibble of the quad per column
#pragma unroll
    for (int c = 0; c < NC; c++) {
      ok[c] = 0xFu;
      if (c < plan.numCols) ok[c] = (r.win[c] >> ((static_cast<uint32_t>(i0) + plan.cols[c].bitOff) & 7u)) & 0xFu;
    }
    // one dispatch per expression per quad (eval_quad / compare_tile), not per element
    uint32_t in[1] = {0u};
#pragma unroll
    for (int j = 0; j < 4; j++) in[0] |= (i0 + j < length ? 1u : 0u) << j;
    uint32_t alive = in[0];
#pragma unroll
    for (int k = 0; k < kFusedFilters; k++) {
      if (k < plan.numFilters) {
        const FusedExpr &e = plan.filters[k];
        uint32_t fv[1][4] = {{0u, 0u, 0u, 0u}}, fok[1] = {0u}, kb[1];
#pragma unroll
        for (int c = 0; c < NC; c++)
          if (c == e.col) {
#pragma unroll
            for (int j = 0; j < 4; j++) fv[0][j] = r.v[c][j];
            fok[0] = ok[c];
          }
        compare_tile<1>(e.f, fv, fok, in, fc[k].y, kb);
        alive &= kb[0];
      }
    }
    // dimension d reads column slot d, the measure slot ND (fixed by the host), so only the filters
    // select their operand at run time
    uint32_t dimBits[ND_][4], dimOk[ND_];
#pragma unroll
    for (int d = 0; d < ND_; d++) {
      const FusedExpr &e = plan.dims[d];
      uint32_t rb[4];
      dimOk[d] = eval_quad(e.f, r.v[d], ok[d], dc[d].y, dc[d].fd, rb);
      const bool plain = e.f.rk == e.outKind || (e.f.rk != K_F32 && e.outKind != K_F32 && e.f.rk != K_BOOL);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (plain) {
          dimBits[d][j] = rb[j];
        } else {
          DVal x;
          x.bits = rb[j];
          x.ok = 1;
          dimBits[d][j] = cvt32(x, e.f.rk, e.outKind).bits;
        }
      }
    }
    uint32_t mb[4];
    const uint32_t mok = eval_quad(plan.measure.f, r.v[ND_], ok[ND_], mc.y, mc.fd, mb);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      Murmur32Stream ms(0);
#pragma unroll
      for (int d = 0; d < ND_; d++) ms.push(dimBits[d][j], 4);
#pragma unroll
      for (int d = 0; d < ND_; d++) ms.push((dimOk[d] >> j) & 1u, 1);
      h[j] = ms.finish();
      DVal x;
      x.bits = mb[j];
      x.ok = (mok >> j) & 1u;
      v[j] = fused_measure_bits(plan, x);
    }
    return alive;
  }
};

template <int ND>
__global__ __launch_bounds__(kThreads) void hr_fused_scan_kernel(FusedPlanD plan, uint32_t rowBase, AggSpec a, int length,
                                                                 Workspace ws) {
  FusedSource<ND> src(plan, rowBase);
  partition_body(src, a, length, ws);
}

// dimension values + validity of ONE source row (group representative), for the merge's emission
template <int ND>
__device__ __forceinline__ void fused_eval_row(const FusedPlanD &plan, uint32_t row, uint32_t (&bits)[ND], uint32_t (&ok)[ND]) {
#pragma unroll
  for (int d = 0; d < ND; d++) {
    const FusedExpr &e = plan.dims[d];
    const FusedColumn col = plan.cols[e.col];
    const FusedConst c = fused_const(e.f);
    const uint32_t raw = col.vals[row];
    const uint32_t rok = col.nulls ? get_bit(col.nulls, row + col.bitOff) : 1u;
    const DVal x = eval_fast(e.f, raw, rok, c.y, c.fd);
    bits[d] = cvt32(x, e.f.rk, e.outKind).bits;
    ok[d] = x.ok ? 1u : 0u;
  }
}

// ---- kernel 2: per-partition merge in LDS, emit groups ---------------------------------------------
// One record into the round's table; claims a slot only while the round's attempt budget lasts.
__device__ __forceinline__ void merge_record(uint64_t *sKeys, uint64_t *sVals, uint32_t *sClaimed, uint32_t *sOverflow,
                                             const uint4 r, const AggSpec &a) {
  // a round that has run out of slots is discarded as a whole: stop filling the table (every lane
  // claims at most one more slot after the flag is up, so the table can never fill completely and
  // the probe loop below always terminates)
  if (__hip_atomic_load(sOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
  const uint32_t h = r.y;
  const uint64_t mine = (static_cast<uint64_t>(h) << 32) | r.x;
  int slot = static_cast<int>(h) & kSlotMask;
  for (;;) {
    uint64_t cur = sKeys[slot];
    if (cur == kEmpty) {
      unsigned long long expected = kEmpty;
      if (__hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned long long *>(sKeys + slot), &expected,
                                               static_cast<unsigned long long>(mine), __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP)) {
        // occupied slots are counted exactly (lost races are not new groups)
        if (__hip_atomic_fetch_add(sClaimed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >=
            static_cast<uint32_t>(kMergeLimit))
          __hip_atomic_store(sOverflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        break;
      }
      cur = expected;
    }
    if (static_cast<uint32_t>(cur >> 32) == h) {
      if (mine < cur)
        __hip_atomic_fetch_min(reinterpret_cast<unsigned long long *>(sKeys + slot), static_cast<unsigned long long>(mine),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    }
    slot = (slot + 1) & kSlotMask;
  }
  lds_aggregate(sVals + slot, (static_cast<uint64_t>(r.w) << 32) | r.z, a);
}

constexpr int kMergeBatch = 4;  // records per lane per pipeline stage

// ND4 = number of dimensions when all are 4 bytes wide (vectorisable emission), 0 = any layout
// FUSED: rows >= prevSize are source rows of the fused scan — their dimensions are re-evaluated
// from the columns; rows < prevSize are previous results in dimIn (stride prevCapacity).
template <int ND4, bool FUSED>
__device__ __forceinline__ void merge_body(const uint8_t *__restrict__ dimIn, size_t inCapacity, uint8_t *__restrict__ dimOut,
                                           const DimLayoutD &L, size_t capacity, uint8_t *__restrict__ outputValues,
                                           const AggSpec &a, const Workspace &ws, const FusedPlanD *plan, uint32_t prevSize) {
  __shared__ uint64_t sKeys[kSlots];
  __shared__ uint64_t sVals[kSlots];
  __shared__ uint32_t sAttempts, sOverflow, sCount, sBase, sClaims, sProgress;
  const int p = blockIdx.x;
  const uint32_t cursor = ws.cursors[p];
  const uint64_t n = cursor < ws.cap ? cursor : ws.cap;
  if (n == 0) return;
  const uint4 *__restrict__ rec = ws.records + static_cast<uint64_t>(p) * ws.cap;
  // Rounds over sub-ranges of the hash bits below the partition bits (left-aligned to 32 bits).
  // The first round is optimistic — the whole range: a partition usually holds far fewer groups than
  // the table has slots.  A round that runs out of slots stops streaming at once, and how far it
  // got tells how much narrower the next attempt must be, so a wrong guess costs a partial pass.
  const int pb = ws.partBits;
  uint64_t lo = 0, width = 1ull << 32;
  const uint64_t stride = static_cast<uint64_t>(kMergeBatch) * kThreads;
  while (lo < (1ull << 32)) {
    clear_table(sKeys, sVals, a.identity);
    if (threadIdx.x == 0) { sAttempts = 0; sOverflow = 0; sCount = 0; sClaims = 0; }
    __syncthreads();
    const uint64_t hi = lo + width;
    // two register stages: the loads of the next kMergeBatch records per lane are in flight while
    // the current ones go through the LDS table.  The loads are unconditional (the index is clamped
    // to the last record; `consume` ignores positions past the end): with branches around them the
    // compiler cannot count outstanding loads and waits for the prefetch as well.
    uint4 ra[kMergeBatch], rb[kMergeBatch];
    auto load = [&](uint4 (&r)[kMergeBatch], uint64_t base) {
#pragma unroll
      for (int k = 0; k < kMergeBatch; k++) {
        const uint64_t i = base + static_cast<uint64_t>(k) * kThreads + threadIdx.x;
        r[k] = rec[i < n ? i : n - 1];
      }
    };
    auto consume = [&](const uint4 (&r)[kMergeBatch], uint64_t base) {
#pragma unroll
      for (int k = 0; k < kMergeBatch; k++) {
        const uint64_t i = base + static_cast<uint64_t>(k) * kThreads + threadIdx.x;
        const uint32_t h = r[k].y;
        const uint64_t u = static_cast<uint64_t>(pb ? (h << pb) : h);
        if (i < n && u >= lo && u < hi) {
          if (ws.debug & 8) {  // experiment: loads only
            if (h == 0x12345678u && r[k].z == 0x9abcdefu) sOverflow = 1u;
          } else {
            merge_record(sKeys, sVals, &sAttempts, &sOverflow, r[k], a);
          }
        }
      }
    };
    load(ra, 0);
    uint64_t base = 0;
    for (; base < n; base += 2 * stride) {
      if (__hip_atomic_load(&sOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) && !(ws.debug & 8)) break;
      load(rb, base + stride);
      consume(ra, base);
      load(ra, base + 2 * stride);
      consume(rb, base + stride);
    }
    if (threadIdx.x == 0) sProgress = static_cast<uint32_t>(base < n ? base : n);
    __syncthreads();
    const bool overflowed = sOverflow != 0 && !(ws.debug & 8);
    const uint64_t progress = sProgress;
    __syncthreads();  // everyone has read the flags before the next round resets them
    if (overflowed && width > 1) {  // too many groups in this sub-range: nothing is emitted
      // ~kMergeLimit groups showed up in the first `progress` records: aim for half a table per round
      uint64_t shrink = 2 * n / (progress > stride ? progress : stride);
      do {  // (a single hash value cannot overflow the table)
        width >>= 1;
        shrink >>= 1;
      } while (shrink > 1 && width > 1);
      continue;
    }
    // emit: count occupied slots, reserve output rows once, then copy
    constexpr int kPerLane = kSlots / kThreads;
    uint32_t mineCount = 0;
#pragma unroll
    for (int k = 0; k < kPerLane; k++) mineCount += sKeys[threadIdx.x + k * kThreads] != kEmpty;
    if (mineCount) __hip_atomic_fetch_add(&sCount, mineCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const uint32_t total = sCount;
    if (threadIdx.x == 0 && total) sBase = atomicAdd(ws.outCount, total);
    __syncthreads();
    if (total && !(ws.debug & 16)) {
      // one output range per wavefront per sweep: consecutive lanes write consecutive rows
      const int lane = threadIdx.x & 63;
      uint32_t at[kPerLane], row[kPerLane];
      bool has[kPerLane];
#pragma unroll
      for (int k = 0; k < kPerLane; k++) {
        const uint64_t key = sKeys[threadIdx.x + k * kThreads];
        has[k] = key != kEmpty;
        row[k] = static_cast<uint32_t>(key);
        const uint64_t m = __ballot(has[k]);
        uint32_t waveBase = 0;
        if (lane == 0 && m)
          waveBase = __hip_atomic_fetch_add(&sClaims, static_cast<uint32_t>(__popcll(m)), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
        waveBase = __builtin_amdgcn_readfirstlane(waveBase);
        at[k] = sBase + waveBase +
                __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
      }
      if (ND4 > 0) {
        // all gathers of the lane's groups are issued before the first store
        const uint8_t *nullsIn = dimIn + static_cast<size_t>(4 * ND4) * inCapacity;
        uint8_t *nullsOut = dimOut + static_cast<size_t>(4 * ND4) * capacity;
        // (in two halves: 4 groups x ND4 gathers per lane in flight keeps the kernel inside 128 VGPRs)
        constexpr int kHalf = kPerLane / 2;
#pragma unroll
        for (int half = 0; half < 2; half++) {
          uint32_t dv[kHalf][ND4 > 0 ? ND4 : 1];
          uint32_t nv[kHalf][ND4 > 0 ? ND4 : 1];
#pragma unroll
          for (int kk = 0; kk < kHalf; kk++) {
            const int k = half * kHalf + kk;
            if (!has[k]) continue;
            if (FUSED && row[k] >= prevSize) {
              fused_eval_row<(ND4 > 0 ? ND4 : 1)>(*plan, row[k] - prevSize, dv[kk], nv[kk]);
              continue;
            }
#pragma unroll
            for (int d = 0; d < ND4; d++) {
              dv[kk][d] = *reinterpret_cast<const uint32_t *>(dimIn + static_cast<size_t>(4 * d) * inCapacity + 4ull * row[k]);
              nv[kk][d] = nullsIn[static_cast<size_t>(d) * inCapacity + row[k]];
            }
          }
#pragma unroll
          for (int kk = 0; kk < kHalf; kk++) {
            const int k = half * kHalf + kk;
            if (!has[k]) continue;
#pragma unroll
            for (int d = 0; d < ND4; d++) {
              *reinterpret_cast<uint32_t *>(dimOut + static_cast<size_t>(4 * d) * capacity + 4ull * at[k]) = dv[kk][d];
              nullsOut[static_cast<size_t>(d) * capacity + at[k]] = static_cast<uint8_t>(nv[kk][d]);
            }
            store_value_bits(outputValues, a, at[k], sVals[threadIdx.x + k * kThreads]);
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < kPerLane; k++) {
          if (!has[k]) continue;
          copy_dim_row(dimIn, inCapacity, dimOut, capacity, L, row[k], at[k]);
          store_value_bits(outputValues, a, at[k], sVals[threadIdx.x + k * kThreads]);
        }
      }
    }
    __syncthreads();
    lo = hi;
    if (total < static_cast<uint32_t>(kMergeLimit / 4) && width < (1ull << 32)) width <<= 1;
    if (lo + width > (1ull << 32)) width = (1ull << 32) - lo;
  }
}

template <int ND4>
__global__ __launch_bounds__(kThreads) void hr_merge_kernel(const uint8_t *__restrict__ dimIn, uint8_t *__restrict__ dimOut,
                                                            DimLayoutD L, size_t capacity,
                                                            uint8_t *__restrict__ outputValues, AggSpec a, Workspace ws) {
  merge_body<ND4, false>(dimIn, capacity, dimOut, L, capacity, outputValues, a, ws, nullptr, 0u);
}

template <int ND>
__global__ __launch_bounds__(kThreads) void hr_fused_merge_kernel(FusedPlanD plan, const uint8_t *__restrict__ prevDims,
                                                                  size_t prevCapacity, uint32_t prevSize,
                                                                  uint8_t *__restrict__ dimOut, size_t outCapacity,
                                                                  uint8_t *__restrict__ outputValues, AggSpec a, Workspace ws) {
  DimLayoutD L;  // unused by the all-4-byte emission
  L.numDims = ND;
  merge_body<ND, true>(prevDims, prevCapacity, dimOut, L, outCapacity, outputValues, a, ws, &plan, prevSize);
}

}  // namespace

bool hash_reduce_lds_supported(const AggSpec &a) {
  if (a.op == OP_AVG) return false;
  if (a.vtype == V_F32 && a.op != OP_SUM) return false;  // float min/max keep the reference's compare form
  return true;
}

int hash_reduce_lds(const DimensionVector &inputKeys, const uint8_t *inputValues, const DimensionVector &outputKeys,
                    uint8_t *outputValues, const AggSpec &a, int length, hipStream_t stream) {
  const DimLayoutD L = make_dim_layout(inputKeys.NumDimsPerDimWidth);
  int partBits = 0;
  while ((8192ll << partBits) < static_cast<int64_t>(length) && (1 << partBits) < kMaxPartitions) partBits++;
  const int numParts = 1 << partBits;
  Workspace ws;
  ws.partBits = partBits;
  const char *dbg = getenv("ARES_HR_DEBUG");
  ws.debug = dbg ? atoi(dbg) : 0;
  // region stride = cap * 16 B; keep it off large powers of two (cap = 17 mod 64 records) so that the
  // merge workgroups, which stream their regions in lockstep, do not camp on the same HBM channels
  ws.cap = ((2ull * (static_cast<uint64_t>(length) / numParts) + 2 * kSlots) | 63ull) + 18;
  const size_t headBytes = sizeof(uint32_t) * (numParts + 2);
  const size_t headPadded = (headBytes + 255) / 256 * 256;
  StreamBuffer buf(headPadded + sizeof(uint4) * ws.cap * numParts, stream);
  ws.cursors = buf.as<uint32_t>();
  ws.outCount = ws.cursors + numParts;
  ws.overflow = ws.outCount + 1;
  ws.records = reinterpret_cast<uint4 *>(buf.as<uint8_t>() + headPadded);
  hip_check(hipMemsetAsync(buf.get(), 0, headPadded, stream), "hipMemsetAsync");
  const int64_t tiles = (static_cast<int64_t>(length) + kTileRows - 1) / kTileRows;
  const int grid = static_cast<int>(tiles < 256 ? tiles : 256);
  bool all4 = L.numDims >= 1 && L.numDims <= 4;  // beyond 4 dims the double-buffered quads spill
  for (int d = 0; d < L.numDims; d++) all4 = all4 && L.width[d] == 4;
  if (all4) {
    const int64_t qtiles = (static_cast<int64_t>(length) + kQuadTile - 1) / kQuadTile;
    const int qgrid = static_cast<int>(qtiles < 256 ? qtiles : 256);
#define ARES_HR_CASE(ND)                                                                                             \
  case ND:                                                                                                           \
    if (a.width == 8)                                                                                                \
      ARES_LAUNCH("hr_partition4_kernel", (hr_partition4_kernel<ND, 8>), qgrid, kThreads, stream, inputKeys.DimValues, \
                  static_cast<size_t>(inputKeys.VectorCapacity), inputValues, a, length, ws);                       \
    else                                                                                                             \
      ARES_LAUNCH("hr_partition4_kernel", (hr_partition4_kernel<ND, 4>), qgrid, kThreads, stream, inputKeys.DimValues, \
                  static_cast<size_t>(inputKeys.VectorCapacity), inputValues, a, length, ws);                       \
    break;
    switch (L.numDims) {
      ARES_HR_CASE(1) ARES_HR_CASE(2) ARES_HR_CASE(3) ARES_HR_CASE(4)
    }
#undef ARES_HR_CASE
  } else {
    ARES_LAUNCH("hr_partition_kernel", hr_partition_kernel, grid, kThreads, stream, inputKeys.DimValues, L,
                static_cast<size_t>(inputKeys.VectorCapacity), inputValues, a, length, ws);
  }
#define ARES_HR_MERGE(ND)                                                                                          \
  ARES_LAUNCH("hr_merge_kernel", hr_merge_kernel<ND>, numParts, kThreads, stream, inputKeys.DimValues, outputKeys.DimValues, L, \
              static_cast<size_t>(inputKeys.VectorCapacity), outputValues, a, ws)
  switch (all4 ? L.numDims : 0) {
    case 1: ARES_HR_MERGE(1); break;
    case 2: ARES_HR_MERGE(2); break;
    case 3: ARES_HR_MERGE(3); break;
    case 4: ARES_HR_MERGE(4); break;
    default: ARES_HR_MERGE(0); break;
  }
#undef ARES_HR_MERGE
  uint32_t result[2] = {0, 0};  // {groups, overflow}
  read_back_u32(ws.outCount, result, 2, stream);
  if (result[1]) return -1;
  return static_cast<int>(result[0]);
}


// The fused pipeline for an already built plan (shared by the extension entry point and by the
// in-ABI fusion of pending transforms into HashReduce, transform.hip).  Returns the number of groups
// or -1 when a partition region overflowed.
int fused_hash_reduce_run(const FusedPlanD &plan, int batchRows, const DimensionVector &prevKeys, const uint8_t *prevValues,
                          int prevSize, const DimensionVector &outKeys, uint8_t *outValues, const AggSpec &a,
                          hipStream_t stream) {
  int nd = 0;
  for (int k = 0; k < NUM_DIM_WIDTH; k++) nd += outKeys.NumDimsPerDimWidth[k];
  const int mw = plan.measureWidth;
  const int64_t length = static_cast<int64_t>(batchRows) + prevSize;
  if (length == 0) return 0;
  int partBits = 0;
  while ((8192ll << partBits) < length && (1 << partBits) < kMaxPartitions) partBits++;
  const int numParts = 1 << partBits;
  Workspace ws;
  ws.partBits = partBits;
  {
    const char *dbg = getenv("ARES_HR_DEBUG");  // timing experiments only
    ws.debug = dbg ? atoi(dbg) : 0;
  }
  ws.cap = ((2ull * (static_cast<uint64_t>(length) / numParts) + 2 * kSlots) | 63ull) + 18;
  const size_t headBytes = sizeof(uint32_t) * (numParts + 2);
  const size_t headPadded = (headBytes + 255) / 256 * 256;
  StreamBuffer buf(headPadded + sizeof(uint4) * ws.cap * numParts, stream);
  ws.cursors = buf.as<uint32_t>();
  ws.outCount = ws.cursors + numParts;
  ws.overflow = ws.outCount + 1;
  ws.records = reinterpret_cast<uint4 *>(buf.as<uint8_t>() + headPadded);
  hip_check(hipMemsetAsync(buf.get(), 0, headPadded, stream), "hipMemsetAsync");

  auto grid_for = [](int64_t rows) {
    const int64_t tiles = (rows + kQuadTile - 1) / kQuadTile;
    return static_cast<int>(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
  };
#define ARES_FUSED_CASE(ND)                                                                                            \
  case ND:                                                                                                             \
    if (prevSize > 0) {                                                                                                \
      if (mw == 8)                                                                                                     \
        ARES_LAUNCH("hr_partition4_kernel", (hr_partition4_kernel<ND, 8>), grid_for(prevSize), kThreads, stream,        \
                    prevKeys.DimValues, static_cast<size_t>(prevKeys.VectorCapacity), prevValues, a, prevSize, ws);     \
      else                                                                                                             \
        ARES_LAUNCH("hr_partition4_kernel", (hr_partition4_kernel<ND, 4>), grid_for(prevSize), kThreads, stream,        \
                    prevKeys.DimValues, static_cast<size_t>(prevKeys.VectorCapacity), prevValues, a, prevSize, ws);     \
    }                                                                                                                  \
    if (batchRows > 0)                                                                                                 \
      ARES_LAUNCH("hr_fused_scan_kernel", hr_fused_scan_kernel<ND>, grid_for(batchRows), kThreads, stream, plan,       \
                  static_cast<uint32_t>(prevSize), a, batchRows, ws);                                                  \
    ARES_LAUNCH("hr_fused_merge_kernel", hr_fused_merge_kernel<ND>, numParts, kThreads, stream, plan, prevKeys.DimValues, \
                static_cast<size_t>(prevKeys.VectorCapacity), static_cast<uint32_t>(prevSize), outKeys.DimValues,      \
                static_cast<size_t>(outKeys.VectorCapacity), outValues, a, ws);                                        \
    break;
  switch (nd) {
    ARES_FUSED_CASE(1) ARES_FUSED_CASE(2) ARES_FUSED_CASE(3) ARES_FUSED_CASE(4)
  }
#undef ARES_FUSED_CASE
  uint32_t result[2] = {0, 0};
  read_back_u32(ws.outCount, result, 2, stream);
  if (result[1]) return -1;
  return static_cast<int>(result[0]);
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
      if (plan.cols[c].vals == f.vals && plan.cols[c].nulls == f.nulls && plan.cols[c].bitOff == f.bitOff) return c;
  if (plan.numCols >= maxCols) throw NotFusable("too many distinct columns");
  FusedColumn &col = plan.cols[plan.numCols];
  col.vals = f.vals;
  col.nulls = f.nulls;
  col.bitOff = f.bitOff;
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

int fused_filter_hash_reduce(const AresFusedQuery &q, int batchRows, const DimensionVector &prevKeys, uint8_t *prevValues,
                             int prevSize, const DimensionVector &outKeys, uint8_t *outValues, hipStream_t stream) {
  if (q.numDims < 1 || q.numDims > kFusedDims) throw NotFusable("1..4 dimensions");
  if (q.numFilters < 0 || q.numFilters > kFusedFilters) throw NotFusable("at most 4 filters");
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

  const int groups = fused_hash_reduce_run(plan, batchRows, prevKeys, prevValues, prevSize, outKeys, outValues, a, stream);
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
  resHandle.res = ares::int_result(ares::fused_filter_hash_reduce(*query, batchRows, prevKeys, prevValues, prevSize, outKeys,
                                                                  outValues, reinterpret_cast<hipStream_t>(cudaStream)));
  ARES_ABI_END("AresFusedFilterHashReduce")
}
