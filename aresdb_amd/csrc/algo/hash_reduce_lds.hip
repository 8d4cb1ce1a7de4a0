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
#include "dim_layout.hpp"
#include "hash_reduce_lds.hpp"

namespace ares {

namespace {

constexpr int kThreads = 1024;
constexpr int kSlots = 8192;              // LDS table slots (16 bytes each: 128 KiB)
constexpr int kSlotMask = kSlots - 1;
constexpr int kRowsPerLane = 2;           // rows inserted per lane between occupancy checks
constexpr int kTileRows = kThreads * kRowsPerLane;
constexpr int kFlushAt = kSlots * 3 / 4 - kTileRows;  // flush when more entries than this are held
constexpr int kMergeLimit = kSlots * 7 / 8;           // claim attempts per merge round
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
constexpr int kStage = kQuadTile;                         // records staged per DIRECT tile

template <int ND, int VW>
__global__ __launch_bounds__(kThreads) void hr_partition4_kernel(const uint8_t *dimValues, size_t capacity,
                                                                 const uint8_t *inputValues, AggSpec a, int length,
                                                                 Workspace ws) {
  // TABLE mode: sKeys / sVals are the hash table.  DIRECT mode: the same 128 KiB stage 4096..8192
  // sorted records (sKeys[k] = {row, hash}, sVals[k] = value).
  __shared__ uint64_t sKeys[kSlots];
  __shared__ uint64_t sVals[kSlots];
  __shared__ uint32_t sPartCount[kMaxPartitions];
  __shared__ uint32_t sPartBase[kMaxPartitions];   // global base of the partition's run
  __shared__ uint32_t sPartLocal[kMaxPartitions];  // DIRECT: first staged index of the partition
  __shared__ uint32_t sWaveSum[kThreads / 64];
  __shared__ uint32_t sClaims;
  const int numParts = 1 << ws.partBits;
  const int pb = ws.partBits;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  clear_table(sKeys, sVals, a.identity);
  for (int p = threadIdx.x; p < numParts; p += kThreads) sPartCount[p] = 0;
  if (threadIdx.x == 0) sClaims = 0;
  __syncthreads();

  const int64_t numTiles = (static_cast<int64_t>(length) + kQuadTile - 1) / kQuadTile;
  bool direct = false;
  uint32_t rowsSinceFlush = 0;
  QuadRows<ND> bufA, bufB;
  int64_t tile = blockIdx.x;
  if (tile < numTiles) load_quad<ND, VW>(bufA, dimValues, capacity, inputValues, tile * kQuadTile + 4 * threadIdx.x, length);

  // processes one tile held in registers; returns false when the workgroup is done
  auto step = [&](QuadRows<ND> &q, int64_t t) -> bool {
    const bool more = t < numTiles;
    const int64_t i0 = t * kQuadTile + 4 * static_cast<int64_t>(threadIdx.x);
    if (more && !direct) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (i0 + j < length) {
          const uint32_t h = hash_quad_row<ND>(q, j);
          if (ws.debug & 1) {
            if (h == 0x12345678u && q.val[j] == 0x9abcdef012345678ull) sClaims = 1;
          } else {
            const int slot = lds_find_or_claim(sKeys, h, static_cast<uint32_t>(i0 + j), &sClaims);
            lds_aggregate(sVals + slot, q.val[j], a);
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
    // ---- DIRECT: rows -> records, counting-sorted by partition in LDS, coalesced write-back ----
    if (!more) return false;
    uint32_t h[4], rank[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      h[j] = hash_quad_row<ND>(q, j);
      rank[j] = 0;
      if (i0 + j < length) {
        const uint32_t p = pb ? h[j] >> (32 - pb) : 0u;
        rank[j] = __hip_atomic_fetch_add(&sPartCount[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
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
        if (c) sPartBase[threadIdx.x] = atomicAdd(ws.cursors + threadIdx.x, c);
        sPartCount[threadIdx.x] = 0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (i0 + j < length) {
        const uint32_t p = pb ? h[j] >> (32 - pb) : 0u;
        const uint32_t at = sPartLocal[p] + rank[j];
        sKeys[at] = (static_cast<uint64_t>(h[j]) << 32) | static_cast<uint32_t>(i0 + j);
        sVals[at] = q.val[j];
      }
    }
    __syncthreads();
    {
      const int64_t remaining = static_cast<int64_t>(length) - t * kQuadTile;
      const uint32_t staged = remaining < kQuadTile ? static_cast<uint32_t>(remaining) : static_cast<uint32_t>(kQuadTile);
      for (uint32_t k = threadIdx.x; k < staged; k += kThreads) {
        const uint64_t key = sKeys[k];
        const uint64_t v = sVals[k];
        const uint32_t p = pb ? static_cast<uint32_t>(key >> (64 - pb)) : 0u;
        const uint64_t at = static_cast<uint64_t>(sPartBase[p]) + (k - sPartLocal[p]);
        if (ws.debug & 2) {
        } else if (at < ws.cap) {
          ws.records[static_cast<uint64_t>(p) * ws.cap + at] =
              make_uint4(static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32), static_cast<uint32_t>(v),
                         static_cast<uint32_t>(v >> 32));
        } else {
          *ws.overflow = 1u;
        }
      }
    }
    __syncthreads();  // the stage is reused by the next tile
    return true;
  };

  for (;;) {
    const int64_t tileB = tile + gridDim.x;
    if (tileB < numTiles) load_quad<ND, VW>(bufB, dimValues, capacity, inputValues, tileB * kQuadTile + 4 * threadIdx.x, length);
    if (!step(bufA, tile)) break;
    const int64_t tileA = tileB + gridDim.x;
    if (tileA < numTiles) load_quad<ND, VW>(bufA, dimValues, capacity, inputValues, tileA * kQuadTile + 4 * threadIdx.x, length);
    if (!step(bufB, tileB)) break;
    tile = tileA;
  }
}

// ---- kernel 2: per-partition merge in LDS, emit groups ---------------------------------------------
__global__ __launch_bounds__(kThreads) void hr_merge_kernel(const uint8_t *dimIn, uint8_t *dimOut, DimLayoutD L,
                                                            size_t capacity, uint8_t *outputValues, AggSpec a,
                                                            Workspace ws) {
  __shared__ uint64_t sKeys[kSlots];
  __shared__ uint64_t sVals[kSlots];
  __shared__ uint32_t sAttempts, sOverflow, sCount, sBase, sClaims;
  const int p = blockIdx.x;
  const uint32_t cursor = ws.cursors[p];
  const uint64_t n = cursor < ws.cap ? cursor : ws.cap;
  if (n == 0) return;
  const uint4 *rec = ws.records + static_cast<uint64_t>(p) * ws.cap;
  // sub-range of the hash bits below the partition bits, left-aligned to 32 bits
  const int pb = ws.partBits;
  uint64_t lo = 0, width = 1ull << 32;
  if (n > 16ull * kMergeLimit) {
    uint64_t parts = 1;
    while (parts * 16ull * kMergeLimit < n && width > 1) { parts <<= 1; width >>= 1; }
  }
  while (lo < (1ull << 32)) {
    clear_table(sKeys, sVals, a.identity);
    if (threadIdx.x == 0) { sAttempts = 0; sOverflow = 0; sCount = 0; sClaims = 0; }
    __syncthreads();
    const uint64_t hi = lo + width;
    // kMergeBatch independent 16-byte loads per lane are in flight before any of them is used
    constexpr int kMergeBatch = 8;
    for (uint64_t base = 0; base < n; base += static_cast<uint64_t>(kMergeBatch) * kThreads) {
      uint4 r[kMergeBatch];
#pragma unroll
      for (int k = 0; k < kMergeBatch; k++) {
        const uint64_t i = base + static_cast<uint64_t>(k) * kThreads + threadIdx.x;
        r[k] = i < n ? rec[i] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < kMergeBatch; k++) {
        const uint64_t i = base + static_cast<uint64_t>(k) * kThreads + threadIdx.x;
        const uint32_t h = r[k].y;
        const uint64_t u = static_cast<uint64_t>(pb ? (h << pb) : h);
        if (i >= n || u < lo || u >= hi) continue;
        if (ws.debug & 8) {  // experiment: loads only
          if (h == 0x12345678u && r[k].z == 0x9abcdefu) sOverflow = 1u;
          continue;
        }
        // find the group; claim a slot only while the round's attempt budget lasts
        const uint64_t mine = (static_cast<uint64_t>(h) << 32) | r[k].x;
        int slot = static_cast<int>(h) & kSlotMask;
        bool placed = false;
        for (;;) {
          uint64_t cur = sKeys[slot];
          if (cur == kEmpty) {
            if (__hip_atomic_fetch_add(&sAttempts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >=
                static_cast<uint32_t>(kMergeLimit)) {
              sOverflow = 1u;
              break;
            }
            unsigned long long expected = kEmpty;
            if (__hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned long long *>(sKeys + slot), &expected,
                                                     static_cast<unsigned long long>(mine), __ATOMIC_RELAXED,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
              placed = true;
              break;
            }
            cur = expected;
          }
          if (static_cast<uint32_t>(cur >> 32) == h) {
            if (mine < cur)
              __hip_atomic_fetch_min(reinterpret_cast<unsigned long long *>(sKeys + slot),
                                     static_cast<unsigned long long>(mine), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
            placed = true;
            break;
          }
          slot = (slot + 1) & kSlotMask;
        }
        if (placed) lds_aggregate(sVals + slot, (static_cast<uint64_t>(r[k].w) << 32) | r[k].z, a);
      }
    }
    __syncthreads();
    const bool overflowed = sOverflow != 0;
    __syncthreads();  // everyone has read the flag before the next round resets it
    if (overflowed && width > 1) {  // too many groups in this sub-range: nothing is emitted, retry with half of it
      width >>= 1;                  // (a single hash value cannot overflow the table)
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
#pragma unroll
      for (int k = 0; k < kPerLane; k++) {
        const int s = threadIdx.x + k * kThreads;
        const uint64_t key = sKeys[s];
        const uint64_t m = __ballot(key != kEmpty);
        if (m == 0) continue;
        uint32_t waveBase = 0;
        if (lane == 0)
          waveBase = __hip_atomic_fetch_add(&sClaims, static_cast<uint32_t>(__popcll(m)), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
        waveBase = __builtin_amdgcn_readfirstlane(waveBase);
        if (key == kEmpty) continue;
        const uint32_t at = sBase + waveBase +
                            __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                      __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
        copy_dim_row(dimIn, capacity, dimOut, capacity, L, static_cast<uint32_t>(key), at);
        store_value_bits(outputValues, a, at, sVals[s]);
      }
    }
    __syncthreads();
    lo = hi;
    if (total < static_cast<uint32_t>(kMergeLimit / 4) && width < (1ull << 32)) width <<= 1;
    if (lo + width > (1ull << 32)) width = (1ull << 32) - lo;
  }
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
  ARES_LAUNCH("hr_merge_kernel", hr_merge_kernel, numParts, kThreads, stream, inputKeys.DimValues, outputKeys.DimValues, L,
              static_cast<size_t>(inputKeys.VectorCapacity), outputValues, a, ws);
  uint32_t result[2] = {0, 0};  // {groups, overflow}
  read_back_u32(ws.outCount, result, 2, stream);
  if (result[1]) return -1;
  return static_cast<int>(result[0]);
}

}  // namespace ares
