// HashReduce, partitioned: device code (kernels' bodies) of hash_reduce_lds.hip.
//
// Why (tools/ubench_atomics.hip, MI355X): read-modify-writes on global memory run at ~24 G/s for
// the whole chip whatever the table size or scope, LDS atomics at 1.2-1.7 T/s.  Group-by therefore
// aggregates in LDS and exchanges hash-partitioned records through HBM:
//
//   1. partition (one 1024-lane workgroup per CU): rows are hashed (murmur3_x86_32 of the packed
//      dimension row, as the reference: query/hash_reduction.cu:216-243) and
//        TABLE mode  — aggregated into an 8192-slot LDS hash table (key = hash32 << 32 | row,
//                      value 8 bytes) that is flushed as 16-byte records {row, hash, value},
//                      counting-sorted by the top bits of the hash, into that partition's shared
//                      region A (one cursor reservation per partition per flush).  Low-cardinality
//                      input never flushes until the end: one record per group per workgroup;
//        DIRECT mode — when a flush shows (almost) no duplicates, aggregation is pure overhead:
//                      every row becomes a 12-byte record {row, hash, 4-byte measure} appended to
//                      the workgroup's PRIVATE stream of its partition (region B).  The stream
//                      cursors live in LDS for the workgroup's lifetime, so there is no global
//                      atomic, no barrier and no LDS staging in the loop; the partial cache lines
//                      of a stream are completed by the same workgroup a few hundred rows later
//                      and combine in the XCD's L2.
//   2. merge: one workgroup per partition inserts the partition's previous groups (read straight
//      from the partition-grouped previous result, see GroupedRanges), streams the partition's
//      records (region A, then the private runs of region B, one run per wavefront at a time) through
//      an LDS table and emits final groups (dimension row of the representative + value).  A
//      partition with more groups than the table holds is processed in rounds over disjoint hash
//      sub-ranges chosen between rounds only, so membership never depends on timing.
//
// Group identity (the 32-bit hash, as the reference), representative row (lowest row index, via
// 64-bit atomic min on the key) and the unspecified output order are those of hash_reduce.hip / the
// reference.  A region that overflows (adversarial hash skew) makes the host fall back to the
// global-table path.
#pragma once

#include <hip/hip_runtime.h>

#include "aggregate.hpp"
#include "device_model.hpp"
#include "dim_layout.hpp"
#include "fast_eval.hpp"
#include "hash_reduce_lds.hpp"

namespace ares {
namespace hr {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kSlots = 8192;              // LDS table slots (16 bytes each: 128 KiB)
constexpr int kSlotMask = kSlots - 1;
constexpr int kRowsPerLane = 2;           // generic kernel: rows inserted per lane between occupancy checks
constexpr int kTileRows = kThreads * kRowsPerLane;
constexpr int kFlushAt = kSlots * 3 / 4 - kTileRows;  // flush when more entries than this are held
constexpr int kMergeLimit = kSlots - kThreads - 128;   // groups per merge round (every lane may claim one more)
constexpr int kMaxPartitions = 512;
constexpr int kMaxStreams = 256;          // workgroups of a partition launch = private streams per partition
constexpr int kMaxRanges = 8;             // output ranges a partition may emit and still count as "grouped"
constexpr int kRangeWords = 1 + 2 * kMaxRanges;  // per partition: n, then (start, count) pairs
constexpr uint64_t kEmpty = ~0ull;

// How the 4 bytes a DIRECT record carries become the aggregate's value.
//   mode 0: they are the value (4-byte aggregates);
//   mode 1: MeasureProxy widening (query/iterator.hpp:616-647) of a kind-rk value to an 8-byte sum.
struct Widen {
  int mode, rk, dtype;
};
__device__ __forceinline__ uint64_t widen_value(const Widen &w, uint32_t raw) {
  if (w.mode == 0) return raw;
  if (w.dtype == Float64) {
    DVal r;
    r.bits = raw;
    r.ok = 1;
    return static_cast<uint64_t>(__double_as_longlong(to_double32(r, w.rk)));
  }
  return static_cast<uint64_t>(w.rk == K_F32 ? static_cast<int64_t>(bits_f(raw))
                               : w.rk == K_I32 ? static_cast<int64_t>(static_cast<int32_t>(raw))
                                               : static_cast<int64_t>(raw));
}

struct Workspace {
  // region A: 16-byte records {row, hash, value64}, one region of capA records per partition
  uint4 *recA;
  uint32_t *cursorsA;    // records appended per partition
  uint64_t capA;
  // region B: private streams [workgroup][partition][capB] of rwB-word records {row, hash, value...}
  // (workgroup-major: the 512 streams one workgroup appends to lie within a few MB — a few TLB
  // entries — while the merge reads whole runs, where a TLB miss per run does not matter)
  uint32_t *recB;
  uint32_t *countsB;     // [workgroup][partition]
  uint32_t capB;
  int streams;           // workgroups of the partition launch that wrote B (0 = none)
  // B records written in whole 128-byte lines by the run-time compiled scans (hr_rtc.hip):
  //   8  — 16-byte records {row, hash, 4-byte carried measure (or the low word of the value), 0 (or the high word)};
  //        a stream ends with null records (row = ~0) that pad its last line;
  //   14 — compact lines (kCompactLineRecords): two 64-byte halves of {header u64, 7 x record u64}; a record is
  //        {lo: carried measure, hi: (hash << partBits) | (row within the workgroup's chunk >> 9)} — the partition
  //        bits of the hash are implied by the stream — and the header holds the low 9 row bits of its 7 records
  //        (9 bits each).  Workgroup g scans rows [g * chunkRows, (g + 1) * chunkRows) of the batch, so
  //        row = rowBase + g * chunkRows + rowInChunk.  capB counts LINES, countsB records (no padding marker).
  int lineRecords;
  uint32_t chunkRows, rowBase;
  uint32_t *outCount;    // groups emitted; [1] = a region overflowed; [2] = the grouped previous result is stale
  int partBits;
  Widen widen;
  // previous result, grouped by partition: kRangeWords words per partition (null = none)
  const uint32_t *prevRanges;
  uint32_t *outRanges;   // where the merge records the ranges it emits (null = not wanted)
};
__device__ __forceinline__ uint32_t *ws_overflow(const Workspace &ws) { return ws.outCount + 1; }
__device__ __forceinline__ uint32_t *ws_stale(const Workspace &ws) { return ws.outCount + 2; }

struct __attribute__((packed, aligned(4))) Rec3 { uint32_t row, hash, val; };

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

// Flush of the LDS table into region A: counting sort of the entries by partition, one cursor
// reservation per partition.  All lanes of the workgroup call it; the table is empty afterwards.
__device__ __forceinline__ void flush_table(uint64_t *sKeys, uint64_t *sVals, uint32_t *sPartCount, uint32_t *sPartBase,
                                            uint32_t *sClaims, const AggSpec &a, const Workspace &ws) {
  constexpr int kPerLane = kSlots / kThreads;
  const int pb = ws.partBits;
  const int numParts = 1 << pb;
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
      sPartBase[p] = atomicAdd(ws.cursorsA + p, c);
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
      if (at < ws.capA) {
        ws.recA[static_cast<uint64_t>(p) * ws.capA + at] =
            make_uint4(static_cast<uint32_t>(key), static_cast<uint32_t>(key >> 32), static_cast<uint32_t>(v),
                       static_cast<uint32_t>(v >> 32));
      } else {
        *ws_overflow(ws) = 1u;
      }
      sKeys[s] = kEmpty;
      sVals[s] = a.identity;
    }
  }
  if (threadIdx.x == 0) *sClaims = 0;
  __syncthreads();
}

// ---- kernel 1, generic dimension layout: aggregate in LDS, spill hash-partitioned records ---------
__device__ __forceinline__ void partition_generic_body(const uint8_t *dimValues, const DimLayoutD &L, size_t capacity,
                                                       const uint8_t *inputValues, const AggSpec &a, int length,
                                                       const Workspace &ws) {
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
  for (int64_t tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
#pragma unroll
    for (int k = 0; k < kRowsPerLane; k++) {
      const int64_t i = tile * kTileRows + static_cast<int64_t>(k) * kThreads + threadIdx.x;
      if (i < length) {
        const uint32_t row = static_cast<uint32_t>(i);
        Murmur32Stream ms(0);
        hash_dim_row(ms, dimValues, L, capacity, row);
        const uint32_t h = ms.finish();
        const uint64_t v = load_value_bits(inputValues, a, row);
        const int slot = lds_find_or_claim(sKeys, h, row, &sClaims);
        lds_aggregate(sVals + slot, v, a);
      }
    }
    __syncthreads();
    if (sClaims > static_cast<uint32_t>(kFlushAt))  // uniform: read after the barrier
      flush_table(sKeys, sVals, sPartCount, sPartBase, &sClaims, a, ws);
  }
  __syncthreads();
  if (sClaims > 0) flush_table(sKeys, sVals, sPartCount, sPartBase, &sClaims, a, ws);
}

// ---- kernel 1, hot layout: every dimension 4 bytes wide --------------------------------------------
// Each lane owns 4 consecutive rows per tile: the ND value vectors are read 16 bytes per lane, the
// validity bytes 4 per lane, the measures 16/32 bytes per lane (byte-aligned vector accesses are
// native on gfx950), and the NEXT tile's loads are issued before the current tile is hashed and
// inserted, so HBM latency hides behind the LDS work.
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
                                          const uint8_t *inputValues, int64_t i0, int64_t end) {
  const uint8_t *nulls = dimValues + static_cast<size_t>(4 * ND) * capacity;
  if (i0 + 3 < end) {
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
        if (i0 + j < end) {
          q.dim[d][j] = *reinterpret_cast<const uint32_t *>(dimValues + static_cast<size_t>(4 * d) * capacity + 4 * (i0 + j));
          q.nul[d] |= static_cast<uint32_t>(nulls[static_cast<size_t>(d) * capacity + i0 + j]) << (8 * j);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      q.val[j] = 0;
      if (i0 + j < end)
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

// Row source "dimension vector": rows [rowStart, rowStart + length) of the ABI's HashReduce input
// (values per dimension, validity bytes, measures), already projected by the transform calls.
template <int ND_, int VW>
struct DimVectorSource {
  static constexpr int ND = ND_;
  static constexpr int RW = VW == 8 ? 4 : 3;  // words per DIRECT record
  using Raw = QuadRows<ND_>;
  const uint8_t *dimValues;
  size_t capacity;
  const uint8_t *inputValues;
  uint32_t rowStart;
  __device__ __forceinline__ uint32_t row_id(int64_t i) const { return rowStart + static_cast<uint32_t>(i); }
  __device__ __forceinline__ void prepare() {}
  __device__ __forceinline__ void load(Raw &r, int64_t i0, int length) const {
    load_quad<ND_, VW>(r, dimValues, capacity, inputValues, static_cast<int64_t>(rowStart) + i0,
                       static_cast<int64_t>(rowStart) + length);
  }
  // hash + carried measure bits of the quad's four rows; returns which of them take part (4-bit mask)
  __device__ __forceinline__ uint32_t rows(const Raw &r, int64_t i0, int length, uint32_t (&h)[4], uint64_t (&c)[4]) const {
    uint32_t alive = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      alive |= (i0 + j < length ? 1u : 0u) << j;
      h[j] = hash_quad_row<ND_>(r, j);
      c[j] = r.val[j];
    }
    return alive;
  }
  __device__ __forceinline__ uint64_t final_bits(const Workspace &, uint64_t c) const { return c; }
};

// ---- row source "fused scan": filter + projection evaluated from the source columns -------------
// (in-ABI second-stage fusion and the extension entry point AresFusedFilterHashReduce,
// include/ares_extensions.h).  The batch's columns are read once, 16 bytes per lane; the conjunction
// of comparison filters decides which rows take part; dimensions and the measure are evaluated with
// the very functions the transform kernels use (fast_eval.hpp), so the (value, validity) pairs that
// are hashed are bit-identical to what the UnaryTransform / BinaryTransform calls would have stored
// in the dimension vector.
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

// what a record carries for an evaluated measure value (MeasureProxy, query/iterator.hpp:616-647, no
// run lengths): 4-byte aggregates carry the stored value (null -> the aggregate's identity), 8-byte
// sums carry the kind-rk bits (null -> 0, which widens to the sum's identity 0)
__device__ __forceinline__ uint32_t fused_carry(const FusedPlanD &p, DVal r) {
  const int rk = p.measure.f.rk;
  if (p.measureWidth == 8) return r.ok ? r.bits : 0u;
  if (!r.ok) return static_cast<uint32_t>(p.identity);
  return cvt32(r, rk, p.measureDtype == Int32 ? K_I32 : p.measureDtype == Uint32 ? K_U32 : K_F32).bits;
}

template <int ND_>
struct FusedSource {
  static constexpr int ND = ND_;
  static constexpr int RW = 3;
  static constexpr int NC = ND_ + 2;  // distinct columns a plan of ND dimensions may touch
  struct Raw {
    uint32_t v[NC][4];
    uint32_t win[NC];  // 16-bit validity window starting at the byte of the quad's first row
  };
  const FusedPlanD &plan;
  uint32_t rowBase;
  FusedConst fc[kFusedFilters], dc[ND_], mc;

  __device__ __forceinline__ FusedSource(const FusedPlanD &p, uint32_t base) : plan(p), rowBase(base) {}
  __device__ __forceinline__ uint32_t row_id(int64_t i) const { return rowBase + static_cast<uint32_t>(i); }
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
  __device__ __forceinline__ uint32_t rows(const Raw &r, int64_t i0, int length, uint32_t (&h)[4], uint64_t (&cv)[4]) const {
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
      cv[j] = fused_carry(plan, x);
    }
    return alive;
  }
  __device__ __forceinline__ uint64_t final_bits(const Workspace &ws, uint64_t c) const {
    return widen_value(ws.widen, static_cast<uint32_t>(c));
  }
};

// The partition kernel body, shared by the ABI path (DimVectorSource) and the fused scan
// (FusedSource: filter + projection evaluated on the fly from the source columns).
// allowDirect = 0 keeps the workgroup in TABLE mode whatever the data does (region A only).
template <typename Source>
__device__ __forceinline__ void partition_body(Source &src, const AggSpec &a, int length, const Workspace &ws,
                                               int allowDirect) {
  __shared__ uint64_t sKeys[kSlots];
  __shared__ uint64_t sVals[kSlots];
  // TABLE flush: records per partition; DIRECT: the cursors of the workgroup's private streams
  __shared__ uint32_t sPartCount[kMaxPartitions];
  __shared__ uint32_t sPartBase[kMaxPartitions];
  __shared__ uint32_t sClaims;
  constexpr int RW = Source::RW;
  const int pb = ws.partBits;
  const int numParts = 1 << pb;
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

  // ---- TABLE mode: one register buffer is enough to overlap HBM latency with the LDS work — the
  // tile's rows are reduced to (hash, measure) pairs first, then the NEXT tile's loads are issued
  // into the same registers before the current tile goes through the table.
  while (tile < numTiles && !direct) {
    const int64_t i0 = tile * kQuadTile + 4 * static_cast<int64_t>(threadIdx.x);
    uint32_t h[4];
    uint64_t c[4];
    const uint32_t alive = src.rows(buf, i0, length, h, c);
    const int64_t next = tile + gridDim.x;
    if (next < numTiles) src.load(buf, next * kQuadTile + 4 * threadIdx.x, length);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if ((alive >> j) & 1u) {
        const int slot = lds_find_or_claim(sKeys, h[j], src.row_id(i0 + j), &sClaims);
        lds_aggregate(sVals + slot, src.final_bits(ws, c[j]), a);
      }
    }
    rowsSinceFlush += kQuadTile;
    tile = next;
    __syncthreads();
    const uint32_t held = sClaims;  // uniform: read after the barrier
    if (held > static_cast<uint32_t>(kQuadFlushAt)) {
      flush_table(sKeys, sVals, sPartCount, sPartBase, &sClaims, a, ws);
      // (almost) every row became its own entry: stop aggregating, just partition
      if (allowDirect && static_cast<uint64_t>(held) * 5 > static_cast<uint64_t>(rowsSinceFlush) * 4) direct = true;
      rowsSinceFlush = 0;
    }
  }
  if (!direct) {
    __syncthreads();
    if (sClaims > 0) flush_table(sKeys, sVals, sPartCount, sPartBase, &sClaims, a, ws);
  }

  // ---- DIRECT mode: rows -> records in the workgroup's private streams.  sPartCount[p] (zero after
  // every flush) is the stream cursor of partition p; no barrier, no global atomic: the wavefronts
  // run free and hide each other's latencies.
  const uint32_t capB = ws.capB;
  uint32_t *myB = ws.recB + static_cast<uint64_t>(blockIdx.x) * numParts * capB * RW;  // this workgroup's streams
  while (tile < numTiles) {
    const int64_t i0 = tile * kQuadTile + 4 * static_cast<int64_t>(threadIdx.x);
    uint32_t h[4];
    uint64_t c[4];
    const uint32_t alive = src.rows(buf, i0, length, h, c);
    const int64_t next = tile + gridDim.x;
    if (next < numTiles) src.load(buf, next * kQuadTile + 4 * threadIdx.x, length);
    uint32_t rank[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      rank[j] = capB;
      if ((alive >> j) & 1u) {
        const uint32_t p = pb ? h[j] >> (32 - pb) : 0u;
        rank[j] = __hip_atomic_fetch_add(&sPartCount[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (((alive >> j) & 1u) && rank[j] >= capB) {
        // The private stream is full: rows that arrive clustered by partition (previous results fed back
        // without their grouping being known) overrun it by design.  Such records go to the shared
        // region A instead — one global cursor reservation each, the slow but unbounded path.
        const uint32_t p = pb ? h[j] >> (32 - pb) : 0u;
        const uint64_t at = atomicAdd(ws.cursorsA + p, 1u);
        const uint64_t v = src.final_bits(ws, c[j]);
        if (at < ws.capA)
          ws.recA[static_cast<uint64_t>(p) * ws.capA + at] =
              make_uint4(src.row_id(i0 + j), h[j], static_cast<uint32_t>(v), static_cast<uint32_t>(v >> 32));
        else
          *ws_overflow(ws) = 1u;
      }
      if (rank[j] < capB) {
        const uint32_t p = pb ? h[j] >> (32 - pb) : 0u;
        uint32_t *dst = myB + (p * capB + rank[j]) * RW;
        if constexpr (RW == 4) {
          *reinterpret_cast<uint4 *>(dst) = make_uint4(src.row_id(i0 + j), h[j], static_cast<uint32_t>(c[j]),
                                                       static_cast<uint32_t>(c[j] >> 32));
        } else {
          Rec3 rec;
          rec.row = src.row_id(i0 + j);
          rec.hash = h[j];
          rec.val = static_cast<uint32_t>(c[j]);
          *reinterpret_cast<Rec3 *>(dst) = rec;
        }
      }
    }
    tile = next;
  }

  // ---- epilogue: publish the lengths of this workgroup's runs
  if (ws.streams > 0) {
    __syncthreads();
    for (int p = threadIdx.x; p < numParts; p += kThreads) {
      uint32_t cnt = direct ? sPartCount[p] : 0u;
      if (cnt > capB) cnt = capB;  // the rest went to region A
      ws.countsB[static_cast<uint64_t>(blockIdx.x) * numParts + p] = cnt;
    }
  }
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
// One record into the round's table; claims a slot only while the round's budget lasts.
__device__ __forceinline__ void merge_record(uint64_t *sKeys, uint64_t *sVals, uint32_t *sClaimed, uint32_t *sOverflow,
                                             uint32_t row, uint32_t h, uint64_t value, const AggSpec &a) {
  // a round that has run out of slots is discarded as a whole: stop filling the table (every lane
  // claims at most one more slot after the flag is up, so the table can never fill completely and
  // the probe loop below always terminates)
  if (__hip_atomic_load(sOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
  const uint64_t mine = (static_cast<uint64_t>(h) << 32) | row;
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
  lds_aggregate(sVals + slot, value, a);
}

constexpr int kMergeBatch = 4;                      // records per lane per pipeline stage
constexpr uint32_t kChunk = 64 * kMergeBatch;       // records per wavefront per stage

struct Chunk {
  const uint32_t *ptr;  // first record (compact lines: start of the run)
  uint32_t rem;         // records from ptr to the end of the run (0 = no chunk)
  uint32_t first = 0;   // compact lines: index of the chunk's first record inside the run
  uint32_t rowBase = 0; // compact lines: row of the first row of the scanning workgroup's chunk
};
constexpr uint32_t kCompactLineRecords = 14;  // Workspace::lineRecords of the compact format

template <int RW>
struct RecStage {
  uint32_t w[kMergeBatch][RW];
};

// The loads are unconditional (the index is clamped into the run; `consume` ignores positions past
// the end): with branches around them the compiler cannot count outstanding loads and waits for the
// prefetch as well.  An absent chunk points at valid memory with rem = 0.
template <int RW>
__device__ __forceinline__ void load_chunk(RecStage<RW> &s, const Chunk &c, int lane) {
  const uint32_t last = c.rem ? c.rem - 1 : 0u;
#pragma unroll
  for (int k = 0; k < kMergeBatch; k++) {
    const uint32_t i = static_cast<uint32_t>(k) * 64u + lane;
    const uint32_t *p = c.ptr + static_cast<uint64_t>(i < last ? i : last) * RW;
    if constexpr (RW == 4) {
      const uint4 r = *reinterpret_cast<const uint4 *>(p);
      s.w[k][0] = r.x; s.w[k][1] = r.y; s.w[k][2] = r.z; s.w[k][3] = r.w;
    } else {
      const Rec3 r = *reinterpret_cast<const Rec3 *>(p);
      s.w[k][0] = r.row; s.w[k][1] = r.hash; s.w[k][2] = r.val;
    }
  }
}

// the same for a run of compact lines (Workspace::lineRecords == 14): record i of the run sits in line i / 14,
// half (i % 14) / 7, behind that half's header; it is expanded to {row, hash, carried measure, 0}
__device__ __forceinline__ void load_chunk_compact(RecStage<4> &s, const Chunk &c, int lane, int pb, uint32_t part) {
  const uint32_t last = c.rem ? c.rem - 1 : 0u;
#pragma unroll
  for (int k = 0; k < kMergeBatch; k++) {
    const uint32_t i = static_cast<uint32_t>(k) * 64u + lane;
    const uint32_t slot = c.first + (i < last ? i : last);
    const uint32_t line = slot / kCompactLineRecords, r = slot % kCompactLineRecords;
    const uint32_t half = r >= 7u ? 1u : 0u, k7 = r - 7u * half;
    const uint64_t *L = reinterpret_cast<const uint64_t *>(c.ptr) + static_cast<uint64_t>(line) * 16u + 8u * half;
    const uint64_t hdr = L[0], rec = L[1u + k7];
    const uint32_t lo9 = static_cast<uint32_t>(hdr >> (9u * k7)) & 511u, hiw = static_cast<uint32_t>(rec >> 32);
    s.w[k][0] = c.rowBase + (((hiw & ((1u << pb) - 1u)) << 9) | lo9);
    s.w[k][1] = (pb ? part << (32 - pb) : 0u) | (hiw >> pb);
    s.w[k][2] = static_cast<uint32_t>(rec);
    s.w[k][3] = 0u;
  }
}

// ND4 = number of dimensions when all are 4 bytes wide (vectorisable emission, grouped previous
// results), 0 = any layout.  FUSED: rows >= prevSize are source rows of the fused scan — their
// dimensions are re-evaluated from the columns; rows < prevSize are previous results in dimIn.
// RWB = words per region-B record.
template <int ND4, bool FUSED, int RWB>
__device__ __forceinline__ void merge_body(const uint8_t *__restrict__ dimIn, size_t inCapacity,
                                           const uint8_t *__restrict__ inValues, uint8_t *__restrict__ dimOut,
                                           const DimLayoutD &L, size_t capacity, uint8_t *__restrict__ outputValues,
                                           const AggSpec &a, const Workspace &ws, const FusedPlanD *plan, uint32_t prevSize) {
  __shared__ uint64_t sKeys[kSlots];
  __shared__ uint64_t sVals[kSlots];
  __shared__ uint32_t sRunCount[kMaxStreams];
  __shared__ uint32_t sClaimed, sOverflow, sCount, sBase, sEmit, sProgress, sTotal;
  const int p = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pb = ws.partBits;
  const int G = ws.streams;
  const uint32_t cursorA = ws.cursorsA[p];
  const uint32_t nA = cursorA < ws.capA ? cursorA : static_cast<uint32_t>(ws.capA);
  const uint4 *__restrict__ recA = ws.recA + static_cast<uint64_t>(p) * ws.capA;
  const uint32_t *prevRanges = (ND4 > 0 && ws.prevRanges) ? ws.prevRanges + static_cast<size_t>(p) * kRangeWords : nullptr;
  uint32_t nPrevRanges = prevRanges ? prevRanges[0] : 0u;
  if (nPrevRanges > static_cast<uint32_t>(kMaxRanges)) {  // the previous merge emitted more ranges than it could record
    if (threadIdx.x == 0) *ws_stale(ws) = 1u;
    nPrevRanges = 0;
  }
  if (threadIdx.x == 0) sTotal = 0;
  __syncthreads();
  {
    uint32_t mine = 0;
    if (static_cast<int>(threadIdx.x) < G) {
      mine = ws.countsB[static_cast<uint64_t>(threadIdx.x) * (1u << pb) + p];
      sRunCount[threadIdx.x] = mine;
    }
    if (threadIdx.x < nPrevRanges) mine += prevRanges[2 + 2 * threadIdx.x];
    if (threadIdx.x == 0) mine += nA;
    if (mine) __hip_atomic_fetch_add(&sTotal, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  const uint64_t n = sTotal;  // records + previous groups of this partition
  uint32_t rangesOut = 0;
  // Rounds over sub-ranges of the hash bits below the partition bits (left-aligned to 32 bits).
  // The first round is optimistic — the whole range: a partition usually holds far fewer groups than
  // the table has slots.  A round that runs out of slots stops streaming at once, and how far it
  // got tells how much narrower the next attempt must be, so a wrong guess costs a partial pass.
  uint64_t lo = 0, width = 1ull << 32;
  while (n > 0 && lo < (1ull << 32)) {
    clear_table(sKeys, sVals, a.identity);
    if (threadIdx.x == 0) { sClaimed = 0; sOverflow = 0; sCount = 0; sEmit = 0; sProgress = 0; }
    __syncthreads();
    const uint64_t hi = lo + width;
    uint32_t processed = 0;  // per wavefront (uniform)
    auto in_round = [&](uint32_t h) {
      const uint64_t u = static_cast<uint64_t>(pb ? (h << pb) : h);
      return u >= lo && u < hi;
    };

    // ---- previous groups of this partition: rows of the grouped previous result (always the lowest
    // row indices, so they stay the representatives)
    if (ND4 > 0) {
      const uint8_t *nullsIn = dimIn + static_cast<size_t>(4 * (ND4 > 0 ? ND4 : 1)) * inCapacity;
      for (uint32_t r = 0; r < nPrevRanges; r++) {
        const uint32_t start = prevRanges[1 + 2 * r], cnt = prevRanges[2 + 2 * r];
        for (uint32_t i = threadIdx.x; i < cnt; i += kThreads) {
          const uint32_t row = start + i;
          Murmur32Stream ms(0);
#pragma unroll
          for (int d = 0; d < ND4; d++)
            ms.push(*reinterpret_cast<const uint32_t *>(dimIn + static_cast<size_t>(4 * d) * inCapacity + 4ull * row), 4);
#pragma unroll
          for (int d = 0; d < ND4; d++) ms.push(nullsIn[static_cast<size_t>(d) * inCapacity + row], 1);
          const uint32_t h = ms.finish();
          if (row >= prevSize || (pb && (h >> (32 - pb)) != static_cast<uint32_t>(p))) {
            *ws_stale(ws) = 1u;  // not what the previous merge wrote: the host re-runs without the shortcut
          } else if (in_round(h)) {
            merge_record(sKeys, sVals, &sClaimed, &sOverflow, row, h, load_value_bits(inValues, a, row), a);
          }
        }
        processed += (cnt + kWaves - 1) / kWaves;
      }
    }

    // ---- region A (whole workgroup, chunks dealt round-robin to the wavefronts), then region B (each
    // wavefront streams whole runs).  Two register stages: the next chunk's loads are in flight while
    // the current one goes through the LDS table.
    const uint32_t *dummy = ws.cursorsA;
    {
      uint32_t offA = static_cast<uint32_t>(wave) * kChunk;
      auto next = [&]() -> Chunk {
        Chunk c{dummy, 0u};
        if (offA < nA) {
          c.ptr = reinterpret_cast<const uint32_t *>(recA + offA);
          c.rem = nA - offA;
          offA += kWaves * kChunk;
        }
        return c;
      };
      auto consume = [&](const RecStage<4> &s, const Chunk &c) {
        const uint32_t take = c.rem < kChunk ? c.rem : kChunk;
#pragma unroll
        for (int k = 0; k < kMergeBatch; k++) {
          const uint32_t i = static_cast<uint32_t>(k) * 64u + lane;
          if (i < take && in_round(s.w[k][1]))
            merge_record(sKeys, sVals, &sClaimed, &sOverflow, s.w[k][0], s.w[k][1],
                         (static_cast<uint64_t>(s.w[k][3]) << 32) | s.w[k][2], a);
        }
        processed += take;
      };
      RecStage<4> sa, sb;
      Chunk ca = next();
      load_chunk<4>(sa, ca, lane);
      while (ca.rem) {
        Chunk cb = next();
        load_chunk<4>(sb, cb, lane);
        consume(sa, ca);
        if (!cb.rem || __hip_atomic_load(&sOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        ca = next();
        load_chunk<4>(sa, ca, lane);
        consume(sb, cb);
        if (__hip_atomic_load(&sOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
      }
    }
    if (G > 0) {
      int g = wave;
      uint32_t offB = 0;
      const uint32_t capB = ws.capB;
      auto next = [&]() -> Chunk {
        Chunk c{dummy, 0u};
        while (g < G) {
          const uint32_t cnt = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(sRunCount[g])));
          if (offB < cnt) {
            if (RWB == 4 && ws.lineRecords == static_cast<int>(kCompactLineRecords)) {  // capB counts lines
              c.ptr = ws.recB + (static_cast<uint64_t>(g) * (1u << pb) + p) * capB * 32u;
              c.first = offB;
              c.rowBase = ws.rowBase + static_cast<uint32_t>(g) * ws.chunkRows;
            } else {
              c.ptr = ws.recB + ((static_cast<uint64_t>(g) * (1u << pb) + p) * capB + offB) * RWB;
            }
            c.rem = cnt - offB;
            offB += kChunk;
            break;
          }
          g += kWaves;
          offB = 0;
        }
        return c;
      };
      auto consume = [&](const RecStage<RWB> &s, const Chunk &c) {
        const uint32_t take = c.rem < kChunk ? c.rem : kChunk;
#pragma unroll
        for (int k = 0; k < kMergeBatch; k++) {
          const uint32_t i = static_cast<uint32_t>(k) * 64u + lane;
          if (i < take && in_round(s.w[k][1])) {
            uint64_t v;
            if constexpr (RWB == 4) {
              if (ws.lineRecords) {
                if (ws.lineRecords == 8 && s.w[k][0] == 0xFFFFFFFFu) continue;  // padding of a stream's last line
                v = ws.widen.mode == 2 ? (static_cast<uint64_t>(s.w[k][3]) << 32) | s.w[k][2]  // the whole value travels
                                       : widen_value(ws.widen, s.w[k][2]);
              } else {
                v = (static_cast<uint64_t>(s.w[k][3]) << 32) | s.w[k][2];
              }
            } else {
              v = widen_value(ws.widen, s.w[k][2]);
            }
            merge_record(sKeys, sVals, &sClaimed, &sOverflow, s.w[k][0], s.w[k][1], v, a);
          }
        }
        processed += take;
      };
      auto load = [&](RecStage<RWB> &s, const Chunk &c) {
        if constexpr (RWB == 4) {
          if (ws.lineRecords == static_cast<int>(kCompactLineRecords)) {
            load_chunk_compact(s, c, lane, pb, static_cast<uint32_t>(p));
            return;
          }
        }
        load_chunk<RWB>(s, c, lane);
      };
      RecStage<RWB> sa, sb;
      Chunk ca = next();
      load(sa, ca);
      while (ca.rem) {
        Chunk cb = next();
        load(sb, cb);
        consume(sa, ca);
        if (!cb.rem || __hip_atomic_load(&sOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        ca = next();
        load(sa, ca);
        consume(sb, cb);
        if (__hip_atomic_load(&sOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
      }
    }
    if (lane == 0 && processed) __hip_atomic_fetch_add(&sProgress, processed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const bool overflowed = sOverflow != 0;
    const uint64_t progress = sProgress;
    __syncthreads();  // everyone has read the flags before the next round resets them
    if (overflowed && width > 1) {  // too many groups in this sub-range: nothing is emitted
      // ~kMergeLimit groups showed up in the first `progress` records: aim for half a table per round
      const uint64_t floorP = static_cast<uint64_t>(kMergeBatch) * kThreads;
      uint64_t shrink = 2 * n / (progress > floorP ? progress : floorP);
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
    if (threadIdx.x == 0 && total) {
      const uint32_t base = atomicAdd(ws.outCount, total);
      sBase = base;
      if (ws.outRanges && rangesOut < static_cast<uint32_t>(kMaxRanges)) {
        uint32_t *o = ws.outRanges + static_cast<size_t>(p) * kRangeWords;
        o[1 + 2 * rangesOut] = base;
        o[2 + 2 * rangesOut] = total;
      }
    }
    if (total) rangesOut++;
    __syncthreads();
    if (total) {
      // one output range per wavefront per sweep: consecutive lanes write consecutive rows
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
          waveBase = __hip_atomic_fetch_add(&sEmit, static_cast<uint32_t>(__popcll(m)), __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_WORKGROUP);
        waveBase = __builtin_amdgcn_readfirstlane(waveBase);
        at[k] = sBase + waveBase +
                __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
      }
      if (ND4 > 0) {
        constexpr int NDX = ND4 > 0 ? ND4 : 1;
        // all gathers of the lane's groups are issued before the first store
        const uint8_t *nullsIn = dimIn + static_cast<size_t>(4 * NDX) * inCapacity;
        uint8_t *nullsOut = dimOut + static_cast<size_t>(4 * NDX) * capacity;
        // (in two halves: 4 groups x ND4 gathers per lane in flight keeps the kernel inside 128 VGPRs)
        constexpr int kHalf = kPerLane / 2;
#pragma unroll
        for (int half = 0; half < 2; half++) {
          uint32_t dv[kHalf][NDX];
          uint32_t nv[kHalf][NDX];
#pragma unroll
          for (int kk = 0; kk < kHalf; kk++) {
            const int k = half * kHalf + kk;
            if (!has[k]) continue;
            if (FUSED && row[k] >= prevSize) {
              fused_eval_row<NDX>(*plan, row[k] - prevSize, dv[kk], nv[kk]);
              continue;
            }
#pragma unroll
            for (int d = 0; d < NDX; d++) {
              dv[kk][d] = *reinterpret_cast<const uint32_t *>(dimIn + static_cast<size_t>(4 * d) * inCapacity + 4ull * row[k]);
              nv[kk][d] = nullsIn[static_cast<size_t>(d) * inCapacity + row[k]];
            }
          }
#pragma unroll
          for (int kk = 0; kk < kHalf; kk++) {
            const int k = half * kHalf + kk;
            if (!has[k]) continue;
#pragma unroll
            for (int d = 0; d < NDX; d++) {
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
  if (threadIdx.x == 0 && ws.outRanges) ws.outRanges[static_cast<size_t>(p) * kRangeWords] = rangesOut;
}

}  // namespace hr
}  // namespace ares
