// Device code of InitIndexVector, Unary/BinaryTransform and Unary/BinaryFilter for MI355X (gfx950): every kernel of
// transform.hip, which keeps the host side (operand binding, the deferral state machine, the ABI entry points).
// Included by transform.hip only.
//
// Reference behaviour: query/algorithm.cu:22-41, query/transform.cu:21-86 + transform.hpp:57-89, query/filter.cu:130-253.
// The reference runs thrust::transform (+ thrust::remove_if for filters: a second and third pass over the batch plus a
// host sync).  Here a transform is ONE pass — decode (value, validity) of every operand, apply the functor, write the sink,
// ITEMS independent coalesced loads per lane issued before any use —, several root transforms of a batch share one pass
// over the index vector (transform_multi_kernel), and a filter of the hot shape is counted in row space without writing
// anything observable (filter_rows_kernel) or, when the compacted vector is needed, runs as predicate pass + scan of the
// tile counts + chain-free in-place compaction.
#pragma once

#include <hip/hip_runtime.h>

#include "binding.hpp"
#include "common.hpp"
#include "device_model.hpp"
#include "fast_eval.hpp"
#include "lookback.hpp"

namespace ares {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

// ---------------------------------------------------------------------------------------------
// InitIndexVector
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void init_index_kernel(uint32_t *idx, uint32_t start, int n) {
  // 4 consecutive entries per lane -> one 16-byte store per lane
  const int64_t quads = (static_cast<int64_t>(n) + 3) >> 2;
  for (int64_t q = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; q < quads;
       q += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t i = q << 2;
    const uint32_t v = start + static_cast<uint32_t>(i);
    if (i + 3 < n && (reinterpret_cast<uintptr_t>(idx) & 15) == 0) {
      *reinterpret_cast<uint4 *>(idx + i) = make_uint4(v, v + 1, v + 2, v + 3);
    } else {
      for (int k = 0; k < 4 && i + k < n; k++) idx[i + k] = v + k;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 32-bit value path: transform
// ---------------------------------------------------------------------------------------------

template <int ITEMS>
__global__ __launch_bounds__(kBlock) void transform32_kernel(EvalParams p, SinkD s, int n) {
  const int64_t tile = static_cast<int64_t>(kBlock) * ITEMS;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * tile; base < n;
       base += static_cast<int64_t>(gridDim.x) * tile) {
    uint32_t rows[ITEMS];
    DVal va[ITEMS], vb[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int64_t i = base + k * kBlock + threadIdx.x;
      rows[k] = (i < n && p.needRow) ? p.idx[i] : static_cast<uint32_t>(i);
    }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int64_t i = base + k * kBlock + threadIdx.x;
      if (i < n) {
        va[k] = load32(p.a, static_cast<uint32_t>(i), rows[k], p.baseCounts, p.startCount);
        if (p.arity == 2) vb[k] = load32(p.b, static_cast<uint32_t>(i), rows[k], p.baseCounts, p.startCount);
      }
    }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int64_t i = base + k * kBlock + threadIdx.x;
      if (i < n) {
        DVal x = cvt32(va[k], p.a.kind, p.I);
        DVal r = p.arity == 1 ? unary32(p.functor, p.I, x)
                              : binary32(p.functor, p.I, x, cvt32(vb[k], p.b.kind, p.I));
        sink_store32(s, static_cast<uint32_t>(i), rows[k], r, p.rk);
      }
    }
  }
}

// A joined column copied into a dimension or measure vector — UnaryTransform(Noop) over a 32-bit foreign column
// (query/iterator.hpp:911-930: RecordID -> batch -> value) — the transform of a join query's hot path.  Three dependent loads
// per row (RecordID, the batch's descriptor, the value [+ its validity bit]): the descriptors sit in LDS and every lane keeps
// ITEMS rows in flight, phase by phase (the generic kernel above walks four rows through an interpreter: 1.38 ms per
// 64 Mi rows over a 50 M-row dimension table where the gathers alone need 0.6).
constexpr int kForeignBatchesInLds = 512;
template <int ITEMS>
__global__ __launch_bounds__(kBlock) void transform_foreign_kernel(EvalParams p, SinkD s, int n) {
  __shared__ ForeignBatchD sBatches[kForeignBatchesInLds];
  const OperandD &a = p.a;
  const bool inLds = a.numBatches <= kForeignBatchesInLds;
  if (inLds)
    for (int b = threadIdx.x; b < a.numBatches; b += kBlock) sBatches[b] = a.batches[b];
  __syncthreads();
  const int64_t tile = static_cast<int64_t>(kBlock) * ITEMS;
  for (int64_t base = static_cast<int64_t>(blockIdx.x) * tile; base < n; base += static_cast<int64_t>(gridDim.x) * tile) {
    RecordID rid[ITEMS];
    DVal v[ITEMS];
    const uint8_t *nullAt[ITEMS];
    uint32_t nullBit[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int64_t i = base + k * kBlock + threadIdx.x;
      rid[k] = i < n ? a.rids[i] : RecordID{0, 0};
    }
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      v[k].bits = 0;
      v[k].ok = 0;
      nullAt[k] = nullptr;
      nullBit[k] = 0;
      if (rid[k].batchID != 0 && (rid[k].batchID - a.baseBatchID < a.numBatches - 1 || rid[k].index < static_cast<uint32_t>(a.numRecLast))) {
        const ForeignBatchD b = inLds ? sBatches[rid[k].batchID - a.baseBatchID] : a.batches[rid[k].batchID - a.baseBatchID];
        if (b.isConst) {
          v[k].bits = a.cbits;
          v[k].ok = a.cok;
        } else {
          const uint32_t at = rid[k].index;
          v[k] = read_stored32(b.base + b.valuesOff, a.kind, a.step, at, at + b.bitOff);
          v[k].ok = 1u;
          if (b.valuesOff != 0) {  // (the validity byte is fetched with the values, looked at below)
            nullAt[k] = b.base + b.nullsOff + ((at + b.bitOff) >> 3);
            nullBit[k] = (at + b.bitOff) & 7u;
          }
        }
      }
    }
    uint32_t nb[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) nb[k] = nullAt[k] ? *nullAt[k] : 0xFFu;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
      const int64_t i = base + k * kBlock + threadIdx.x;
      if (i < n) {
        if (nullAt[k]) v[k].ok = (nb[k] >> nullBit[k]) & 1u;
        sink_store32(s, static_cast<uint32_t>(i), static_cast<uint32_t>(i), unary32(Noop, p.I, cvt32(v[k], a.kind, p.I)), p.rk);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wide value path (Int64 / UUID / GeoPoint first operand): rare, one element per lane
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ WVal load_wide(const OperandD &op, uint32_t i, uint32_t row, const uint32_t *baseCounts,
                                          uint32_t startCount) {
  WVal r;
  r.lo = r.hi = 0;
  r.ok = 0;
  const int w = op.kind == K_UUID ? 16 : 8;
  switch (op.type) {
    case OP_CONST:
      r.lo = op.c64[0]; r.hi = op.c64[1]; r.ok = op.cok;
      return r;
    case OP_SCRATCH:
      r.lo = *reinterpret_cast<const uint64_t *>(op.base + static_cast<size_t>(w) * i);
      if (w == 16) r.hi = *reinterpret_cast<const uint64_t *>(op.base + static_cast<size_t>(w) * i + 8);
      r.ok = op.base[op.nullsOff + i] != 0;
      return r;
    case OP_COLUMN: {
      const uint32_t p = locate(op, row, baseCounts, startCount);
      const uint8_t *v = op.base + op.valuesOff + static_cast<size_t>(w) * p;
      r.lo = *reinterpret_cast<const uint64_t *>(v);
      if (w == 16) r.hi = *reinterpret_cast<const uint64_t *>(v + 8);
      r.ok = op.mode >= 2 ? get_bit(op.base + op.nullsOff, p + op.bitOff) : 1u;
      return r;
    }
    default: {
      const RecordID rid = op.rids[i];
      if (rid.batchID != 0 && (rid.batchID - op.baseBatchID < op.numBatches - 1 ||
                               rid.index < static_cast<uint32_t>(op.numRecLast))) {
        const ForeignBatchD b = op.batches[rid.batchID - op.baseBatchID];
        if (b.isConst) { r.lo = op.c64[0]; r.hi = op.c64[1]; r.ok = op.cok; return r; }
        const uint8_t *v = b.base + b.valuesOff + static_cast<size_t>(w) * rid.index;
        r.lo = *reinterpret_cast<const uint64_t *>(v);
        if (w == 16) r.hi = *reinterpret_cast<const uint64_t *>(v + 8);
        r.ok = b.valuesOff != 0 ? get_bit(b.base + b.nullsOff, rid.index + b.bitOff) : 1u;
      }
      return r;
    }
  }
}

__device__ __forceinline__ void store_from_i64(const SinkD &s, uint32_t i, uint32_t row, int64_t v, uint32_t ok) {
  uint8_t *dst = s.values + static_cast<size_t>(s.width) * i;
  if (s.type == SINK_PRED) { s.values[i] = v != 0; return; }
  if (s.type == SINK_MEASURE) {
    if (!ok) {
      if (s.width == 8) *reinterpret_cast<uint64_t *>(dst) = s.identity;
      else *reinterpret_cast<uint32_t *>(dst) = static_cast<uint32_t>(s.identity);
      return;
    }
    const bool isAvg = s.agg == AGGR_AVG_FLOAT;
    uint32_t count = 1;
    if ((isAvg || (s.agg >= AGGR_SUM_UNSIGNED && s.agg <= AGGR_SUM_FLOAT)) && s.baseCounts)
      count = s.baseCounts[row + 1] - s.baseCounts[row];
    if (isAvg) {
      float f;
      switch (s.dtype) {
        case Float64: f = static_cast<float>(static_cast<double>(v)); break;
        case Int32: f = static_cast<float>(static_cast<int32_t>(v)); break;
        case Uint32: f = static_cast<float>(static_cast<uint32_t>(v)); break;
        default: f = static_cast<float>(v); break;
      }
      reinterpret_cast<uint32_t *>(dst)[0] = f_bits(f);
      reinterpret_cast<uint32_t *>(dst)[1] = count;
      return;
    }
    switch (s.dtype) {
      case Int32: case Uint32: *reinterpret_cast<uint32_t *>(dst) = static_cast<uint32_t>(v) * count; break;
      case Float32: *reinterpret_cast<float *>(dst) = static_cast<float>(v) * static_cast<float>(count); break;
      case Int64: *reinterpret_cast<uint64_t *>(dst) = static_cast<uint64_t>(v) * count; break;
      default: *reinterpret_cast<double *>(dst) = static_cast<double>(v) * static_cast<double>(count); break;
    }
    return;
  }
  switch (s.dtype) {
    case Bool: *dst = v != 0; break;
    case Int8: case Uint8: *dst = static_cast<uint8_t>(v); break;
    case Int16: case Uint16: *reinterpret_cast<uint16_t *>(dst) = static_cast<uint16_t>(v); break;
    case Int32: case Uint32: *reinterpret_cast<uint32_t *>(dst) = static_cast<uint32_t>(v); break;
    case Float32: *reinterpret_cast<float *>(dst) = static_cast<float>(v); break;
    case Int64: *reinterpret_cast<int64_t *>(dst) = v; break;
    default: break;
  }
  s.nulls[i] = ok ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void transform_wide_kernel(EvalParams p, SinkD s, int n) {
  for (int64_t i64 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i64 < n;
       i64 += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint32_t i = static_cast<uint32_t>(i64);
    const uint32_t row = p.needRow ? p.idx[i] : i;
    const WVal a = load_wide(p.a, i, row, p.baseCounts, p.startCount);
    const int K = p.a.kind;
    const int ft = p.functor;
    const bool sinkWide = s.type != SINK_PRED && s.type != SINK_MEASURE && (s.dtype == UUID || s.dtype == GeoPoint);
    if (p.arity == 2) {  // only Equal against a constant of the same kind (functor.hpp:1079-1133)
      DVal r;
      r.bits = 0;
      r.ok = 0;
      if (!sinkWide && ft == Equal) {
        const WVal b = load_wide(p.b, i, row, p.baseCounts, p.startCount);
        r.ok = a.ok && b.ok;
        if (r.ok) {
          if (K == K_UUID) r.bits = a.lo == b.lo && a.hi == b.hi;
          else {
            const float alat = bits_f(static_cast<uint32_t>(a.lo)), along = bits_f(static_cast<uint32_t>(a.lo >> 32));
            const float blat = bits_f(static_cast<uint32_t>(b.lo)), blong = bits_f(static_cast<uint32_t>(b.lo >> 32));
            r.bits = alat == blat && along == blong;
          }
        }
      }
      sink_store32(s, i, row, r, K_BOOL);
      continue;
    }
    if (K == K_UUID || K == K_GEO) {
      if (sinkWide) {  // X -> X is a copy, anything else (zero, null) (functor.hpp:800-881)
        const bool same = (K == K_UUID) == (s.dtype == UUID);
        uint8_t *dst = s.values + static_cast<size_t>(s.width) * i;
        reinterpret_cast<uint64_t *>(dst)[0] = same ? a.lo : 0;
        if (s.width == 16) reinterpret_cast<uint64_t *>(dst)[1] = same ? a.hi : 0;
        s.nulls[i] = (same && a.ok) ? 1 : 0;
        continue;
      }
      DVal r;
      r.bits = 0;
      r.ok = 0;
      if (K == K_UUID && ft == GetHLLValue && a.ok) {
        r.bits = hll_from_hash(a.lo ^ a.hi);
        r.ok = 1;
      }
      sink_store32(s, i, row, r, K_U32);
      continue;
    }
    // Int64 input: the generic unary functor (functor.hpp:660-697)
    if (sinkWide) {
      uint8_t *dst = s.values + static_cast<size_t>(s.width) * i;
      reinterpret_cast<uint64_t *>(dst)[0] = 0;
      if (s.width == 16) reinterpret_cast<uint64_t *>(dst)[1] = 0;
      s.nulls[i] = 0;
      continue;
    }
    const int64_t v = static_cast<int64_t>(a.lo);
    DVal r;
    switch (ft) {
      case Not: r.ok = a.ok; r.bits = a.ok ? (v == 0) : 0; sink_store32(s, i, row, r, K_BOOL); continue;
      case IsNull: r.ok = 1; r.bits = !a.ok; sink_store32(s, i, row, r, K_BOOL); continue;
      case IsNotNull: r.ok = 1; r.bits = a.ok != 0; sink_store32(s, i, row, r, K_BOOL); continue;
      case Negate: store_from_i64(s, i, row, a.ok ? static_cast<int64_t>(0ull - a.lo) : 0, a.ok); continue;
      case BitwiseNot: store_from_i64(s, i, row, a.ok ? ~v : 0, a.ok); continue;
      case GetHLLValue: {
        r.ok = a.ok;
        r.bits = 0;
        if (a.ok) {
          uint64_t q[2] = {a.lo, 0};
          r.bits = hll_from_hash(murmur3_128_lo<2>(q, 8, 0));
        }
        sink_store32(s, i, row, r, K_U32);
        continue;
      }
      default: break;
    }
    if (ft >= GetWeekStart && ft <= GetQuarterOfYear) {
      DVal t;
      t.bits = static_cast<uint32_t>(a.lo);
      t.ok = a.ok;
      sink_store32(s, i, row, unary32(ft, K_U32, t), K_U32);
      continue;
    }
    store_from_i64(s, i, row, v, a.ok);  // Noop and unknown functors
  }
}

// ---------------------------------------------------------------------------------------------
// array columns: ArrayLength / ArrayContains / ArrayElementAt (query/iterator.hpp:377-451,
// query/functor.hpp:468-640).  An array column is [offset u32, length u32] x Length followed by the
// array values [length u32][elements][validity bits]; output position i reads array i — the
// reference binds the bare iterator, not one zipped with the index vector (binder.hpp:385-426).
// One lane per array: the descriptors load coalesced, the element walks are short and divergent.
// ---------------------------------------------------------------------------------------------
struct ArrayD {
  const uint8_t *descriptors;  // OffsetLengthVector
  const uint8_t *values;       // descriptors + 8 * Length - ValueOffsetAdj
  int dtype, kind, width;      // element type
  int functor;                 // ArrayLength / ArrayContains / ArrayElementAt
  int enabled;                 // 0: the functor / sink / constant combination yields null everywhere
  int index;                   // ArrayElementAt
  uint64_t c[2];               // ArrayContains: the constant, already cast to the element type
  const uint32_t *idx;         // rows for a measure sink's run lengths
};

__device__ __forceinline__ bool array_elem_equals(const ArrayD &a, const uint8_t *e) {
  switch (a.dtype) {
    case Bool: return (*e != 0) == (a.c[0] != 0);
    case Int8: case Uint8: return *e == static_cast<uint8_t>(a.c[0]);
    case Int16: case Uint16: return *reinterpret_cast<const uint16_t *>(e) == static_cast<uint16_t>(a.c[0]);
    case Int32: case Uint32: return *reinterpret_cast<const uint32_t *>(e) == static_cast<uint32_t>(a.c[0]);
    case Float32: return bits_f(*reinterpret_cast<const uint32_t *>(e)) == bits_f(static_cast<uint32_t>(a.c[0]));
    case Int64: {  // elements start 4 bytes into an 8-byte aligned value: read words
      const uint32_t *w = reinterpret_cast<const uint32_t *>(e);
      return w[0] == static_cast<uint32_t>(a.c[0]) && w[1] == static_cast<uint32_t>(a.c[0] >> 32);
    }
    case GeoPoint: {
      const uint32_t *w = reinterpret_cast<const uint32_t *>(e);
      return bits_f(w[0]) == bits_f(static_cast<uint32_t>(a.c[0])) && bits_f(w[1]) == bits_f(static_cast<uint32_t>(a.c[0] >> 32));
    }
    default: {  // UUID
      const uint32_t *w = reinterpret_cast<const uint32_t *>(e);
      return w[0] == static_cast<uint32_t>(a.c[0]) && w[1] == static_cast<uint32_t>(a.c[0] >> 32) &&
             w[2] == static_cast<uint32_t>(a.c[1]) && w[3] == static_cast<uint32_t>(a.c[1] >> 32);
    }
  }
}

__global__ __launch_bounds__(kBlock) void array_transform_kernel(ArrayD a, SinkD s, int n) {
  const bool sinkWide = s.type != SINK_PRED && s.type != SINK_MEASURE && (s.dtype == UUID || s.dtype == GeoPoint);
  for (int64_t i64 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i64 < n;
       i64 += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint32_t i = static_cast<uint32_t>(i64);
    const uint32_t row = a.idx ? a.idx[i] : i;
    DVal r;
    r.bits = 0;
    r.ok = 0;
    int rk = K_U32;
    if (a.enabled) {
      const uint2 d = *reinterpret_cast<const uint2 *>(a.descriptors + 8ull * i);  // {offset, length}
      const uint8_t *value = d.y ? a.values + d.x : nullptr;
      const bool present = d.y != 0 || d.x != 0;  // (0, 0) is a null array, (x, 0) an empty one
      if (a.functor == ArrayLength) {
        r.ok = present;
        if (value) r.bits = *reinterpret_cast<const uint32_t *>(value);
      } else if (a.functor == ArrayContains) {
        rk = K_BOOL;
        r.ok = present;
        const int len = value ? static_cast<int>(*reinterpret_cast<const uint32_t *>(value)) : 0;
        if (len > 0) {
          const uint8_t *elems = value + 4, *valid = elems + static_cast<size_t>(a.width) * static_cast<uint32_t>(len);
          for (int j = 0; j < len; j++)
            if (((valid[j >> 3] >> (j & 7)) & 1) && array_elem_equals(a, elems + static_cast<size_t>(a.width) * j)) {
              r.bits = 1;
              break;
            }
        }
      } else if (value) {  // ArrayElementAt (functor.hpp:536-571): a negative index counts from the end
        const uint32_t ulen = *reinterpret_cast<const uint32_t *>(value);
        int index = a.index;
        const bool out = (index >= 0 && ulen <= static_cast<uint32_t>(index)) || (index < 0 && ulen < static_cast<uint32_t>(-index));
        const int len = static_cast<int>(ulen);
        if (index < 0) index = len + index;
        const uint8_t *elems = value + 4, *valid = elems + static_cast<size_t>(a.width) * ulen;
        if (!out && len != 0 && index < len && index >= 0 && ((valid[index >> 3] >> (index & 7)) & 1)) {
          const uint8_t *e = elems + static_cast<size_t>(a.width) * index;
          if (a.kind == K_UUID || a.kind == K_GEO) {  // only into a sink of its own type (checked on the host)
            uint8_t *dst = s.values + static_cast<size_t>(s.width) * i;
            const uint32_t *w = reinterpret_cast<const uint32_t *>(e);
            for (int k = 0; k < a.width / 4; k++) reinterpret_cast<uint32_t *>(dst)[k] = w[k];
            s.nulls[i] = 1;
            continue;
          }
          if (a.kind == K_I64) {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(e);
            const int64_t v = static_cast<int64_t>(static_cast<uint64_t>(w[0]) | (static_cast<uint64_t>(w[1]) << 32));
            if (sinkWide) {
              sink_store32(s, i, row, r, rk);
            } else {
              store_from_i64(s, i, row, v, 1u);
            }
            continue;
          }
          r.ok = 1;
          rk = a.kind;
          switch (a.dtype) {
            case Bool: r.bits = *e != 0; break;
            case Int8: r.bits = static_cast<uint32_t>(static_cast<int32_t>(*reinterpret_cast<const int8_t *>(e))); break;
            case Uint8: r.bits = *e; break;
            case Int16: r.bits = static_cast<uint32_t>(static_cast<int32_t>(*reinterpret_cast<const int16_t *>(e))); break;
            case Uint16: r.bits = *reinterpret_cast<const uint16_t *>(e); break;
            default: r.bits = *reinterpret_cast<const uint32_t *>(e); break;
          }
        }
      }
    }
    sink_store32(s, i, row, r, rk);
  }
}

// ---------------------------------------------------------------------------------------------
// filter: fused predicate + stable in-place compaction (decoupled look-back)
// ---------------------------------------------------------------------------------------------
constexpr int kFilterItems = 8;
constexpr int kFilterTile = kBlock * kFilterItems;
struct ScanWorkspace {
  unsigned int *ticket;  // next tile to process
  uint32_t *total;       // number of survivors
  uint32_t *error;       // raised when a look-back spin times out
  uint64_t *status;      // one word per tile: flag | count
};

// MODE 0: evaluate the predicate from the operands, write pred[], compact idx
// MODE 1: read pred[], compact one RecordID vector (8-byte payload)
// MODE 2: read pred[] (already evaluated by the wide-value kernel), compact idx
template <int MODE>
__global__ __launch_bounds__(kBlock) void filter_kernel(EvalParams p, uint8_t *pred, uint32_t *idx, uint64_t *rids,
                                                        ScanWorkspace ws, int n, int numTiles) {
  __shared__ uint32_t sCounts[kFilterItems * kWaves + 1];
  __shared__ int sTile;
  __shared__ uint32_t sBase;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t ltMask = (1ull << lane) - 1;
  for (;;) {
    __syncthreads();  // sTile / sCounts are reused from the previous tile
    if (threadIdx.x == 0) sTile = static_cast<int>(atomicAdd(ws.ticket, 1u));
    __syncthreads();
    const int tile = sTile;
    if (tile >= numTiles) break;
    const int64_t base = static_cast<int64_t>(tile) * kFilterTile;

    uint32_t rows[kFilterItems];
    uint64_t payload[MODE == 1 ? kFilterItems : 1];
    uint32_t keep[kFilterItems];
    if (MODE == 0) {
      DVal va[kFilterItems], vb[kFilterItems];
#pragma unroll
      for (int k = 0; k < kFilterItems; k++) {
        const int64_t i = base + k * kBlock + threadIdx.x;
        rows[k] = i < n ? idx[i] : 0u;
      }
#pragma unroll
      for (int k = 0; k < kFilterItems; k++) {
        const int64_t i = base + k * kBlock + threadIdx.x;
        if (i < n) {
          va[k] = load32(p.a, static_cast<uint32_t>(i), rows[k], p.baseCounts, p.startCount);
          if (p.arity == 2) vb[k] = load32(p.b, static_cast<uint32_t>(i), rows[k], p.baseCounts, p.startCount);
        }
      }
#pragma unroll
      for (int k = 0; k < kFilterItems; k++) {
        const int64_t i = base + k * kBlock + threadIdx.x;
        keep[k] = 0;
        if (i < n) {
          DVal x = cvt32(va[k], p.a.kind, p.I);
          DVal r = p.arity == 1 ? unary32(p.functor, p.I, x)
                                : binary32(p.functor, p.I, x, cvt32(vb[k], p.b.kind, p.I));
          keep[k] = cvt32(r, p.rk, K_BOOL).bits;  // validity is ignored (functor.hpp:903-915)
          pred[i] = static_cast<uint8_t>(keep[k]);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < kFilterItems; k++) {
        const int64_t i = base + k * kBlock + threadIdx.x;
        keep[k] = i < n ? pred[i] : 0u;
        if (MODE == 1) payload[k] = i < n ? rids[i] : 0ull;
        else rows[k] = i < n ? idx[i] : 0u;
      }
    }
    // every input of this tile is in registers before the tile's count becomes visible: later
    // tiles only start overwriting our input range after they have seen our status word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    uint32_t rank[kFilterItems];
#pragma unroll
    for (int k = 0; k < kFilterItems; k++) {
      const uint64_t m = __ballot(keep[k] != 0);
      rank[k] = __popcll(m & ltMask);
      if (lane == 0) sCounts[k * kWaves + wave] = __popcll(m);
    }
    __syncthreads();
    if (wave == 0) {
      // exclusive scan of the kFilterItems*kWaves (=32) partial counts, position order
      uint32_t c = lane < kFilterItems * kWaves ? sCounts[lane] : 0u;
      uint32_t incl = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
      }
      const uint32_t tileCount = __shfl(incl, 63);
      if (lane == 0) st_status(ws.status + tile, (tile == 0 ? kFlagInclusive : kFlagAggregate) | tileCount);
      uint32_t exclusive = 0;
      if (tile > 0) {
        exclusive = static_cast<uint32_t>(lookback_wave(ws.status, tile, lane, ws.error));
        if (lane == 0) st_status(ws.status + tile, kFlagInclusive | (exclusive + tileCount));
      }
      if (lane < kFilterItems * kWaves) sCounts[lane] = incl - c;
      if (lane == 0) {
        sBase = exclusive;
        if (tile == numTiles - 1) *ws.total = exclusive + tileCount;
      }
    }
    __syncthreads();
    const uint32_t blockBase = sBase;
#pragma unroll
    for (int k = 0; k < kFilterItems; k++) {
      if (keep[k]) {
        const uint32_t dst = blockBase + sCounts[k * kWaves + wave] + rank[k];
        if (MODE == 1) rids[dst] = payload[k];
        else idx[dst] = rows[k];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Quad geometry shared by the fast transform / filter kernels
// ---------------------------------------------------------------------------------------------
// Every lane owns QUADS groups of 4 CONSECUTIVE output positions, so the index vector is read and
// the outputs are written 16 bytes per lane per instruction (1 KiB per wavefront instruction), and
// the operand column is read 16 bytes per lane wherever the four rows are consecutive (always for a
// fresh index vector, mostly for a lightly filtered one).  gfx950 global accesses only need dword
// alignment, hence the 4-byte aligned vector types.  Quad k covers positions [4k - pad, 4k - pad + 4)
// with pad = (address of the 1-byte-per-position output) & 3, so that the validity / predicate
// bytes of a quad form one aligned dword.

// Loads rows / values / validity of QUADS quads per lane; positions outside [0, n) get ok = 0.
// Three phases so that every load of the tile is in flight before the first one is consumed:
// (A) index vector, (B) values + one 16-bit window of the validity bitmap per quad, (C) bit
// extraction.  The window starts at the byte holding the first row's bit and covers at least the 8
// following rows, which is where the other three rows of a filtered quad almost always lie; a
// wider quad re-reads single bytes.  Reading one byte past the bitmap is safe: in a mode-2 slice
// the values follow the bitmap inside the same allocation.
template <int QUADS>
__device__ __forceinline__ void load_rows(const uint32_t *idx, int pad, int64_t quad0, int n, uint32_t (&rows)[QUADS][4]) {
#pragma unroll
  for (int q = 0; q < QUADS; q++) {
    const int64_t i0 = (quad0 + static_cast<int64_t>(q) * kBlock) * 4 - pad;
    if (i0 >= 0 && i0 + 3 < n) {
      if (idx) {
        const U32x4 r = *reinterpret_cast<const U32x4 *>(idx + i0);
#pragma unroll
        for (int j = 0; j < 4; j++) rows[q][j] = r.v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) rows[q][j] = static_cast<uint32_t>(i0) + j;
      }
    } else {
      // ragged quad: positions outside [0, n) read row 0 (always a valid row) and are masked later
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int64_t i = i0 + j;
        rows[q][j] = (i >= 0 && i < n) ? (idx ? idx[i] : static_cast<uint32_t>(i)) : 0u;
      }
    }
  }
}

struct __attribute__((packed, aligned(1))) PU32x2 { uint32_t v[2]; };
struct __attribute__((packed, aligned(1))) PU32s1 { uint32_t v; };
template <int QUADS>
__device__ __forceinline__ void issue_values(const FastOperands &f, const uint32_t (&rows)[QUADS][4], uint32_t (&vals)[QUADS][4],
                                             uint32_t (&window)[QUADS]) {
#pragma unroll
  for (int q = 0; q < QUADS; q++) {
    const uint32_t r0 = rows[q][0];
    if (f.step == 2) {  // Int16 / Uint16 / BigEnum: 8 bytes hold the four rows of a consecutive quad (wave-uniform branch)
      const uint16_t *v16 = reinterpret_cast<const uint16_t *>(f.vals);
      uint32_t raw[4];
      if (rows[q][1] == r0 + 1 && rows[q][2] == r0 + 2 && rows[q][3] == r0 + 3) {
        const PU32x2 v = *reinterpret_cast<const PU32x2 *>(v16 + r0);
        raw[0] = v.v[0] & 0xFFFFu; raw[1] = v.v[0] >> 16; raw[2] = v.v[1] & 0xFFFFu; raw[3] = v.v[1] >> 16;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) raw[j] = v16[rows[q][j]];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) vals[q][j] = f.akind == K_I32 ? static_cast<uint32_t>(static_cast<int32_t>(static_cast<int16_t>(raw[j]))) : raw[j];
    } else if (f.step == 1) {  // Int8 / Uint8 / SmallEnum
      const uint8_t *v8 = reinterpret_cast<const uint8_t *>(f.vals);
      uint32_t raw[4];
      if (rows[q][1] == r0 + 1 && rows[q][2] == r0 + 2 && rows[q][3] == r0 + 3) {
        const uint32_t v = reinterpret_cast<const PU32s1 *>(v8 + r0)->v;
        raw[0] = v & 0xFFu; raw[1] = (v >> 8) & 0xFFu; raw[2] = (v >> 16) & 0xFFu; raw[3] = v >> 24;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) raw[j] = v8[rows[q][j]];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) vals[q][j] = f.akind == K_I32 ? static_cast<uint32_t>(static_cast<int32_t>(static_cast<int8_t>(raw[j]))) : raw[j];
    } else if (rows[q][1] == r0 + 1 && rows[q][2] == r0 + 2 && rows[q][3] == r0 + 3) {
      if (f.debug & 128) {  // streaming loads for a column that is read once (set by run_filter_rows)
        typedef uint32_t V4 __attribute__((ext_vector_type(4)));
        typedef V4 V4a __attribute__((aligned(4)));
        const V4 v = __builtin_nontemporal_load(reinterpret_cast<const V4a *>(f.vals + r0));
        vals[q][0] = v.x; vals[q][1] = v.y; vals[q][2] = v.z; vals[q][3] = v.w;
      } else {
        const U32x4 v = *reinterpret_cast<const U32x4 *>(f.vals + r0);
#pragma unroll
        for (int j = 0; j < 4; j++) vals[q][j] = v.v[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) vals[q][j] = f.vals[rows[q][j]];
    }
    window[q] = 0xFFFFu;
    if (f.nulls) window[q] = reinterpret_cast<const PU16 *>(f.nulls + ((r0 + f.bitOff) >> 3))->v;
  }
}

template <int QUADS>
__device__ __forceinline__ void extract_valid(const FastOperands &f, int pad, int64_t quad0, int n,
                                              const uint32_t (&rows)[QUADS][4], const uint32_t (&window)[QUADS],
                                              uint32_t (&okb)[QUADS]) {
#pragma unroll
  for (int q = 0; q < QUADS; q++) {
    const int64_t i0 = (quad0 + static_cast<int64_t>(q) * kBlock) * 4 - pad;
    const uint32_t first = (rows[q][0] + f.bitOff) & ~7u;  // bit position of the window's bit 0
    if (i0 >= 0 && i0 + 3 < n && rows[q][3] - rows[q][0] == 3u && rows[q][1] - rows[q][0] == 1u &&
        rows[q][2] - rows[q][0] == 2u) {
      // four consecutive rows inside the batch: their bits are contiguous in the 16-bit window
      okb[q] = (window[q] >> ((rows[q][0] + f.bitOff) & 7u)) & 0xFu;
      continue;
    }
    okb[q] = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t i = i0 + j;
      if (i < 0 || i >= n) continue;
      const uint32_t off = rows[q][j] + f.bitOff - first;  // wraps to a huge value when the row precedes the window
      uint32_t bit;
      if (off < 16u) bit = (window[q] >> off) & 1u;
      else bit = f.nulls ? get_bit(f.nulls, rows[q][j] + f.bitOff) : 1u;
      okb[q] |= bit << j;
    }
  }
}

template <int QUADS>
__device__ __forceinline__ void load_values(const FastOperands &f, int pad, int64_t quad0, int n,
                                            const uint32_t (&rows)[QUADS][4], uint32_t (&vals)[QUADS][4],
                                            uint32_t (&okb)[QUADS]) {
  uint32_t window[QUADS];
  issue_values<QUADS>(f, rows, vals, window);
  extract_valid<QUADS>(f, pad, quad0, n, rows, window, okb);
}

template <int QUADS>
__device__ __forceinline__ void load_quads(const FastOperands &f, int64_t quad0, int n, uint32_t (&rows)[QUADS][4],
                                           uint32_t (&vals)[QUADS][4], uint32_t (&okb)[QUADS]) {
  load_rows<QUADS>(f.idx, f.pad, quad0, n, rows);
  load_values<QUADS>(f, f.pad, quad0, n, rows, vals, okb);
}

// ---------------------------------------------------------------------------------------------
// fast transform: 32-bit column (x constant) -> 4-byte dimension / scratch vector or measure
// ---------------------------------------------------------------------------------------------
constexpr int kTQ = 4;  // quads per lane per tile: 16 rows per lane, 4096 rows per workgroup tile

struct __attribute__((packed, aligned(1))) PU32s { uint32_t v; };

// evaluates and stores one tile (kTQ quads per lane) of one transform whose rows are already loaded;
// ALIGNED_NULLS: the quad grid was shifted so that the 4 validity bytes form an aligned dword
template <int QUADS, bool ALIGNED_NULLS>
__device__ __forceinline__ void store_tile(const FastOperands &f, const SinkD &s, int pad, int64_t quad0, int n,
                                           const uint32_t (&rows)[QUADS][4], const uint32_t (&vals)[QUADS][4],
                                           const uint32_t (&okb)[QUADS]) {
  DVal y;
  y.bits = f.bbits;
  y.ok = f.bok;
  y = cvt32(y, f.bkind, f.I);
  const uint32_t ymag = (f.I == K_I32 && static_cast<int32_t>(y.bits) < 0) ? 0u - y.bits : y.bits;
  const FastDivisor fd = make_fast_divisor(ymag);
#pragma unroll
  for (int q = 0; q < QUADS; q++) {
    const int64_t i0 = (quad0 + static_cast<int64_t>(q) * kBlock) * 4 - pad;
    DVal r[4];
    {
      uint32_t rb[4];
      const uint32_t rok = eval_quad(f, vals[q], okb[q], y, fd, rb);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        r[j].bits = rb[j];
        r[j].ok = (rok >> j) & 1u;
      }
    }
    const bool full = i0 >= 0 && i0 + 3 < n;
    if (!full) {  // ragged first / last quad (or past the end): the generic element-wise sink
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int64_t i = i0 + j;
        if (i >= 0 && i < n) sink_store32(s, static_cast<uint32_t>(i), rows[q][j], r[j], f.rk);
      }
    } else if (s.type == SINK_MEASURE) {
      if (s.width == 8) {
        uint64_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (!r[j].ok) o[j] = s.identity;
          else if (s.dtype == Float64) o[j] = static_cast<uint64_t>(__double_as_longlong(to_double32(r[j], f.rk)));
          else o[j] = static_cast<uint64_t>(f.rk == K_F32 ? static_cast<int64_t>(bits_f(r[j].bits))
                                            : f.rk == K_I32 ? static_cast<int64_t>(static_cast<int32_t>(r[j].bits))
                                                            : static_cast<int64_t>(r[j].bits));
        }
        U64x2 lo, hi;
        lo.v[0] = o[0]; lo.v[1] = o[1]; hi.v[0] = o[2]; hi.v[1] = o[3];
        U64x2 *dst = reinterpret_cast<U64x2 *>(s.values + static_cast<size_t>(8) * i0);
        dst[0] = lo;
        dst[1] = hi;
      } else {
        U32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (!r[j].ok) o.v[j] = static_cast<uint32_t>(s.identity);
          else o.v[j] = cvt32(r[j], f.rk, s.dtype == Int32 ? K_I32 : s.dtype == Uint32 ? K_U32 : K_F32).bits;
        }
        *reinterpret_cast<U32x4 *>(s.values + static_cast<size_t>(4) * i0) = o;
      }
    } else if (s.width < 4) {  // 1- / 2-byte dimension slot (integer kinds only, see fast_sink): the value truncated, as store_typed32
      uint32_t nb = 0, lo = 0, hi = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) nb |= (r[j].ok ? 1u : 0u) << (8 * j);
      if (s.width == 2) {
        lo = (r[0].bits & 0xFFFFu) | (r[1].bits << 16);
        hi = (r[2].bits & 0xFFFFu) | (r[3].bits << 16);
        PU32x2 o;
        o.v[0] = lo; o.v[1] = hi;
        *reinterpret_cast<PU32x2 *>(s.values + static_cast<size_t>(2) * i0) = o;
      } else {
        lo = (r[0].bits & 0xFFu) | ((r[1].bits & 0xFFu) << 8) | ((r[2].bits & 0xFFu) << 16) | (r[3].bits << 24);
        reinterpret_cast<PU32s *>(s.values + i0)->v = lo;
      }
      if (ALIGNED_NULLS) *reinterpret_cast<uint32_t *>(s.nulls + i0) = nb;
      else reinterpret_cast<PU32s *>(s.nulls + i0)->v = nb;
    } else {  // 4-byte dimension / scratch value + one validity byte per row
      const int ok_kind = s.dtype == Int32 ? K_I32 : s.dtype == Uint32 ? K_U32 : K_F32;
      U32x4 o;
      uint32_t nb = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        o.v[j] = cvt32(r[j], f.rk, ok_kind).bits;
        nb |= (r[j].ok ? 1u : 0u) << (8 * j);
      }
      *reinterpret_cast<U32x4 *>(s.values + static_cast<size_t>(4) * i0) = o;
      if (ALIGNED_NULLS) *reinterpret_cast<uint32_t *>(s.nulls + i0) = nb;  // pad = nulls address & 3
      else reinterpret_cast<PU32s *>(s.nulls + i0)->v = nb;                // byte-aligned dword store
    }
  }
}

__global__ __launch_bounds__(kBlock) void transform_fast_kernel(FastOperands f, SinkD s, int n, int64_t numQuads) {
  const int64_t tileQuads = static_cast<int64_t>(kBlock) * kTQ;
  for (int64_t tq = static_cast<int64_t>(blockIdx.x) * tileQuads; tq < numQuads;
       tq += static_cast<int64_t>(gridDim.x) * tileQuads) {
    uint32_t rows[kTQ][4], vals[kTQ][4], okb[kTQ];
    load_rows<kTQ>(f.idx, f.pad, tq + threadIdx.x, n, rows);
    load_values<kTQ>(f, f.pad, tq + threadIdx.x, n, rows, vals, okb);
    store_tile<kTQ, true>(f, s, f.pad, tq + threadIdx.x, n, rows, vals, okb);
  }
}

// Several transforms of one batch over the same index vector in ONE pass (cross-call fusion, see
// include/ares_extensions.h): the index vector is read once per tile, each job then reads its own
// column and writes its own sink.  The quad grid is unshifted (pad 0): validity bytes use
// byte-aligned dword stores.  (Measured: running the jobs one after the other per tile, 16 rows per
// lane each, beats issuing every job's loads up front — 4.4 vs 3.2 TB/s on BASELINE config C3.)
constexpr int kMaxMultiJobs = 10;  // eight dimensions (MAX_DIMENSIONS) + the measure + one to spare
struct MultiJobs {
  int count;
  FastOperands f[kMaxMultiJobs];
  SinkD s[kMaxMultiJobs];
};

__global__ __launch_bounds__(kBlock, 3) void transform_multi_kernel(MultiJobs jobs, const uint32_t *idx, int n, int64_t numQuads) {
  const int64_t tileQuads = static_cast<int64_t>(kBlock) * kTQ;
  for (int64_t tq = static_cast<int64_t>(blockIdx.x) * tileQuads; tq < numQuads;
       tq += static_cast<int64_t>(gridDim.x) * tileQuads) {
    uint32_t rows[kTQ][4];
    load_rows<kTQ>(idx, 0, tq + threadIdx.x, n, rows);
    for (int j = 0; j < jobs.count; j++) {
      uint32_t vals[kTQ][4], okb[kTQ];
      load_values<kTQ>(jobs.f[j], 0, tq + threadIdx.x, n, rows, vals, okb);
      store_tile<kTQ, false>(jobs.f[j], jobs.s[j], 0, tq + threadIdx.x, n, rows, vals, okb);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fast filter: predicate + stable in-place compaction, 8192-row tiles
// ---------------------------------------------------------------------------------------------
// One returning atomic on the ticket word hands out a tile; a single word sustains ~90 tickets per
// microsecond on this chip, so the tile must be large (8192 rows -> > 700 G rows/s) for the ticket
// not to cap the kernel.  Survivors are ranked with a packed wavefront scan, staged in LDS in
// final order and written back with fully coalesced stores.
constexpr int kFQ = 8;
constexpr int kFastTile = kBlock * 4 * kFQ;

__global__ __launch_bounds__(kBlock) void filter_fast_kernel(FastOperands f, uint8_t *pred, uint32_t *idx,
                                                             ScanWorkspace ws, int n, int numTiles) {
  __shared__ uint32_t sOut[kFastTile];
  __shared__ uint32_t sCounts[kFQ * kWaves];
  __shared__ int sTile;
  __shared__ uint32_t sBase, sTileCount;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  DVal y;
  y.bits = f.bbits;
  y.ok = f.bok;
  y = cvt32(y, f.bkind, f.I);
  for (int iter = 0;; iter++) {
    __syncthreads();  // LDS of the previous tile is free again
    if (threadIdx.x == 0)
      sTile = (f.debug & 4) ? static_cast<int>(blockIdx.x + iter * gridDim.x) : static_cast<int>(atomicAdd(ws.ticket, 1u));
    __syncthreads();
    const int tile = sTile;
    if (tile >= numTiles) break;
    const int64_t tq = static_cast<int64_t>(tile) * (kBlock * kFQ);

    uint32_t rows[kFQ][4], vals[kFQ][4], okb[kFQ];
    load_quads<kFQ>(f, tq + threadIdx.x, n, rows, vals, okb);
    uint32_t keep = 0;
#pragma unroll
    for (int q = 0; q < kFQ; q++) {
      const int64_t i0 = (tq + threadIdx.x + static_cast<int64_t>(q) * kBlock) * 4 - f.pad;
      uint32_t kb = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int64_t i = i0 + j;
        if (i >= 0 && i < n) {
          kb |= compare_fast(f, vals[q][j], (okb[q] >> j) & 1u, y) << j;  // result validity is ignored (functor.hpp:903-915)
        }
      }
      keep |= kb << (4 * q);
      const uint32_t bytes = (kb & 1u) | ((kb & 2u) << 7) | ((kb & 4u) << 14) | ((kb & 8u) << 21);
      if (i0 >= 0 && i0 + 3 < n) {
        *reinterpret_cast<uint32_t *>(pred + i0) = bytes;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (i0 + j >= 0 && i0 + j < n) pred[i0 + j] = static_cast<uint8_t>((kb >> j) & 1u);
      }
    }
    // Every index-vector word of this tile has been consumed (the predicate depends on it) before
    // the tile's count becomes visible: later tiles only overwrite our input range after that.

    // rank of every survivor inside its wavefront: ballots + mbcnt (pure VALU/SALU, no cross-lane
    // traffic); position order inside the tile is (quad, lane, j)
    uint32_t lanePrefix[kFQ];
#pragma unroll
    for (int q = 0; q < kFQ; q++) {
      uint32_t before = 0, total = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint64_t m = __ballot((keep >> (4 * q + j)) & 1u);
        before += __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
        total += static_cast<uint32_t>(__popcll(m));
      }
      lanePrefix[q] = before;
      if (lane == 0) sCounts[q * kWaves + wave] = total;
    }
    __syncthreads();
    if (wave == 0) {
      // exclusive scan of the kFQ * kWaves (= 32) partial counts in position order
      uint32_t c = lane < kFQ * kWaves ? sCounts[lane] : 0u;
      uint32_t incl = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
      }
      const uint32_t tileCount = __shfl(incl, 63);
      if (lane == 0) st_status(ws.status + tile, (tile == 0 ? kFlagInclusive : kFlagAggregate) | tileCount);
      uint32_t exclusive = 0;
      if (tile > 0 && !(f.debug & 1)) {
        exclusive = static_cast<uint32_t>(lookback_wave(ws.status, tile, lane, ws.error));
        if (lane == 0) st_status(ws.status + tile, kFlagInclusive | (exclusive + tileCount));
      }
      if (f.debug & 1) exclusive = static_cast<uint32_t>(tile) * 7000u;
      if (lane < kFQ * kWaves) sCounts[lane] = incl - c;
      if (lane == 0) {
        sBase = exclusive;
        sTileCount = (f.debug & 2) ? 0u : tileCount;
        if (tile == numTiles - 1) *ws.total = exclusive + tileCount;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kFQ; q++) {
      uint32_t at = sCounts[q * kWaves + wave] + lanePrefix[q];
      const uint32_t kb = (keep >> (4 * q)) & 0xFu;
#pragma unroll
      for (int j = 0; j < 4; j++)
        if ((kb >> j) & 1u) sOut[at++] = rows[q][j];
    }
    __syncthreads();
    const uint32_t count = sTileCount, gbase = sBase;
    for (uint32_t k = threadIdx.x; k < count; k += kBlock) idx[gbase + k] = sOut[k];
  }
}

// ---------------------------------------------------------------------------------------------
// two-phase filter: predicate + per-tile counts, scan, in-place compaction without a chain
// ---------------------------------------------------------------------------------------------
// The one-pass kernel above is bound by the latency of its per-tile chain (index -> column loads,
// ranking, look-back, staging).  Splitting it removes every inter-tile dependency from the hot
// loops: (1) a streaming kernel evaluates the predicate, writes the predicate bytes and one
// survivor count per 4096-row tile; (2) a single small workgroup turns the counts into offsets;
// (3) the compaction kernel re-reads predicate bytes + index vector and writes each tile's
// survivors at its known offset.  In-place safety in (3): a tile writes only after every tile whose
// INPUT range overlaps its OUTPUT range has flagged "loaded" (those are earlier tiles, claimed
// earlier by running workgroups through the ticket, so the wait cannot deadlock); with a virtual
// index vector (rows = position, see the iota registry below) nothing is read from the index
// vector and no wait is needed.  Traffic: 9.1 + 8.6 B/row instead of 12.7, at streaming speed.
constexpr int kPQ = 4;                        // quads per lane per tile
constexpr int kPTile = kBlock * 4 * kPQ;      // 4096 rows
constexpr int kTilesPerTicket = 4;

// blockTotals (not null): one word per workgroup receives the survivors of its tiles — the host adds them up, so that
// the count a filter returns does not wait for the scan of the tile counts (only a compaction needs the offsets) and no
// word is the target of thousands of atomics (4096 workgroups adding to ONE word measured +0.017 ms on a 0.113 ms kernel).
__global__ __launch_bounds__(kBlock) void filter_pred_kernel(FastOperands f, uint8_t *pred, uint32_t *tileCounts, int n,
                                                             int numTiles, uint32_t *blockTotals) {
  __shared__ uint32_t sTotal;
  if (threadIdx.x == 0) sTotal = 0;
  uint32_t mine = 0;  // lane 0 of each wavefront: survivors of the tiles this workgroup has seen
  const int lane = threadIdx.x & 63;
  DVal y;
  y.bits = f.bbits;
  y.ok = f.bok;
  y = cvt32(y, f.bkind, f.I);
  for (int tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
    const int64_t tq = static_cast<int64_t>(tile) * (kBlock * kPQ);
    uint32_t rows[kPQ][4], vals[kPQ][4], okb[kPQ];
    load_quads<kPQ>(f, tq + threadIdx.x, n, rows, vals, okb);
    uint32_t count = 0, in[kPQ], kbs[kPQ];
#pragma unroll
    for (int q = 0; q < kPQ; q++) {
      const int64_t i0 = (tq + threadIdx.x + static_cast<int64_t>(q) * kBlock) * 4 - f.pad;
      in[q] = 0xFu;
      if (i0 < 0 || i0 + 3 >= n) {
        in[q] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) in[q] |= ((i0 + j >= 0 && i0 + j < n) ? 1u : 0u) << j;
      }
    }
    compare_tile<kPQ>(f, vals, okb, in, y, kbs);  // result validity is ignored (functor.hpp:903-915)
#pragma unroll
    for (int q = 0; q < kPQ; q++) {
      const int64_t i0 = (tq + threadIdx.x + static_cast<int64_t>(q) * kBlock) * 4 - f.pad;
      const uint32_t kb = kbs[q];
      count += __popc(kb);
      const uint32_t bytes = (kb & 1u) | ((kb & 2u) << 7) | ((kb & 4u) << 14) | ((kb & 8u) << 21);
      if (i0 >= 0 && i0 + 3 < n) {
        *reinterpret_cast<uint32_t *>(pred + i0) = bytes;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (i0 + j >= 0 && i0 + j < n) pred[i0 + j] = static_cast<uint8_t>((kb >> j) & 1u);
      }
    }
    // one atomic per wavefront into the (zeroed) tile count: no workgroup barrier in this kernel
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) count += __shfl_xor(count, off);
    if (lane == 0 && count) atomicAdd(tileCounts + tile, count);
    mine += count;
  }
  if (!blockTotals) return;
  __syncthreads();
  if (lane == 0 && mine) atomicAdd(&sTotal, mine);
  __syncthreads();
  if (threadIdx.x == 0) blockTotals[blockIdx.x] = sTotal;
}

// ---------------------------------------------------------------------------------------------
// lazy filter in ROW space: count + survivor bits, no predicate bytes, no index vector
// ---------------------------------------------------------------------------------------------
// While every filter of a batch so far is of the hot shape and nothing has read the index vector, the vector is still
// iota(0 .. n0) "with k filters pending": the survivors of filter k + 1 are the rows of the batch that pass filters
// 1 .. k + 1, whatever order compactions would have put them in.  This kernel evaluates ONE more filter over the batch's
// rows (rows = positions: no index vector is read), ANDs it with the survivors so far — one bit per row — and returns
// the count through one partial per workgroup: 4.1 B/row read + 0.13 B/row of bits each way, against predicate bytes,
// compaction and a gather through the compacted vector for every filter after the first (the Go host puts two time
// filters in front of every fact-table query's own filters, query/aql_processor.go:543-559).  Nothing observable is
// written: the predicate and index vectors are produced by replaying the filters with the kernels above when (if)
// somebody needs them (run_compaction).
// Survivor bits are kept in the kernel's own lane layout: the tile geometry of this file gives lane t of a workgroup the
// rows (1024 tile + 256 q + t) * 4 + j of quads q = 0..3 — 16 rows per lane and tile — so a tile's survivors are one
// 16-bit word per lane (bit 4 q + j), written and read back with ONE coalesced 2-byte access per lane and tile.  (A
// first version kept ballots — bit = lane — and paid 16 ballots, 16 population counts and 16 single-lane stores per
// wavefront and tile: 0.102 ms per 64 Mi rows against the 0.079 ms of the kernel that writes predicate bytes.)
// `TWO`: a second filter `g` over the SAME column is evaluated as well (the host predicts it from the previous batch of
// the stream: ts >= from is followed by ts < to); blockTotals holds two partials per workgroup, bitsOut2 the survivors
// of both.
template <bool TWO, bool HAS_IN>
__global__ __launch_bounds__(kBlock) void filter_rows_kernel(FastOperands f, FastOperands g, const uint16_t *bitsIn,
                                                             uint16_t *bitsOut, uint16_t *bitsOut2, int n, int numTiles,
                                                             uint32_t *blockTotals) {
  __shared__ uint32_t sTotal[2];
  if (threadIdx.x < 2) sTotal[threadIdx.x] = 0;
  uint32_t mine = 0, mine2 = 0;  // survivors among this lane's rows
  const int lane = threadIdx.x & 63;
  DVal y, z;
  y.bits = f.bbits;
  y.ok = f.bok;
  y = cvt32(y, f.bkind, f.I);
  z.bits = g.bbits;
  z.ok = g.bok;
  z = cvt32(z, g.bkind, g.I);
  for (int tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
    const int64_t tq = static_cast<int64_t>(tile) * (kBlock * kPQ);
    const size_t word = static_cast<size_t>(tile) * kBlock + threadIdx.x;
    uint32_t alive = 0xFFFFu;
    if (HAS_IN) alive = bitsIn[word];
    uint32_t rows[kPQ][4], vals[kPQ][4], okb[kPQ];
    load_quads<kPQ>(f, tq + threadIdx.x, n, rows, vals, okb);
    uint32_t in[kPQ], kbs[kPQ], kbs2[kPQ];
#pragma unroll
    for (int q = 0; q < kPQ; q++) {
      const int64_t i0 = (tq + threadIdx.x + static_cast<int64_t>(q) * kBlock) * 4;
      in[q] = 0xFu;
      if (i0 + 3 >= n) {
        in[q] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) in[q] |= (i0 + j < n ? 1u : 0u) << j;
      }
    }
    compare_tile<kPQ>(f, vals, okb, in, y, kbs);  // result validity is ignored (functor.hpp:903-915)
    uint32_t k1 = 0;
#pragma unroll
    for (int q = 0; q < kPQ; q++) k1 |= kbs[q] << (4 * q);
    k1 &= alive;
    if (f.debug & 8) {  // (experiment: pairs of lanes store one dword instead of two shorts)
      const uint32_t other = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(k1), 0xB1, 0xF, 0xF, true));
      if (!(lane & 1)) reinterpret_cast<uint32_t *>(bitsOut)[word >> 1] = k1 | (other << 16);
    } else {
      bitsOut[word] = static_cast<uint16_t>(k1);
    }
    mine += __popc(k1);
    if (TWO) {
      compare_tile<kPQ>(g, vals, okb, in, z, kbs2);
      uint32_t k2 = 0;
#pragma unroll
      for (int q = 0; q < kPQ; q++) k2 |= kbs2[q] << (4 * q);
      k2 &= k1;
      bitsOut2[word] = static_cast<uint16_t>(k2);
      mine2 += __popc(k2);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mine += __shfl_xor(mine, off);
    if (TWO) mine2 += __shfl_xor(mine2, off);
  }
  __syncthreads();
  if (lane == 0) {
    if (mine) atomicAdd(&sTotal[0], mine);
    if (mine2) atomicAdd(&sTotal[1], mine2);
  }
  __syncthreads();
  if (threadIdx.x < 2) blockTotals[2 * blockIdx.x + threadIdx.x] = sTotal[threadIdx.x];
}

// exclusive scan of the tile counts by ONE workgroup (numTiles <= a few hundred thousand)
__global__ __launch_bounds__(1024) void filter_scan_kernel(const uint32_t *tileCounts, uint32_t *tileOffsets, int numTiles,
                                                           uint32_t *total) {
  constexpr int kPer = 8;  // consecutive tiles per lane
  __shared__ uint32_t sWave[16];
  __shared__ uint32_t sCarry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) sCarry = 0;
  __syncthreads();
  for (int base = 0; base < numTiles; base += 1024 * kPer) {
    const int t0 = base + threadIdx.x * kPer;
    uint32_t c[kPer], mine = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      c[k] = t0 + k < numTiles ? tileCounts[t0 + k] : 0u;
      mine += c[k];
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    uint32_t before = sCarry;
    for (int w = 0; w < wave; w++) before += sWave[w];
    uint32_t run = before + incl - mine;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      if (t0 + k < numTiles) tileOffsets[t0 + k] = run;
      run += c[k];
    }
    __syncthreads();
    if (threadIdx.x == 1023) sCarry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    tileOffsets[numTiles] = sCarry;
    *total = sCarry;
  }
}

struct CompactWorkspace {
  unsigned int *ticket;
  uint32_t *error;
  const uint32_t *tileOffsets;
  uint32_t *loaded;  // one word per tile: its input is in registers
};

// PAYLOAD: uint32_t (index vector) or uint64_t (RecordID vector).  VIRTUAL: the index vector is
// iota(start) and has not been materialised: nothing is read from it.
template <typename PAYLOAD, bool VIRTUAL>
__global__ __launch_bounds__(kBlock) void filter_compact_kernel(const uint8_t *pred, PAYLOAD *data, uint32_t iotaStart, int pad,
                                                                CompactWorkspace ws, int n, int numTiles) {
  __shared__ PAYLOAD sOut[kPTile];
  __shared__ uint32_t sCounts[kPQ * kWaves];
  __shared__ int sTicket;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) sTicket = static_cast<int>(atomicAdd(ws.ticket, 1u));
    __syncthreads();
    const int firstTile = sTicket * kTilesPerTicket;
    if (firstTile >= numTiles) break;
    // phase A: the input of ALL tiles of this ticket goes to registers, then they are flagged
    // "loaded" at once — nobody ever waits for a tile this workgroup has not reached yet
    PAYLOAD rows[kTilesPerTicket][kPQ][4];
    uint32_t keep[kTilesPerTicket];
#pragma unroll
    for (int tt = 0; tt < kTilesPerTicket; tt++) {
      const int tile = firstTile + tt;
      const int64_t tq = static_cast<int64_t>(tile) * (kBlock * kPQ);
      keep[tt] = 0;
#pragma unroll
      for (int q = 0; q < kPQ; q++) {
        const int64_t i0 = (tq + threadIdx.x + static_cast<int64_t>(q) * kBlock) * 4 - pad;
        uint32_t pb = 0;
        if (tile < numTiles && i0 >= 0 && i0 + 3 < n) {
          pb = *reinterpret_cast<const uint32_t *>(pred + i0);
          if (VIRTUAL) {
#pragma unroll
            for (int j = 0; j < 4; j++) rows[tt][q][j] = static_cast<PAYLOAD>(iotaStart + static_cast<uint32_t>(i0) + j);
          } else if (sizeof(PAYLOAD) == 4) {
            const U32x4 r = *reinterpret_cast<const U32x4 *>(data + i0);
#pragma unroll
            for (int j = 0; j < 4; j++) rows[tt][q][j] = static_cast<PAYLOAD>(r.v[j]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++) rows[tt][q][j] = data[i0 + j];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int64_t i = i0 + j;
            rows[tt][q][j] = 0;
            if (tile < numTiles && i >= 0 && i < n) {
              pb |= static_cast<uint32_t>(pred[i]) << (8 * j);
              rows[tt][q][j] = VIRTUAL ? static_cast<PAYLOAD>(iotaStart + static_cast<uint32_t>(i)) : data[i];
            }
          }
        }
        const uint32_t kb = ((pb & 0xFFu) ? 1u : 0u) | ((pb & 0xFF00u) ? 2u : 0u) | ((pb & 0xFF0000u) ? 4u : 0u) |
                            ((pb & 0xFF000000u) ? 8u : 0u);
        keep[tt] |= kb << (4 * q);
      }
    }
    if (!VIRTUAL) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x < kTilesPerTicket && firstTile + static_cast<int>(threadIdx.x) < numTiles)
        __hip_atomic_store(ws.loaded + firstTile + threadIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // phase B: rank, stage in final order, wait for the input tiles under the output range, write
#pragma unroll
    for (int tt = 0; tt < kTilesPerTicket; tt++) {
      const int tile = firstTile + tt;
      if (tile >= numTiles) break;
      uint32_t lanePrefix[kPQ];
#pragma unroll
      for (int q = 0; q < kPQ; q++) {
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint64_t m = __ballot((keep[tt] >> (4 * q + j)) & 1u);
          before += __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
          total += static_cast<uint32_t>(__popcll(m));
        }
        lanePrefix[q] = before;
        if (lane == 0) sCounts[q * kWaves + wave] = total;
      }
      __syncthreads();
      uint32_t base[kPQ];
      {
        uint32_t run = 0;  // every lane scans the 16 partial counts (position order: quad, wave)
#pragma unroll
        for (int q = 0; q < kPQ; q++)
#pragma unroll
          for (int w = 0; w < kWaves; w++) {
            if (w == wave) base[q] = run;
            run += sCounts[q * kWaves + w];
          }
      }
#pragma unroll
      for (int q = 0; q < kPQ; q++) {
        uint32_t at = base[q] + lanePrefix[q];
        const uint32_t kb = (keep[tt] >> (4 * q)) & 0xFu;
#pragma unroll
        for (int j = 0; j < 4; j++)
          if ((kb >> j) & 1u) sOut[at++] = rows[tt][q][j];
      }
      const uint32_t gbase = ws.tileOffsets[tile], count = ws.tileOffsets[tile + 1] - gbase;
      if (!VIRTUAL && count > 0) {
        const int lo = static_cast<int>((static_cast<int64_t>(gbase) + pad) / kPTile);
        const int hi = static_cast<int>((static_cast<int64_t>(gbase) + count - 1 + pad) / kPTile);
        if (threadIdx.x == 0) {
          for (int s2 = lo; s2 <= hi && s2 < firstTile; s2++) {  // this ticket's own tiles are loaded
            uint32_t spins = 0;
            while (__hip_atomic_load(ws.loaded + s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
              __builtin_amdgcn_s_sleep(2);
              if (++spins > kMaxSpins) {
                atomicOr(ws.error, 1u);
                break;
              }
            }
          }
        }
      }
      __syncthreads();
      for (uint32_t k = threadIdx.x; k < count; k += kBlock) data[gbase + k] = sOut[k];
      __syncthreads();  // sOut / sCounts are reused by the next tile
    }
  }
}

}  // namespace ares
