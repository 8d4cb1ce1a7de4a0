// Device-side value model of the AQL operators: how a (value, validity) pair is decoded from a
// column, combined by a functor and written to a sink.
//
// Semantics follow the reference exactly (query/iterator.hpp:62-289,465-537,616-727,845-931;
// query/functor.hpp:30-351,660-1076; query/utils.hpp:83-94,169-184) but the structure is new:
// instead of one Thrust launch per (iterator type x functor type x output type) template
// instantiation, a kernel receives plain descriptors (OperandD / SinkD) and takes wave-uniform
// branches on them; all lanes of every wavefront follow the same path, so the dispatch costs
// scalar instructions only and the kernels stay HBM-bound.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "ares_algorithm.h"

namespace ares {

// value kinds (the reference's iterator value_type::head_type)
enum : int { K_BOOL = 0, K_I32 = 1, K_U32 = 2, K_F32 = 3, K_I64 = 4, K_UUID = 5, K_GEO = 6, K_NONE = 7 };

enum : int { OP_CONST = 0, OP_COLUMN = 1, OP_SCRATCH = 2, OP_FOREIGN = 3 };
enum : int { SINK_PRED = 0, SINK_SCRATCH = 1, SINK_DIM = 2, SINK_MEASURE = 3 };

// One batch of a dimension-table column as seen by the join (query/iterator.hpp:800-842).
struct ForeignBatchD {
  const uint8_t *base;
  uint32_t nullsOff;
  uint32_t valuesOff;
  uint8_t bitOff;
  uint8_t isConst;
};

struct OperandD {
  int type;
  int kind;  // storage kind
  // OP_CONST (also a mode-0 column): value bits and validity; wide constants use c64
  uint32_t cbits;
  uint32_t cok;
  uint64_t c64[2];
  // OP_COLUMN / OP_SCRATCH
  const uint8_t *base;
  uint32_t nullsOff;
  uint32_t valuesOff;
  uint32_t length;
  uint8_t mode;  // 1 values only, 2 validity+values, 3 run-length counts+validity+values
  uint8_t step;  // bytes per stored value
  uint8_t bitOff;
  // OP_FOREIGN
  const RecordID *rids;
  const ForeignBatchD *batches;
  int32_t baseBatchID;
  int32_t numBatches;
  int32_t numRecLast;
  const int16_t *tz;
  int32_t tzSize;
};

struct SinkD {
  int type;
  int dtype;  // enum DataType of the stored element
  int width;  // bytes per stored element
  uint8_t *values;
  uint8_t *nulls;
  int agg;
  uint64_t identity;  // measure identity, already in the measure's own byte representation
  const uint32_t *baseCounts;
};

struct DVal {  // 32-bit kinds: bool/int32/uint32/float bits + validity
  uint32_t bits;
  uint32_t ok;
};

struct WVal {  // wide kinds: int64 (lo), GeoPoint (lo = {lat,long}), UUID (lo,hi)
  uint64_t lo, hi;
  uint32_t ok;
};

__device__ __forceinline__ float bits_f(uint32_t b) { return __uint_as_float(b); }
__device__ __forceinline__ uint32_t f_bits(float f) { return __float_as_uint(f); }

__device__ __forceinline__ uint32_t get_bit(const uint8_t *p, uint32_t i) {
  return (p[i >> 3] >> (i & 7)) & 1u;
}

// static_cast between the 32-bit kinds (implicit thrust::tuple<A,bool> -> tuple<B,bool>)
__device__ __forceinline__ DVal cvt32(DVal x, int from, int to) {
  if (from == to) return x;
  DVal r;
  r.ok = x.ok;
  switch (to) {
    case K_BOOL: r.bits = (from == K_F32) ? (bits_f(x.bits) != 0.0f) : (x.bits != 0u); break;
    case K_I32: r.bits = (from == K_F32) ? static_cast<uint32_t>(static_cast<int32_t>(bits_f(x.bits))) : x.bits; break;
    case K_U32: r.bits = (from == K_F32) ? static_cast<uint32_t>(bits_f(x.bits)) : x.bits; break;
    default:  // K_F32
      r.bits = (from == K_I32) ? f_bits(static_cast<float>(static_cast<int32_t>(x.bits)))
                               : f_bits(static_cast<float>(x.bits));
      break;
  }
  return r;
}

// ---- murmur3 (query/utils.cu:113-241) on zero-padded little-endian words -------------------
// With the key zero-padded to a whole number of words the reference's byte-wise tail switch
// reduces to "mix the (possibly zero) tail words": mixing a zero word is the identity.
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

template <int MAXW>
__device__ __forceinline__ uint32_t murmur3_32_words(const uint32_t (&w)[MAXW], int bytes, uint32_t seed) {
  uint32_t h = seed;
  const int nblocks = bytes >> 2;
#pragma unroll
  for (int i = 0; i < MAXW; i++) {
    if (i < nblocks) {
      uint32_t k = w[i] * 0xcc9e2d51u;
      k = rotl32(k, 15) * 0x1b873593u;
      h ^= k;
      h = rotl32(h, 13) * 5u + 0xe6546b64u;
    } else if (i == nblocks) {
      uint32_t k = w[i] * 0xcc9e2d51u;
      k = rotl32(k, 15) * 0x1b873593u;
      h ^= k;
    }
  }
  h ^= static_cast<uint32_t>(bytes);
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

// low 64 bits of murmur3_x64_128; MAXQ = number of 64-bit words available (zero padded, and at
// least 2 words beyond the last full 16-byte block)
template <int MAXQ>
__device__ __forceinline__ uint64_t murmur3_128_lo(const uint64_t (&q)[MAXQ], int len, uint32_t seed) {
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = seed, h2 = seed;
  const int nblocks = len >> 4;
#pragma unroll
  for (int i = 0; i < MAXQ / 2; i++) {
    uint64_t k1 = q[2 * i], k2 = q[2 * i + 1];
    if (i < nblocks) {
      k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
      h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
      k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
      h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    } else if (i == nblocks) {
      k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
      k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
  }
  h1 ^= static_cast<uint64_t>(len);
  h2 ^= static_cast<uint64_t>(len);
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2;
  return h1;
}

// ---- calendar (query/functor.cu:70-212) ------------------------------------------------------
enum : int { TB_YEAR, TB_QUARTER, TB_MONTH, TB_DAY_OF_MONTH, TB_DAY_OF_YEAR, TB_MONTH_OF_YEAR, TB_QUARTER_OF_YEAR };

__device__ __forceinline__ uint32_t days_before_month(uint32_t month, bool leap) {
  // cumulative days 0,31,59,... packed as a closed form to stay out of constant memory:
  // the reference's DAYS_BEFORE_MONTH table (query/utils.cu:22-36)
  const uint16_t tbl[13] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334, 365};
  uint32_t d = tbl[month > 12 ? 12 : month];
  if (leap && month >= 2) d++;
  return d;
}

__device__ inline uint32_t resolve_time_bucketizer(int64_t ts, int bucketizer) {
  const int64_t kSecondsPerDay = 86400;
  const int64_t kDays400 = 365 * 400 + 97, kDays100 = 365 * 100 + 24, kDays4 = 365 * 4 + 1;
  const int64_t kAbsoluteZero = -62135596800LL;
  ts -= kAbsoluteZero;
  uint32_t days = static_cast<uint32_t>(ts / kSecondsPerDay);
  int64_t n = days / kDays400;
  uint16_t year = static_cast<uint16_t>(400 * n);
  int64_t start = n * kDays400 * kSecondsPerDay;
  days -= static_cast<uint32_t>(kDays400 * n);
  n = days / kDays100;
  n -= n >> 2;
  year += static_cast<uint16_t>(100 * n);
  start += n * kDays100 * kSecondsPerDay;
  days -= static_cast<uint32_t>(kDays100 * n);
  n = days / kDays4;
  year += static_cast<uint16_t>(4 * n);
  start += n * kDays4 * kSecondsPerDay;
  days -= static_cast<uint32_t>(kDays4 * n);
  n = days / 365;
  n -= n >> 2;
  year += static_cast<uint16_t>(n);
  days -= static_cast<uint32_t>(365 * n);
  start += n * 365 * kSecondsPerDay;
  start += kAbsoluteZero;
  if (bucketizer == TB_YEAR) return static_cast<uint32_t>(start);
  if (bucketizer == TB_DAY_OF_YEAR) return days;
  const uint16_t y1 = static_cast<uint16_t>(year + 1);
  const bool leap = (y1 % 4 == 0) && (y1 % 100 != 0 || y1 % 400 == 0);
  uint32_t month = (days / 31) & 0xffu;
  const uint32_t monthEnd = days_before_month(month + 1, leap);
  if (days >= monthEnd) month++;
  if (bucketizer == TB_MONTH || bucketizer == TB_DAY_OF_MONTH) {
    const uint32_t dbm = days_before_month(month, leap);
    if (bucketizer == TB_MONTH) return static_cast<uint32_t>(start + static_cast<int64_t>(dbm) * kSecondsPerDay);
    return days - dbm;
  }
  if (bucketizer == TB_MONTH_OF_YEAR) return month;
  const uint32_t quarter = month / 3;
  if (bucketizer == TB_QUARTER_OF_YEAR) return quarter;
  return static_cast<uint32_t>(start + static_cast<int64_t>(days_before_month(quarter * 3, leap)) * kSecondsPerDay);
}

__device__ __forceinline__ uint32_t week_start(uint32_t ts) {
  const uint32_t fourDays = 4 * 86400, week = 7 * 86400;
  if (ts < fourDays) return 0;
  return ts - (ts - fourDays) % week;
}

// GetHLLValue (query/functor.hpp:431-466): rho<<16 | register, from the 64-bit hash
__device__ __forceinline__ uint32_t hll_from_hash(uint64_t hashed) {
  const uint32_t group = static_cast<uint32_t>(hashed & ((1u << HLL_BITS) - 1));
  uint32_t rho = 0;
  for (;;) {
    // 32-bit int shift of the original: count wraps mod 32, only the low word is probed
    const uint32_t h = static_cast<uint32_t>(hashed) & (1u << ((rho + HLL_BITS) & 31));
    if (rho + HLL_BITS < 64 && h == 0) rho++;
    else break;
  }
  return rho << 16 | group;
}

// ---- operand loads ------------------------------------------------------------------------------
// physical position of logical row `row` (query/iterator.hpp:209-278)
__device__ __forceinline__ uint32_t locate(const OperandD &op, uint32_t row, const uint32_t *baseCounts,
                                           uint32_t startCount) {
  if (op.mode != 3) return row;
  const uint32_t x = baseCounts ? baseCounts[row] : startCount + row;
  const uint32_t *counts = reinterpret_cast<const uint32_t *>(op.base);
  uint32_t first = 0, last = op.length;
  while (first < last) {
    const uint32_t mid = first + ((last - first) >> 1);
    if (counts[mid] > x) last = mid; else first = mid + 1;
  }
  return first - 1;
}

__device__ __forceinline__ DVal read_stored32(const uint8_t *values, int kind, int step, uint32_t p, uint32_t bit) {
  DVal r;
  r.ok = 1;
  switch (kind) {
    case K_BOOL: r.bits = get_bit(values, bit); break;
    case K_U32:
      r.bits = step == 4 ? reinterpret_cast<const uint32_t *>(values)[p]
               : step == 2 ? reinterpret_cast<const uint16_t *>(values)[p] : values[p];
      break;
    case K_I32:
      r.bits = step == 4 ? reinterpret_cast<const uint32_t *>(values)[p]
               : step == 2 ? static_cast<uint32_t>(static_cast<int32_t>(reinterpret_cast<const int16_t *>(values)[p]))
                           : static_cast<uint32_t>(static_cast<int32_t>(reinterpret_cast<const int8_t *>(values)[p]));
      break;
    default: r.bits = reinterpret_cast<const uint32_t *>(values)[p]; break;
  }
  return r;
}

// value of a 32-bit-kind operand at output position i (row = indexVector[i] for columns)
__device__ __forceinline__ DVal load32(const OperandD &op, uint32_t i, uint32_t row, const uint32_t *baseCounts,
                                       uint32_t startCount) {
  DVal r;
  switch (op.type) {
    case OP_CONST:
      r.bits = op.cbits;
      r.ok = op.cok;
      return r;
    case OP_SCRATCH:
      r.bits = reinterpret_cast<const uint32_t *>(op.base)[i];
      r.ok = op.base[op.nullsOff + i] != 0;
      return r;
    case OP_COLUMN: {
      const uint32_t p = locate(op, row, baseCounts, startCount);
      r = read_stored32(op.base + op.valuesOff, op.kind, op.step, p, p + op.bitOff);
      r.ok = op.mode >= 2 ? get_bit(op.base + op.nullsOff, p + op.bitOff) : 1u;
      return r;
    }
    default: {  // OP_FOREIGN (query/iterator.hpp:911-930)
      const RecordID rid = op.rids[i];
      r.bits = 0;
      r.ok = 0;
      if (rid.batchID != 0 && (rid.batchID - op.baseBatchID < op.numBatches - 1 ||
                               rid.index < static_cast<uint32_t>(op.numRecLast))) {
        const ForeignBatchD b = op.batches[rid.batchID - op.baseBatchID];
        if (b.isConst) {
          r.bits = op.cbits;
          r.ok = op.cok;
          return r;
        }
        const uint32_t p = rid.index;
        r = read_stored32(b.base + b.valuesOff, op.kind, op.step, p, p + b.bitOff);
        r.ok = b.valuesOff != 0 ? get_bit(b.base + b.nullsOff, p + b.bitOff) : 1u;
        if (op.tz) {  // enum -> utc offset (query/iterator.hpp:894-908)
          DVal e = cvt32(r, op.kind, K_I32);
          const int32_t ev = static_cast<int32_t>(e.bits);
          DVal t;
          t.ok = r.ok;
          t.bits = ev < op.tzSize ? static_cast<uint32_t>(static_cast<int32_t>(op.tz[ev])) : 0u;
          r = cvt32(t, K_I32, op.kind);
        }
      }
      return r;
    }
  }
}

// ---- functors --------------------------------------------------------------------------------------
// result kind of a unary functor applied to input kind I
__host__ __device__ __forceinline__ int unary_result_kind(int ft, int I) {
  switch (ft) {
    case Not: case IsNull: case IsNotNull: return K_BOOL;
    case Negate: case Noop: return I;
    default: break;
  }
  if (I == K_F32) return I;  // float specialisation returns its argument
  if (ft >= GetWeekStart && ft <= GetHLLValue) return K_U32;
  return I;  // BitwiseNot and unknown functors
}

__device__ __forceinline__ DVal unary32(int ft, int I, DVal t) {
  DVal r;
  switch (ft) {
    case Not: {
      DVal a = cvt32(t, I, K_BOOL);
      r.ok = a.ok;
      r.bits = a.ok ? (a.bits ^ 1u) : 0u;
      return r;
    }
    case IsNull: r.bits = !t.ok; r.ok = 1; return r;
    case IsNotNull: r.bits = t.ok != 0; r.ok = 1; return r;
    case Noop: return t;
    case Negate:
      if (!t.ok) { r.bits = 0; r.ok = 0; return r; }
      r.ok = 1;
      r.bits = I == K_F32 ? f_bits(-bits_f(t.bits)) : I == K_BOOL ? t.bits : (0u - t.bits);
      return r;
    default: break;
  }
  if (I == K_F32) return t;
  switch (ft) {
    case BitwiseNot:
      if (!t.ok) { r.bits = 0; r.ok = 0; return r; }
      r.ok = 1;
      r.bits = I == K_BOOL ? 1u : ~t.bits;
      return r;
    case GetHLLValue: {
      if (!t.ok) { r.bits = 0; r.ok = 0; return r; }
      uint64_t q[2] = {t.bits, 0};
      r.bits = hll_from_hash(murmur3_128_lo<2>(q, I == K_BOOL ? 1 : 4, 0));
      r.ok = 1;
      return r;
    }
    default: break;
  }
  if (ft >= GetWeekStart && ft <= GetQuarterOfYear) {
    DVal a = cvt32(t, I, K_U32);
    if (!a.ok) { r.bits = 0; r.ok = 0; return r; }
    r.ok = 1;
    switch (ft) {
      case GetWeekStart: r.bits = week_start(a.bits); break;
      case GetMonthStart: r.bits = resolve_time_bucketizer(a.bits, TB_MONTH); break;
      case GetQuarterStart: r.bits = resolve_time_bucketizer(a.bits, TB_QUARTER); break;
      case GetYearStart: r.bits = resolve_time_bucketizer(a.bits, TB_YEAR); break;
      case GetDayOfMonth: r.bits = resolve_time_bucketizer(a.bits, TB_DAY_OF_MONTH); break;
      case GetDayOfYear: r.bits = resolve_time_bucketizer(a.bits, TB_DAY_OF_YEAR); break;
      case GetMonthOfYear: r.bits = resolve_time_bucketizer(a.bits, TB_MONTH_OF_YEAR); break;
      default: r.bits = resolve_time_bucketizer(a.bits, TB_QUARTER_OF_YEAR); break;
    }
    return r;
  }
  return t;
}

__host__ __device__ __forceinline__ int binary_result_kind(int ft, int I) {
  if (ft >= And && ft <= GreaterThanOrEqual) return K_BOOL;
  return I;
}

__device__ __forceinline__ DVal binary32(int ft, int I, DVal a, DVal b) {
  DVal r;
  const uint32_t nul = !(a.ok && b.ok);
  if (ft == And) {
    DVal x = cvt32(a, I, K_BOOL), y = cvt32(b, I, K_BOOL);
    r.ok = !nul;
    r.bits = nul ? 0u : (x.bits & y.bits);
    return r;
  }
  if (ft == Or) {
    DVal x = cvt32(a, I, K_BOOL), y = cvt32(b, I, K_BOOL);
    if ((x.bits && x.ok) || (y.bits && y.ok)) { r.bits = 1; r.ok = 1; return r; }
    r.bits = 0;
    r.ok = !nul;
    return r;
  }
  if (ft >= Equal && ft <= GreaterThanOrEqual) {
    r.ok = !nul;
    bool c;
    if (I == K_F32) {
      const float x = bits_f(a.bits), y = bits_f(b.bits);
      c = ft == Equal ? x == y : ft == NotEqual ? x != y : ft == LessThan ? x < y
          : ft == LessThanOrEqual ? x <= y : ft == GreaterThan ? x > y : x >= y;
    } else if (I == K_I32) {
      const int32_t x = static_cast<int32_t>(a.bits), y = static_cast<int32_t>(b.bits);
      c = ft == Equal ? x == y : ft == NotEqual ? x != y : ft == LessThan ? x < y
          : ft == LessThanOrEqual ? x <= y : ft == GreaterThan ? x > y : x >= y;
    } else {
      const uint32_t x = a.bits, y = b.bits;
      c = ft == Equal ? x == y : ft == NotEqual ? x != y : ft == LessThan ? x < y
          : ft == LessThanOrEqual ? x <= y : ft == GreaterThan ? x > y : x >= y;
    }
    r.bits = nul ? 0u : static_cast<uint32_t>(c);
    return r;
  }
  if (I == K_F32) {
    if (ft < Plus || ft > Divide) return a;  // "return t1"
    if (nul) { r.bits = 0; r.ok = 0; return r; }
    const float x = bits_f(a.bits), y = bits_f(b.bits);
    r.ok = 1;
    r.bits = f_bits(ft == Plus ? x + y : ft == Minus ? x - y : ft == Multiply ? x * y : x / y);
    return r;
  }
  if (ft < Plus || ft > Floor) return a;
  if (nul) { r.bits = 0; r.ok = 0; return r; }
  r.ok = 1;
  const uint32_t ux = a.bits, uy = b.bits;
  switch (ft) {
    case Plus: r.bits = ux + uy; break;
    case Minus: r.bits = ux - uy; break;
    case Multiply: r.bits = ux * uy; break;
    case BitwiseAnd: r.bits = ux & uy; break;
    case BitwiseOr: r.bits = ux | uy; break;
    case BitwiseXor: r.bits = ux ^ uy; break;
    default:
      if (I == K_I32) {
        const int32_t x = static_cast<int32_t>(ux), y = static_cast<int32_t>(uy);
        const int32_t q = y != 0 ? x / y : 0, m = y != 0 ? x % y : 0;  // /0 is UB in the reference
        r.bits = static_cast<uint32_t>(ft == Divide ? q : ft == Mod ? m : x - m);
      } else {
        const uint32_t q = uy != 0 ? ux / uy : 0, m = uy != 0 ? ux % uy : 0;
        r.bits = ft == Divide ? q : ft == Mod ? m : ux - m;
      }
      break;
  }
  return r;
}

// ---- sinks ------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_typed32(uint8_t *dst, int dtype, DVal r, int rk) {
  switch (dtype) {
    case Bool: *dst = static_cast<uint8_t>(cvt32(r, rk, K_BOOL).bits); break;
    case Int8:
      *reinterpret_cast<int8_t *>(dst) = rk == K_F32 ? static_cast<int8_t>(bits_f(r.bits)) : static_cast<int8_t>(r.bits);
      break;
    case Uint8:
      *dst = rk == K_F32 ? static_cast<uint8_t>(bits_f(r.bits)) : static_cast<uint8_t>(r.bits);
      break;
    case Int16:
      *reinterpret_cast<int16_t *>(dst) = rk == K_F32 ? static_cast<int16_t>(bits_f(r.bits)) : static_cast<int16_t>(r.bits);
      break;
    case Uint16:
      *reinterpret_cast<uint16_t *>(dst) = rk == K_F32 ? static_cast<uint16_t>(bits_f(r.bits)) : static_cast<uint16_t>(r.bits);
      break;
    case Int32: *reinterpret_cast<uint32_t *>(dst) = cvt32(r, rk, K_I32).bits; break;
    case Uint32: *reinterpret_cast<uint32_t *>(dst) = cvt32(r, rk, K_U32).bits; break;
    case Float32: *reinterpret_cast<uint32_t *>(dst) = cvt32(r, rk, K_F32).bits; break;
    case Int64:
      *reinterpret_cast<int64_t *>(dst) = rk == K_F32 ? static_cast<int64_t>(bits_f(r.bits))
                                          : rk == K_I32 ? static_cast<int64_t>(static_cast<int32_t>(r.bits))
                                                        : static_cast<int64_t>(r.bits);
      break;
    case UUID: reinterpret_cast<uint64_t *>(dst)[0] = 0; reinterpret_cast<uint64_t *>(dst)[1] = 0; break;
    case GeoPoint: *reinterpret_cast<uint64_t *>(dst) = 0; break;
    default: break;
  }
}

__device__ __forceinline__ double to_double32(DVal r, int rk) {
  return rk == K_F32 ? static_cast<double>(bits_f(r.bits))
         : rk == K_I32 ? static_cast<double>(static_cast<int32_t>(r.bits)) : static_cast<double>(r.bits);
}

// MeasureProxy (query/iterator.hpp:616-647): null -> identity; SUM/AVG scale by the run length
__device__ __forceinline__ void store_measure32(const SinkD &s, uint32_t i, uint32_t row, DVal r, int rk) {
  uint8_t *dst = s.values + static_cast<size_t>(s.width) * i;
  if (!r.ok) {
    if (s.width == 8) *reinterpret_cast<uint64_t *>(dst) = s.identity;
    else *reinterpret_cast<uint32_t *>(dst) = static_cast<uint32_t>(s.identity);
    return;
  }
  const bool isAvg = s.agg == AGGR_AVG_FLOAT;
  const bool scaled = isAvg || (s.agg >= AGGR_SUM_UNSIGNED && s.agg <= AGGR_SUM_FLOAT);
  uint32_t count = 1;
  if (scaled && s.baseCounts) count = s.baseCounts[row + 1] - s.baseCounts[row];
  if (isAvg) {
    float f;
    switch (s.dtype) {
      case Float64: f = static_cast<float>(to_double32(r, rk)); break;
      case Int64: f = rk == K_F32 ? static_cast<float>(static_cast<int64_t>(bits_f(r.bits)))
                      : rk == K_I32 ? static_cast<float>(static_cast<int64_t>(static_cast<int32_t>(r.bits)))
                                    : static_cast<float>(static_cast<int64_t>(r.bits));
        break;
      case Int32: f = static_cast<float>(static_cast<int32_t>(cvt32(r, rk, K_I32).bits)); break;
      case Uint32: f = static_cast<float>(cvt32(r, rk, K_U32).bits); break;
      default: f = bits_f(cvt32(r, rk, K_F32).bits); break;
    }
    reinterpret_cast<uint32_t *>(dst)[0] = f_bits(f);
    reinterpret_cast<uint32_t *>(dst)[1] = count;
    return;
  }
  switch (s.dtype) {
    case Int32: *reinterpret_cast<uint32_t *>(dst) = cvt32(r, rk, K_I32).bits * count; break;
    case Uint32: *reinterpret_cast<uint32_t *>(dst) = cvt32(r, rk, K_U32).bits * count; break;
    case Float32: *reinterpret_cast<float *>(dst) = bits_f(cvt32(r, rk, K_F32).bits) * static_cast<float>(count); break;
    case Int64: {
      const int64_t v = rk == K_F32 ? static_cast<int64_t>(bits_f(r.bits))
                        : rk == K_I32 ? static_cast<int64_t>(static_cast<int32_t>(r.bits))
                                      : static_cast<int64_t>(r.bits);
      *reinterpret_cast<uint64_t *>(dst) = static_cast<uint64_t>(v) * static_cast<uint64_t>(count);
      break;
    }
    default: *reinterpret_cast<double *>(dst) = to_double32(r, rk) * static_cast<double>(count); break;  // Float64
  }
}

// writes result r (of kind rk) for output position i; `row` = indexVector[i]
__device__ __forceinline__ void sink_store32(const SinkD &s, uint32_t i, uint32_t row, DVal r, int rk) {
  switch (s.type) {
    case SINK_PRED: s.values[i] = static_cast<uint8_t>(cvt32(r, rk, K_BOOL).bits); break;
    case SINK_MEASURE: store_measure32(s, i, row, r, rk); break;
    default:
      store_typed32(s.values + static_cast<size_t>(s.width) * i, s.dtype, r, rk);
      // a 32-bit value can never become a UUID / GeoPoint: the reference yields (zero, null)
      // (query/functor.hpp:720-735, :836-881)
      s.nulls[i] = (r.ok && s.dtype != UUID && s.dtype != GeoPoint) ? 1 : 0;
      break;
  }
}

}  // namespace ares
