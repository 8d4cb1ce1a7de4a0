// Aggregation operators shared by HashReduce and Reduce: how two partial values combine
// (reference query/sort_reduce.cu:170-216, query/hash_reduction.cu:346-391,
// query/concurrent_unordered_map.hpp:35-76, RollingAvgFunctor query/functor.hpp:1414-1436).
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <climits>
#include <cstring>
#include <stdexcept>

#include "ares_algorithm.h"
#include "device_model.hpp"

namespace ares {

enum : int { V_U32, V_I32, V_F32, V_U64, V_I64, V_F64, V_AVG };
enum : int { OP_SUM, OP_MIN, OP_MAX, OP_AVG };

struct AggSpec {
  int vtype;
  int op;
  int width;          // bytes per value
  uint64_t identity;  // initial slot value (bit pattern)
};

// `identity` is the operator's true neutral element (what a group slot starts from before
// partials are merged into it), not the measure-transform identity of query/utils.hpp:169-184.
inline AggSpec make_agg_spec(int aggFunc, int valueBytes) {
  AggSpec a;
  a.identity = 0;
  switch (aggFunc) {
    case AGGR_SUM_UNSIGNED: a.op = OP_SUM; a.vtype = valueBytes == 4 ? V_U32 : V_U64; break;
    case AGGR_SUM_SIGNED: a.op = OP_SUM; a.vtype = valueBytes == 4 ? V_I32 : V_I64; break;
    case AGGR_SUM_FLOAT: a.op = OP_SUM; a.vtype = valueBytes == 4 ? V_F32 : V_F64; break;
    case AGGR_MIN_UNSIGNED: a.op = OP_MIN; a.vtype = V_U32; a.identity = UINT32_MAX; break;
    case AGGR_MIN_SIGNED: a.op = OP_MIN; a.vtype = V_I32; a.identity = static_cast<uint32_t>(INT32_MAX); break;
    case AGGR_MIN_FLOAT: { a.op = OP_MIN; a.vtype = V_F32; float f = INFINITY; memcpy(&a.identity, &f, 4); break; }
    case AGGR_MAX_UNSIGNED: a.op = OP_MAX; a.vtype = V_U32; a.identity = 0; break;
    case AGGR_MAX_SIGNED: a.op = OP_MAX; a.vtype = V_I32; a.identity = static_cast<uint32_t>(INT32_MIN); break;
    case AGGR_MAX_FLOAT: { a.op = OP_MAX; a.vtype = V_F32; float f = -INFINITY; memcpy(&a.identity, &f, 4); break; }
    case AGGR_AVG_FLOAT: a.op = OP_AVG; a.vtype = V_AVG; break;
    default: throw std::invalid_argument("Unsupported aggregation function type");
  }
  a.width = (a.vtype == V_U32 || a.vtype == V_I32 || a.vtype == V_F32) ? 4 : 8;
  return a;
}

// RollingAvgFunctor (query/functor.hpp:1414-1436) on packed {float avg, u32 count}
__device__ __forceinline__ uint64_t rolling_avg(uint64_t lhs, uint64_t rhs) {
  const uint32_t lc = static_cast<uint32_t>(lhs >> 32), rc = static_cast<uint32_t>(rhs >> 32);
  const uint32_t total = lc + rc;
  if (total == 0) return 0;
  const float f = bits_f(static_cast<uint32_t>(lhs)) / total * lc + bits_f(static_cast<uint32_t>(rhs)) / total * rc;
  return (static_cast<uint64_t>(total) << 32) | f_bits(f);
}

__device__ __forceinline__ void aggregate_slot(uint8_t *slotValue, const uint8_t *v, const AggSpec &a) {
  switch (a.vtype) {
    case V_U32: {
      const uint32_t x = *reinterpret_cast<const uint32_t *>(v);
      uint32_t *p = reinterpret_cast<uint32_t *>(slotValue);
      if (a.op == OP_SUM) atomicAdd(p, x); else if (a.op == OP_MIN) atomicMin(p, x); else atomicMax(p, x);
      break;
    }
    case V_I32: {
      const int32_t x = *reinterpret_cast<const int32_t *>(v);
      int32_t *p = reinterpret_cast<int32_t *>(slotValue);
      if (a.op == OP_SUM) atomicAdd(p, x); else if (a.op == OP_MIN) atomicMin(p, x); else atomicMax(p, x);
      break;
    }
    case V_F32: {
      const float x = *reinterpret_cast<const float *>(v);
      float *p = reinterpret_cast<float *>(slotValue);
      if (a.op == OP_SUM) {
        atomicAdd(p, x);
      } else {  // (new < old ? new : old) / (new > old ? new : old), concurrent_unordered_map.hpp:35-57
        uint32_t *pu = reinterpret_cast<uint32_t *>(slotValue);
        uint32_t old = __hip_atomic_load(pu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
          const float o = bits_f(old);
          const bool take = a.op == OP_MIN ? (x < o) : (x > o);
          if (!take) break;
          const uint32_t prev = atomicCAS(pu, old, f_bits(x));
          if (prev == old) break;
          old = prev;
        }
      }
      break;
    }
    case V_U64: case V_I64:
      atomicAdd(reinterpret_cast<unsigned long long *>(slotValue), *reinterpret_cast<const unsigned long long *>(v));
      break;
    case V_F64:
      atomicAdd(reinterpret_cast<double *>(slotValue), *reinterpret_cast<const double *>(v));
      break;
    default: {  // V_AVG
      const uint64_t x = *reinterpret_cast<const uint64_t *>(v);
      unsigned long long *p = reinterpret_cast<unsigned long long *>(slotValue);
      unsigned long long old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        const unsigned long long prev = atomicCAS(p, old, rolling_avg(old, x));
        if (prev == old) break;
        old = prev;
      }
      break;
    }
  }
}

// value bits (zero-extended to 64) of element i of a value vector
__device__ __forceinline__ uint64_t load_value_bits(const uint8_t *values, const AggSpec &a, size_t i) {
  return a.width == 8 ? reinterpret_cast<const uint64_t *>(values)[i]
                      : static_cast<uint64_t>(reinterpret_cast<const uint32_t *>(values)[i]);
}

// op(earlier, later) on bit patterns, exactly as the reference's binary functors
__device__ __forceinline__ uint64_t combine_bits(const AggSpec &a, uint64_t x, uint64_t y) {
  switch (a.vtype) {
    case V_U32: {
      const uint32_t p = static_cast<uint32_t>(x), q = static_cast<uint32_t>(y);
      return a.op == OP_SUM ? static_cast<uint32_t>(p + q) : a.op == OP_MIN ? (q < p ? q : p) : (p < q ? q : p);
    }
    case V_I32: {
      const int32_t p = static_cast<int32_t>(x), q = static_cast<int32_t>(y);
      const int32_t r = a.op == OP_SUM ? static_cast<int32_t>(static_cast<uint32_t>(p) + static_cast<uint32_t>(q))
                        : a.op == OP_MIN ? (q < p ? q : p) : (p < q ? q : p);
      return static_cast<uint32_t>(r);
    }
    case V_F32: {
      const float p = bits_f(static_cast<uint32_t>(x)), q = bits_f(static_cast<uint32_t>(y));
      const float r = a.op == OP_SUM ? p + q : a.op == OP_MIN ? (q < p ? q : p) : (p < q ? q : p);
      return f_bits(r);
    }
    case V_U64: case V_I64: return x + y;
    case V_F64: return __double_as_longlong(__longlong_as_double(x) + __longlong_as_double(y));
    default: return rolling_avg(x, y);
  }
}

__device__ __forceinline__ void store_value_bits(uint8_t *values, const AggSpec &a, size_t i, uint64_t bits) {
  if (a.width == 8) reinterpret_cast<uint64_t *>(values)[i] = bits;
  else reinterpret_cast<uint32_t *>(values)[i] = static_cast<uint32_t>(bits);
}

}  // namespace ares
