// Shared by the fast transform / filter kernels (transform.hip) and the fused scan of
// hash_reduce_lds.hip: the "4-byte column (x constant)" operand shape, its evaluation and the host
// code that recognises it.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "binding.hpp"
#include "common.hpp"
#include "device_model.hpp"

namespace ares {

// generic (descriptor-driven) call parameters of one Unary/Binary Transform/Filter call
struct EvalParams {
  OperandD a, b;
  int arity;
  int functor;
  int I;   // common input kind
  int rk;  // result kind
  const uint32_t *idx;
  const uint32_t *baseCounts;
  uint32_t startCount;
  int needRow;
};

// gfx950 global accesses only need dword (or, for these packed forms, byte) alignment
struct __attribute__((aligned(4))) U32x4 { uint32_t v[4]; };
struct __attribute__((aligned(4))) U64x2 { uint64_t v[2]; };

// 32-bit column operand + optional constant second operand, as the fast kernels see them
struct FastOperands {
  const uint32_t *vals;  // column values
  const uint8_t *nulls;  // validity bitmap or nullptr (mode 1)
  uint32_t bitOff;       // bit position of row 0 in the bitmap
  int akind;             // stored kind of the column
  int arity, functor, I, rk;
  int bkind;
  uint32_t bbits, bok;   // constant second operand
  const uint32_t *idx;   // index vector or nullptr (identity)
  int pad;
  int divLike;           // integer Divide / Mod / Floor by the constant: multiply-high division
  int debug;             // ARES_F_DEBUG: timing experiments only
  int step;              // bytes per stored value: 4, or 2 / 1 (Int16 / Uint16 / Int8 / Uint8 / SmallEnum / BigEnum columns,
                         // widened in registers — sign-extended for akind K_I32 — as query/iterator.hpp:146-165 does)
};
// bytes of `rows` values of the operand's column
__host__ __device__ inline uint64_t fast_value_bytes(const FastOperands &f, uint64_t rows) { return static_cast<uint64_t>(f.step ? f.step : 4) * rows; }

struct __attribute__((packed, aligned(1))) PU16 { uint16_t v; };

// x / d and x % d for a divisor that is the same for the whole launch: one multiply-high by
// M = floor(2^32 / d) estimates the quotient to within one (M = 0 marks d < 2).
struct FastDivisor {
  uint32_t d, M;
};
__device__ __forceinline__ FastDivisor make_fast_divisor(uint32_t d) {
  FastDivisor r;
  r.d = d;
  r.M = d >= 2 ? static_cast<uint32_t>((1ull << 32) / d) : 0u;
  return r;
}
__device__ __forceinline__ void fast_divmod(const FastDivisor &fd, uint32_t x, uint32_t &q, uint32_t &r) {
  if (fd.d < 2) {  // 0: the reference divides by zero (undefined); binary32 yields q = r = 0 — keep that
    q = fd.d ? x : 0u;
    r = 0u;
    return;
  }
  q = __umulhi(x, fd.M);
  r = x - q * fd.d;
  if (r >= fd.d) { q++; r -= fd.d; }
}

// The fast kernels inline this once per element (16-32 copies), so it only covers what the hot
// queries use: unary Noop (a bare column) and the binary functors; the calendar / HLL / logical
// unary functors stay on the generic kernels.
__device__ __forceinline__ DVal eval_fast(const FastOperands &f, uint32_t bits, uint32_t ok, DVal y,
                                          const FastDivisor &fd) {
  DVal x;
  x.bits = bits;
  x.ok = ok;
  x = cvt32(x, f.akind, f.I);
  if (f.arity == 1) return x;
  if (f.divLike) {  // Divide / Mod / Floor on integers by the launch-wide constant (functor.hpp:337-351)
    DVal r;
    r.ok = x.ok && y.ok;
    r.bits = 0;
    if (r.ok) {
      uint32_t q, m;
      if (f.I == K_I32) {  // C++ truncating semantics on magnitudes
        const int32_t sx = static_cast<int32_t>(x.bits), sy = static_cast<int32_t>(y.bits);
        const uint32_t ax = sx < 0 ? 0u - x.bits : x.bits;
        fast_divmod(fd, ax, q, m);
        const uint32_t sq = ((sx < 0) != (sy < 0)) ? 0u - q : q;
        const uint32_t sm = sx < 0 ? 0u - m : m;
        r.bits = f.functor == Divide ? sq : f.functor == Mod ? sm : x.bits - sm;
      } else {
        fast_divmod(fd, x.bits, q, m);
        r.bits = f.functor == Divide ? q : f.functor == Mod ? m : x.bits - m;
      }
    }
    return r;
  }
  return binary32(f.functor, f.I, x, y);
}

// comparison functors only (what a filter root is in practice): value of (x ft y), null -> false
__device__ __forceinline__ uint32_t compare_fast(const FastOperands &f, uint32_t bits, uint32_t ok, DVal y) {
  DVal x;
  x.bits = bits;
  x.ok = ok;
  x = cvt32(x, f.akind, f.I);
  const int ft = f.functor;
  bool c;
  if (f.I == K_F32) {
    const float a = bits_f(x.bits), b = bits_f(y.bits);
    c = ft == Equal ? a == b : ft == NotEqual ? a != b : ft == LessThan ? a < b
        : ft == LessThanOrEqual ? a <= b : ft == GreaterThan ? a > b : a >= b;
  } else if (f.I == K_I32) {
    const int32_t a = static_cast<int32_t>(x.bits), b = static_cast<int32_t>(y.bits);
    c = ft == Equal ? a == b : ft == NotEqual ? a != b : ft == LessThan ? a < b
        : ft == LessThanOrEqual ? a <= b : ft == GreaterThan ? a > b : a >= b;
  } else {
    const uint32_t a = x.bits, b = y.bits;
    c = ft == Equal ? a == b : ft == NotEqual ? a != b : ft == LessThan ? a < b
        : ft == LessThanOrEqual ? a <= b : ft == GreaterThan ? a > b : a >= b;
  }
  return (x.ok && y.ok && c) ? 1u : 0u;
}


// Four elements of one expression with ONE dispatch on its shape: bare column, integer
// divide / modulo / floor by the constant, integer add / subtract / multiply; anything else goes
// through eval_fast element by element.  r[j] = result bits (kind f.rk), return = validity nibble.
// Bit-identical to eval_fast: a null operand of a binary functor yields bits 0, of a bare column
// the stored bits (query/functor.hpp:337-351, :660-697).
__device__ __forceinline__ uint32_t eval_quad(const FastOperands &f, const uint32_t (&v)[4], uint32_t ok, DVal y,
                                              const FastDivisor &fd, uint32_t (&r)[4]) {
  const bool intKinds = f.akind != K_F32 && f.I != K_F32 && f.akind != K_BOOL;
  if (f.arity == 1 && (f.akind == f.I || intKinds)) {
#pragma unroll
    for (int j = 0; j < 4; j++) r[j] = v[j];
    return ok;
  }
  if (f.arity == 2 && intKinds && y.ok) {
    if (f.divLike) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t q, m;
        if (f.I == K_I32) {
          const int32_t sx = static_cast<int32_t>(v[j]), sy = static_cast<int32_t>(y.bits);
          fast_divmod(fd, sx < 0 ? 0u - v[j] : v[j], q, m);
          const uint32_t sq = ((sx < 0) != (sy < 0)) ? 0u - q : q;
          const uint32_t sm = sx < 0 ? 0u - m : m;
          r[j] = f.functor == Divide ? sq : f.functor == Mod ? sm : v[j] - sm;
        } else {
          fast_divmod(fd, v[j], q, m);
          r[j] = f.functor == Divide ? q : f.functor == Mod ? m : v[j] - m;
        }
        if (!((ok >> j) & 1u)) r[j] = 0;
      }
      return ok;
    }
    if (f.functor == Plus || f.functor == Minus || f.functor == Multiply) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t x = f.functor == Plus ? v[j] + y.bits : f.functor == Minus ? v[j] - y.bits : v[j] * y.bits;
        r[j] = ((ok >> j) & 1u) ? x : 0u;
      }
      return ok;
    }
  }
  uint32_t outOk = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const DVal x = eval_fast(f, v[j], (ok >> j) & 1u, y, fd);
    r[j] = x.bits;
    outOk |= (x.ok ? 1u : 0u) << j;
  }
  return outOk;
}

// The same comparison for the QUADS x 4 elements of a tile with ONE dispatch on (kind, functor): the
// per-element form above costs a chain of scalar branches per element, which is what bounds the
// streaming filter kernels once the memory system keeps up.  kb[q] = 4 keep bits of quad q;
// in[q] = which of the quad's positions exist.
template <int QUADS>
__device__ __forceinline__ void compare_tile(const FastOperands &f, const uint32_t (&vals)[QUADS][4], const uint32_t (&okb)[QUADS],
                                             const uint32_t (&in)[QUADS], DVal y, uint32_t (&kb)[QUADS]) {
  const bool sameBits = f.akind == f.I || (f.akind != K_F32 && f.I != K_F32 && f.akind != K_BOOL);
  if (!sameBits || !y.ok) {  // value conversion needed (or a null constant): element-wise form
#pragma unroll
    for (int q = 0; q < QUADS; q++) {
      kb[q] = 0;
#pragma unroll
      for (int j = 0; j < 4; j++)
        if ((in[q] >> j) & 1u) kb[q] |= compare_fast(f, vals[q][j], (okb[q] >> j) & 1u, y) << j;
    }
    return;
  }
#define ARES_CMP_LOOP(T, CONV, OP)                                                     \
  _Pragma("unroll") for (int q = 0; q < QUADS; q++) {                                  \
    uint32_t m = 0;                                                                    \
    _Pragma("unroll") for (int j = 0; j < 4; j++) m |= (CONV(vals[q][j]) OP b ? 1u : 0u) << j; \
    kb[q] = m & okb[q] & in[q];                                                        \
  }
#define ARES_CMP_TYPE(T, CONV)                        \
  {                                                   \
    const T b = CONV(y.bits);                         \
    switch (f.functor) {                              \
      case Equal: ARES_CMP_LOOP(T, CONV, ==) break;   \
      case NotEqual: ARES_CMP_LOOP(T, CONV, !=) break; \
      case LessThan: ARES_CMP_LOOP(T, CONV, <) break; \
      case LessThanOrEqual: ARES_CMP_LOOP(T, CONV, <=) break; \
      case GreaterThan: ARES_CMP_LOOP(T, CONV, >) break; \
      default: ARES_CMP_LOOP(T, CONV, >=) break;      \
    }                                                 \
  }
  if (f.I == K_F32) ARES_CMP_TYPE(float, bits_f)
  else if (f.I == K_I32) ARES_CMP_TYPE(int32_t, static_cast<int32_t>)
  else ARES_CMP_TYPE(uint32_t, static_cast<uint32_t>)
#undef ARES_CMP_TYPE
#undef ARES_CMP_LOOP
}

// ---- host side ------------------------------------------------------------------------------------
inline void build_params(const InputVector *ins, int arity, hipStream_t stream, const uint32_t *indexVector,
                         const uint32_t *baseCounts, uint32_t startCount, int functor, EvalParams &p,
                         CallTemps &temps) {
  memset(&p, 0, sizeof(p));
  bind_operand(ins[0], true, stream, p.a, temps);
  p.arity = arity;
  p.functor = functor;
  if (arity == 2) {
    bind_operand(ins[1], false, stream, p.b, temps);
    check_binary_kinds(p.a, p.b, ins[1]);
    p.I = common_kind(p.a.kind, p.b.kind);
    p.rk = is_wide(p.I) ? K_BOOL : binary_result_kind(functor, p.I);
  } else {
    p.I = p.a.kind;
    p.rk = is_wide(p.I) ? K_NONE : unary_result_kind(functor, p.I);
  }
  p.idx = indexVector;
  p.baseCounts = baseCounts;
  p.startCount = startCount;
  p.needRow = indexVector != nullptr && (p.a.type == OP_COLUMN || (arity == 2 && p.b.type == OP_COLUMN));
}


// The hot shape of a live-batch query: a 1-, 2- or 4-byte column (modes 1/2) of a 32-bit kind, optionally combined
// with a constant, feeding a dimension / scratch vector or a measure.  Everything else takes the generic kernels.
inline bool fast_operands(const EvalParams &p, FastOperands &f, bool compareOnly) {
  if (p.arity == 1 && (compareOnly || p.functor != Noop)) return false;
  if (compareOnly && (p.functor < Equal || p.functor > GreaterThanOrEqual)) return false;
  if (p.a.type != OP_COLUMN || p.a.mode > 2 || p.a.kind == K_BOOL || is_wide(p.a.kind)) return false;
  if (p.a.step != 4 && !((p.a.step == 2 || p.a.step == 1) && (p.a.kind == K_I32 || p.a.kind == K_U32))) return false;
  if (p.arity == 2 && (p.b.type != OP_CONST || is_wide(p.b.kind))) return false;
  if (is_wide(p.I)) return false;
  memset(&f, 0, sizeof(f));
  f.vals = reinterpret_cast<const uint32_t *>(p.a.base + p.a.valuesOff);
  f.nulls = p.a.mode == 2 ? p.a.base + p.a.nullsOff : nullptr;
  f.bitOff = p.a.bitOff;
  f.akind = p.a.kind;
  f.arity = p.arity;
  f.functor = p.functor;
  f.I = p.I;
  f.rk = p.rk;
  f.bkind = p.arity == 2 ? p.b.kind : p.I;
  f.bbits = p.b.cbits;
  f.bok = p.b.cok;
  f.idx = p.needRow ? p.idx : nullptr;
  f.divLike = p.arity == 2 && (p.I == K_I32 || p.I == K_U32) && (p.functor == Divide || p.functor == Mod || p.functor == Floor);
  static const int debug = [] {  // kernel-variant switch of the filter experiments (tools/)
    const char *dbg = getenv("ARES_F_DEBUG");
    return dbg ? atoi(dbg) : 0;
  }();
  f.debug = debug;
  f.step = p.a.step;
  return true;
}


}  // namespace ares
