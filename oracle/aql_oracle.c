/* aql_oracle.c — CPU restatement of the AresDB AQL batch-execution path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP library in
 * aresdb_amd/csrc: a scalar, single-threaded, plain-C restatement of what the reference's
 * libalgorithm.so computes in QUERY_MODE=HOST, behind the very same C ABI
 * (include/ares_algorithm.h).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product never does.
 *
 * Pinning: tests/test_oracle_vs_reference.py drives this library and the reference's own
 * HOST build (oracle/_ref, built by oracle/Makefile.ref from the unmodified sources) with the
 * same buffers and requires bit-identical outputs; tests/test_golden_vectors.py replays the
 * known-answer vectors of the reference's gtest suite (query/algorithm_unittest.cu,
 * iterator_unittest.cu, functor_unittest.cu) against it.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference root).  All pointers are host pointers; cudaStream/device are ignored.
 */
#include <float.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ares_algorithm.h"

/* ------------------------------------------------------------------------------------------
 * murmur3 — query/utils.cu:113-155 (x86_32) and :157-241 (x64_128)
 * ---------------------------------------------------------------------------------------- */
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

uint32_t oracle_murmur3_32(const uint8_t *key, int bytes, uint32_t seed) {
  uint32_t h = seed;
  int nblocks = bytes / 4;
  for (int i = 0; i < nblocks; i++) {
    uint32_t k;
    memcpy(&k, key + 4 * i, 4);
    k *= 0xcc9e2d51u;
    k = rotl32(k, 15);
    k *= 0x1b873593u;
    h ^= k;
    h = rotl32(h, 13);
    h = h * 5 + 0xe6546b64u;
  }
  const uint8_t *tail = key + 4 * nblocks;
  uint32_t k = 0;
  switch (bytes & 3) {
    case 3: k ^= (uint32_t)tail[2] << 16; /* fallthrough */
    case 2: k ^= (uint32_t)tail[1] << 8;  /* fallthrough */
    case 1:
      k ^= (uint32_t)tail[0];
      k *= 0xcc9e2d51u;
      k = rotl32(k, 15);
      k *= 0x1b873593u;
      h ^= k;
  }
  h ^= (uint32_t)bytes;
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

static uint64_t fmix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

void oracle_murmur3_128(const uint8_t *key, int len, uint32_t seed, uint64_t out[2]) {
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = seed, h2 = seed;
  int nblocks = len / 16;
  for (int i = 0; i < nblocks; i++) {
    uint64_t k1, k2;
    memcpy(&k1, key + 16 * i, 8);
    memcpy(&k2, key + 16 * i + 8, 8);
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  }
  const uint8_t *tail = key + 16 * nblocks;
  uint64_t k1 = 0, k2 = 0;
  int t = len & 15;
  for (int i = t - 1; i >= 8; i--) k2 ^= (uint64_t)tail[i] << (8 * (i - 8));
  if (t > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
  for (int i = (t > 8 ? 8 : t) - 1; i >= 0; i--) k1 ^= (uint64_t)tail[i] << (8 * i);
  if (t > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }
  h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
  h1 += h2; h2 += h1;
  h1 = fmix64(h1); h2 = fmix64(h2);
  h1 += h2; h2 += h1;
  out[0] = h1; out[1] = h2;
}

/* ------------------------------------------------------------------------------------------
 * calendar — query/functor.cu:70-212
 * ---------------------------------------------------------------------------------------- */
enum { TB_YEAR, TB_QUARTER, TB_MONTH, TB_DAY_OF_MONTH, TB_DAY_OF_YEAR, TB_MONTH_OF_YEAR,
       TB_QUARTER_OF_YEAR };
static const uint16_t DAYS_BEFORE_MONTH[13] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304,
                                               334, 365};
#define SECONDS_PER_DAY 86400
#define DAYS_PER_400_YEARS (365 * 400 + 97)
#define DAYS_PER_100_YEARS (365 * 100 + 24)
#define DAYS_PER_4_YEARS (365 * 4 + 1)
#define ABSOLUTE_ZERO_TS (-62135596800LL)

static uint16_t days_before_month(uint8_t month, bool leap) {
  uint16_t d = DAYS_BEFORE_MONTH[month];
  if (leap && month >= 2) d++;
  return d;
}

/* functor.cu:70-161 — note the deliberately narrow intermediate types of the original
 * (uint32 days, uint16 year, uint8 month): they are part of the observable behaviour. */
uint32_t oracle_resolve_time_bucketizer(int64_t ts, int bucketizer) {
  ts -= ABSOLUTE_ZERO_TS;
  uint32_t days = (uint32_t)(ts / SECONDS_PER_DAY);
  int64_t n = days / DAYS_PER_400_YEARS;
  uint16_t year = (uint16_t)(400 * n);
  int64_t start = n * DAYS_PER_400_YEARS * SECONDS_PER_DAY;
  days -= (uint32_t)(DAYS_PER_400_YEARS * n);
  n = days / DAYS_PER_100_YEARS;
  n -= n >> 2;
  year += (uint16_t)(100 * n);
  start += n * DAYS_PER_100_YEARS * SECONDS_PER_DAY;
  days -= (uint32_t)(DAYS_PER_100_YEARS * n);
  n = days / DAYS_PER_4_YEARS;
  year += (uint16_t)(4 * n);
  start += n * DAYS_PER_4_YEARS * SECONDS_PER_DAY;
  days -= (uint32_t)(DAYS_PER_4_YEARS * n);
  n = days / 365;
  n -= n >> 2;
  year += (uint16_t)n;
  days -= (uint32_t)(365 * n);
  start += n * 365 * SECONDS_PER_DAY;
  start += ABSOLUTE_ZERO_TS;
  if (bucketizer == TB_YEAR) return (uint32_t)start;
  if (bucketizer == TB_DAY_OF_YEAR) return days;
  uint16_t y1 = (uint16_t)(year + 1);
  bool leap = (y1 % 4 == 0) && (y1 % 100 != 0 || y1 % 400 == 0);
  uint8_t month = (uint8_t)(days / 31);
  uint16_t monthEnd = days_before_month((uint8_t)(month + 1), leap);
  if (days >= monthEnd) month++;
  if (bucketizer == TB_MONTH || bucketizer == TB_DAY_OF_MONTH) {
    uint32_t dbm = days_before_month(month, leap);
    if (bucketizer == TB_MONTH) return (uint32_t)(start + (int64_t)dbm * SECONDS_PER_DAY);
    return days - dbm;
  }
  if (bucketizer == TB_MONTH_OF_YEAR) return month;
  int quarter = month / 3;
  if (bucketizer == TB_QUARTER_OF_YEAR) return (uint32_t)quarter;
  return (uint32_t)(start + (int64_t)days_before_month((uint8_t)(quarter * 3), leap) *
                                SECONDS_PER_DAY);
}

/* functor.cu:206-212 */
uint32_t oracle_week_start(uint32_t ts) {
  const uint32_t fourDays = 4 * SECONDS_PER_DAY, week = 7 * SECONDS_PER_DAY;
  if (ts < fourDays) return 0;
  return ts - (ts - fourDays) % week;
}

/* ------------------------------------------------------------------------------------------
 * value model: one (value, validity) pair of the reference's thrust::tuple<T,bool>
 * ---------------------------------------------------------------------------------------- */
typedef enum { K_BOOL, K_I32, K_U32, K_F32, K_I64, K_UUID, K_GEO, K_NONE } Kind;

typedef struct {
  Kind k;
  bool ok;
  union {
    bool b;
    int32_t i;
    uint32_t u;
    float f;
    int64_t l;
    UUIDT uuid;
    GeoPointT geo;
  } v;
} Val;

typedef struct {
  const char *err; /* first error, NULL when fine */
} Err;

static char *dup_err(const char *s) { return strdup(s); }

/* storage kind of a column data type — query/binder.hpp:235-258, :340-346 */
static Kind kind_of_datatype(enum DataType t) {
  switch (t) {
    case Bool: return K_BOOL;
    case Int8: case Int16: case Int32: return K_I32;
    case Uint8: case Uint16: case Uint32: return K_U32;
    case Float32: return K_F32;
    case Int64: return K_I64;
    case UUID: return K_UUID;
    case GeoPoint: return K_GEO;
    default: return K_NONE;
  }
}

/* query/utils.hpp:186-204 */
static int step_in_bytes(enum DataType t) {
  switch (t) {
    case Bool: case Int8: case Uint8: return 1;
    case Int16: case Uint16: return 2;
    case Int32: case Uint32: case Float32: return 4;
    case GeoPoint: case Int64: case Uint64: return 8;
    case UUID: return 16;
    default: return -1;
  }
}

static bool get_bit(const uint8_t *p, uint32_t i) { return (p[i / 8] >> (i % 8)) & 1; }

/* query/utils.hpp:83-94 common_type + :115-126 input_iterator_value_type */
static Kind common_kind(Kind a, Kind b) {
  if (a == K_GEO || a == K_UUID) return a;
  if (a == K_F32 || b == K_F32) return K_F32;
  if (a == K_I64 || b == K_I64) return K_I64;
  if (a == K_I32 || b == K_I32) return K_I32;
  return K_U32;
}

/* implicit thrust::tuple<A,bool> -> tuple<B,bool> conversion == static_cast<B>(A) */
static Val convert(Val x, Kind to) {
  Val r;
  memset(&r, 0, sizeof(r));
  r.k = to;
  r.ok = x.ok;
  if (x.k == to) return x;
  /* widen everything through (int64 | double) views */
  double d = 0; int64_t l = 0; bool isf = false;
  switch (x.k) {
    case K_BOOL: l = x.v.b ? 1 : 0; break;
    case K_I32: l = x.v.i; break;
    case K_U32: l = x.v.u; break;
    case K_I64: l = x.v.l; break;
    case K_F32: d = x.v.f; isf = true; break;
    default: return r; /* uuid/geo never convert */
  }
  switch (to) {
    case K_BOOL: r.v.b = isf ? (x.v.f != 0.0f) : (l != 0); break;
    case K_I32: r.v.i = isf ? (int32_t)x.v.f : (int32_t)l; break;
    case K_U32: r.v.u = isf ? (uint32_t)x.v.f : (uint32_t)l; break;
    case K_I64: r.v.l = isf ? (int64_t)x.v.f : l; break;
    case K_F32:
      if (isf) r.v.f = (float)d;
      else if (x.k == K_U32) r.v.f = (float)x.v.u;
      else if (x.k == K_I64) r.v.f = (float)x.v.l;
      else r.v.f = (float)(int32_t)l;
      break;
    default: break;
  }
  return r;
}

/* ------------------------------------------------------------------------------------------
 * operands — query/iterator.hpp:62-289 (VectorPartyIterator), :465-537 (SimpleIterator),
 * :845-931 (RecordIDJoinIterator); binding rules query/binder.hpp:102-264, :308-426
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const InputVector *in;
  const uint32_t *indexVector;
  const uint32_t *baseCounts;
  uint32_t startCount;
  Kind kind; /* storage kind */
} Operand;

static const char *bind_operand(Operand *op, const InputVector *in, const uint32_t *indexVector,
                                const uint32_t *baseCounts, uint32_t startCount, bool first) {
  op->in = in;
  op->indexVector = indexVector;
  op->baseCounts = baseCounts;
  op->startCount = startCount;
  switch (in->Type) {
    case ConstantInput:
      switch (in->Vector.Constant.DataType) {
        case ConstInt: op->kind = K_I32; return NULL;
        case ConstFloat: op->kind = K_F32; return NULL;
        case ConstGeoPoint: op->kind = K_GEO; return NULL;
        case ConstUUID: op->kind = K_UUID; return NULL;
      }
      return "Unsupported constant type";
    case ScratchSpaceInput:
      switch (in->Vector.ScratchSpace.DataType) {
        case Int32: op->kind = K_I32; return NULL;
        case Uint32: op->kind = K_U32; return NULL;
        case Float32: op->kind = K_F32; return NULL;
        case UUID: op->kind = K_UUID; return NULL;
        case GeoPoint: op->kind = K_GEO; return NULL;
        default: return "Unsupported data type for ScratchSpaceInput";
      }
    case VectorPartyInput: {
      Kind k = kind_of_datatype(in->Vector.VP.DataType);
      /* wide types only bind as the FIRST input (binder.hpp:308-346) */
      if (k == K_NONE || (!first && (k == K_I64 || k == K_UUID || k == K_GEO)))
        return "Unsupported data type for VectorPartyInput";
      op->kind = k;
      return NULL;
    }
    case ForeignColumnInput: {
      Kind k = kind_of_datatype(in->Vector.ForeignVP.DataType);
      if (k == K_NONE || k == K_GEO || (!first && (k == K_I64 || k == K_UUID)))
        return "Unsupported data type for VectorPartyInput";
      op->kind = k;
      return NULL;
    }
    default:
      return "Unsupported input vector type (array columns are not restated)";
  }
}

static Val default_value(const DefaultValue *dv, Kind k) {
  Val r;
  memset(&r, 0, sizeof(r));
  r.k = k;
  r.ok = dv->HasDefault;
  switch (k) {
    case K_BOOL: r.v.b = dv->Value.BoolVal; break;
    case K_I32: r.v.i = dv->Value.Int32Val; break;
    case K_U32: r.v.u = dv->Value.Uint32Val; break;
    case K_F32: r.v.f = dv->Value.FloatVal; break;
    case K_I64: r.v.l = dv->Value.Int64Val; break;
    case K_UUID: r.v.uuid = dv->Value.UUIDVal; break;
    case K_GEO: r.v.geo = dv->Value.GeoPointVal; break;
    default: break;
  }
  return r;
}

/* read element `p` of a non-constant VectorPartySlice — iterator.hpp:128-207 */
static Val vp_read(const VectorPartySlice *vp, Kind k, uint32_t p) {
  Val r;
  memset(&r, 0, sizeof(r));
  r.k = k;
  int mode = vp->ValuesOffset == 0 ? 1 : (vp->NullsOffset == 0 ? 2 : 3);
  const uint8_t *values = vp->BasePtr + vp->ValuesOffset;
  uint32_t bit = p + vp->StartingIndex;
  if (k == K_GEO) { /* iterator.hpp:291-355: validity bitmap always at BasePtr */
    r.ok = vp->ValuesOffset == 0 ? true : get_bit(vp->BasePtr, bit);
  } else {
    r.ok = mode >= 2 ? get_bit(vp->BasePtr + vp->NullsOffset, bit) : true;
  }
  int step = step_in_bytes(vp->DataType);
  switch (k) {
    case K_BOOL: r.v.b = get_bit(values, bit); break;
    case K_U32:
      if (step == 2) { uint16_t x; memcpy(&x, values + 2 * (size_t)p, 2); r.v.u = x; }
      else if (step == 4) { memcpy(&r.v.u, values + 4 * (size_t)p, 4); }
      else r.v.u = values[p];
      break;
    case K_I32:
      if (step == 2) { int16_t x; memcpy(&x, values + 2 * (size_t)p, 2); r.v.i = x; }
      else if (step == 4) { memcpy(&r.v.i, values + 4 * (size_t)p, 4); }
      else r.v.i = (int8_t)values[p];
      break;
    case K_F32: memcpy(&r.v.f, values + 4 * (size_t)p, 4); break;
    case K_I64: memcpy(&r.v.l, values + 8 * (size_t)p, 8); break;
    case K_UUID: memcpy(&r.v.uuid, values + 16 * (size_t)p, 16); break;
    case K_GEO: memcpy(&r.v.geo, values + 8 * (size_t)p, 8); break;
    default: break;
  }
  return r;
}

/* position of logical row `row` inside a (possibly run-length compressed) slice —
 * iterator.hpp:209-278: mode 3 finds the run p with counts[p] <= x < counts[p+1]. */
static uint32_t vp_locate(const VectorPartySlice *vp, Kind k, uint32_t row,
                          const uint32_t *baseCounts, uint32_t startCount) {
  bool mode3 = vp->ValuesOffset != 0 && vp->NullsOffset != 0;
  if (!mode3 || k == K_GEO) return row;
  uint32_t x = baseCounts ? baseCounts[row] : startCount + row;
  const uint32_t *counts = (const uint32_t *)vp->BasePtr;
  uint32_t first = 0, last = vp->Length;
  while (first < last) {
    uint32_t mid = first + (last - first) / 2;
    if (counts[mid] > x) last = mid; else first = mid + 1;
  }
  return first - 1;
}

static Val load_operand(const Operand *op, int i) {
  const InputVector *in = op->in;
  Val r;
  memset(&r, 0, sizeof(r));
  r.k = op->kind;
  switch (in->Type) {
    case ConstantInput: {
      const ConstantVector *c = &in->Vector.Constant;
      r.ok = c->IsValid;
      switch (op->kind) {
        case K_I32: r.v.i = c->Value.IntVal; break;
        case K_F32: r.v.f = c->Value.FloatVal; break;
        case K_GEO: r.v.geo = c->Value.GeoPointVal; break;
        case K_UUID: r.v.uuid = c->Value.UUIDVal; break;
        default: break;
      }
      return r;
    }
    case ScratchSpaceInput: { /* iterator.hpp:501-513 */
      const ScratchSpaceVector *s = &in->Vector.ScratchSpace;
      size_t w = op->kind == K_UUID ? 16 : (op->kind == K_GEO ? 8 : 4);
      memcpy(&r.v, s->Values + w * (size_t)i, w);
      r.ok = s->Values[s->NullsOffset + (size_t)i] != 0;
      return r;
    }
    case VectorPartyInput: {
      const VectorPartySlice *vp = &in->Vector.VP;
      if (vp->BasePtr == NULL) return default_value(&vp->DefaultValue, op->kind);
      uint32_t row = op->indexVector[i];
      uint32_t p = vp_locate(vp, op->kind, row, op->baseCounts, op->startCount);
      return vp_read(vp, op->kind, p);
    }
    case ForeignColumnInput: { /* iterator.hpp:911-930 */
      const ForeignColumnVector *f = &in->Vector.ForeignVP;
      RecordID rid = f->RecordIDs[i];
      if (rid.batchID && (rid.batchID - f->BaseBatchID < f->NumBatches - 1 ||
                          rid.index < (uint32_t)f->NumRecordsInLastBatch)) {
        const VectorPartySlice *vp = &f->Batches[rid.batchID - f->BaseBatchID];
        if (vp->BasePtr == NULL) return default_value(&f->DefaultValue, op->kind);
        VectorPartySlice tmp = *vp;
        tmp.DataType = f->DataType; /* step comes from the foreign column's type */
        Val v = vp_read(&tmp, op->kind, rid.index);
        if (f->TimezoneLookup && op->kind != K_UUID && op->kind != K_GEO) {
          /* iterator.hpp:894-908: enum -> utc offset */
          Val e = convert(v, K_I32);
          int ev = e.v.i;
          int16_t off = ev < f->TimezoneLookupSize ? f->TimezoneLookup[ev] : 0;
          Val t;
          memset(&t, 0, sizeof(t));
          t.k = K_I32; t.ok = v.ok; t.v.i = off;
          return convert(t, op->kind);
        }
        return v;
      }
      r.ok = false; /* value is indeterminate in the reference; we produce 0 */
      return r;
    }
    default:
      return r;
  }
}

/* ------------------------------------------------------------------------------------------
 * functors — query/functor.hpp:30-351 (leaf), :660-697 / :760-915 (UnaryFunctor and its
 * specialisations), :918-970 / :1037-1076 (BinaryFunctor), GetHLLValue :431-466
 * ---------------------------------------------------------------------------------------- */
static Val mk_bool(bool b, bool ok) { Val r; memset(&r, 0, sizeof(r)); r.k = K_BOOL; r.v.b = b; r.ok = ok; return r; }
static Val mk_u32(uint32_t u, bool ok) { Val r; memset(&r, 0, sizeof(r)); r.k = K_U32; r.v.u = u; r.ok = ok; return r; }
static Val mk_zero(Kind k, bool ok) { Val r; memset(&r, 0, sizeof(r)); r.k = k; r.ok = ok; return r; }

static Val hll_value(Val t) { /* functor.hpp:431-466 */
  if (!t.ok) return mk_u32(0, false);
  uint64_t hashed;
  if (t.k == K_UUID) {
    hashed = t.v.uuid.p1 ^ t.v.uuid.p2;
  } else {
    uint8_t buf[8];
    int n;
    if (t.k == K_BOOL) { buf[0] = t.v.b; n = 1; }
    else if (t.k == K_I64) { memcpy(buf, &t.v.l, 8); n = 8; }
    else { memcpy(buf, &t.v.u, 4); n = 4; }
    uint64_t out[2];
    oracle_murmur3_128(buf, n, 0, out);
    hashed = out[0];
  }
  uint32_t group = (uint32_t)(hashed & ((1 << HLL_BITS) - 1));
  uint32_t rho = 0;
  for (;;) {
    /* the original tests `hashed & (1 << (rho + HLL_BITS))` with a 32-bit int shift: the
     * shift count wraps mod 32 and bit 31 sign-extends, so only the low word is ever probed */
    uint32_t h = (uint32_t)(hashed & (uint64_t)(int64_t)(int32_t)(1u << ((rho + HLL_BITS) & 31)));
    if (rho + HLL_BITS < 64 && h == 0) rho++;
    else break;
  }
  return mk_u32(rho << 16 | group, true);
}

static Val date_functor(Val t, int which) {
  Val a = convert(t, K_U32);
  if (!a.ok) return mk_u32(0, false);
  if (which < 0) return mk_u32(oracle_week_start(a.v.u), true);
  return mk_u32(oracle_resolve_time_bucketizer((int64_t)a.v.u, which), true);
}

/* result carries its own kind; the caller converts to the output type */
static Val unary_apply(int ft, Val t, Kind outKind) {
  Kind I = t.k;
  if (outKind == K_UUID) { /* functor.hpp:720-735, :800-812 */
    if (I == K_UUID) return t;
    return mk_zero(K_UUID, false);
  }
  if (outKind == K_GEO) { /* :836-881 */
    if (I == K_GEO) return t;
    return mk_zero(K_GEO, false);
  }
  if (I == K_UUID) { /* :775-797 */
    if (ft == GetHLLValue) return hll_value(t);
    return mk_zero(outKind, false);
  }
  if (I == K_GEO) return mk_zero(outKind, false); /* :815-833 */
  switch (ft) {
    case Not: { Val a = convert(t, K_BOOL); return a.ok ? mk_bool(!a.v.b, true) : mk_bool(false, false); }
    case IsNull: return mk_bool(!t.ok, true);
    case IsNotNull: return mk_bool(t.ok, true);
    case Noop: return t;
    case Negate: {
      if (!t.ok) return mk_zero(I, false);
      Val r = t;
      switch (I) {
        case K_BOOL: r.v.b = t.v.b; break; /* bool(-int(b)) == b */
        case K_I32: r.v.i = (int32_t)(0u - (uint32_t)t.v.i); break;
        case K_U32: r.v.u = 0u - t.v.u; break;
        case K_F32: r.v.f = -t.v.f; break;
        case K_I64: r.v.l = (int64_t)(0ull - (uint64_t)t.v.l); break;
        default: break;
      }
      return r;
    }
    default: break;
  }
  if (I == K_F32) return t; /* float specialisation: everything else returns t (:738-772) */
  switch (ft) {
    case BitwiseNot: {
      if (!t.ok) return mk_zero(I, false);
      Val r = t;
      switch (I) {
        case K_BOOL: r.v.b = true; break; /* bool(~int(b)) is always true */
        case K_I32: r.v.i = ~t.v.i; break;
        case K_U32: r.v.u = ~t.v.u; break;
        case K_I64: r.v.l = ~t.v.l; break;
        default: break;
      }
      return r;
    }
    case GetWeekStart: return date_functor(t, -1);
    case GetMonthStart: return date_functor(t, TB_MONTH);
    case GetQuarterStart: return date_functor(t, TB_QUARTER);
    case GetYearStart: return date_functor(t, TB_YEAR);
    case GetDayOfMonth: return date_functor(t, TB_DAY_OF_MONTH);
    case GetDayOfYear: return date_functor(t, TB_DAY_OF_YEAR);
    case GetMonthOfYear: return date_functor(t, TB_MONTH_OF_YEAR);
    case GetQuarterOfYear: return date_functor(t, TB_QUARTER_OF_YEAR);
    case GetHLLValue: return hll_value(t);
    default: return t;
  }
}

static Val binary_apply(int ft, Val a, Val b, Kind outKind) {
  Kind I = a.k; /* == b.k after conversion to the common type */
  if (outKind == K_UUID || outKind == K_GEO) return mk_zero(outKind, false); /* :973-1005 */
  if (I == K_UUID || I == K_GEO) { /* :1079-1133: only Equal */
    if (ft != Equal) return mk_zero(K_BOOL, false);
    if (!a.ok || !b.ok) return mk_bool(false, false);
    if (I == K_UUID)
      return mk_bool(a.v.uuid.p1 == b.v.uuid.p1 && a.v.uuid.p2 == b.v.uuid.p2, true);
    return mk_bool(a.v.geo.Lat == b.v.geo.Lat && a.v.geo.Long == b.v.geo.Long, true);
  }
  bool nul = !a.ok || !b.ok;
  switch (ft) {
    case And: { /* :30-42 */
      Val x = convert(a, K_BOOL), y = convert(b, K_BOOL);
      if (nul) return mk_bool(false, false);
      return mk_bool(x.v.b && y.v.b, true);
    }
    case Or: { /* :44-66 */
      Val x = convert(a, K_BOOL), y = convert(b, K_BOOL);
      if ((x.v.b && x.ok) || (y.v.b && y.ok)) return mk_bool(true, true);
      if (nul) return mk_bool(false, false);
      return mk_bool(false, true);
    }
    default: break;
  }
  if (ft >= Equal && ft <= GreaterThanOrEqual) {
    if (nul) return mk_bool(false, false);
    bool r = false;
#define CMP(x, y)                                                                     \
  switch (ft) {                                                                       \
    case Equal: r = (x) == (y); break;                                                \
    case NotEqual: r = (x) != (y); break;                                             \
    case LessThan: r = (x) < (y); break;                                              \
    case LessThanOrEqual: r = (x) <= (y); break;                                      \
    case GreaterThan: r = (x) > (y); break;                                           \
    default: r = (x) >= (y); break;                                                   \
  }
    if (I == K_F32) { CMP(a.v.f, b.v.f) }
    else if (I == K_I32) { CMP(a.v.i, b.v.i) }
    else { CMP(a.v.u, b.v.u) }
#undef CMP
    return mk_bool(r, true);
  }
  if (I == K_F32) { /* :1037-1076 */
    if (ft < Plus || ft > Divide) return a; /* default: return t1 */
    if (nul) return mk_zero(K_F32, false);
    Val r = mk_zero(K_F32, true);
    switch (ft) {
      case Plus: r.v.f = a.v.f + b.v.f; break;
      case Minus: r.v.f = a.v.f - b.v.f; break;
      case Multiply: r.v.f = a.v.f * b.v.f; break;
      default: r.v.f = a.v.f / b.v.f; break;
    }
    return r;
  }
  if (ft < Plus || ft > Floor) return a; /* unknown functor: return t1 (:962-966) */
  if (nul) return mk_zero(I, false);
  Val r = mk_zero(I, true);
  if (I == K_I32) {
    int32_t x = a.v.i, y = b.v.i;
    uint32_t ux = (uint32_t)x, uy = (uint32_t)y;
    switch (ft) {
      case Plus: r.v.i = (int32_t)(ux + uy); break;
      case Minus: r.v.i = (int32_t)(ux - uy); break;
      case Multiply: r.v.i = (int32_t)(ux * uy); break;
      case Divide: r.v.i = x / y; break;
      case Mod: r.v.i = x % y; break;
      case BitwiseAnd: r.v.i = x & y; break;
      case BitwiseOr: r.v.i = x | y; break;
      case BitwiseXor: r.v.i = x ^ y; break;
      default: r.v.i = x - x % y; break; /* Floor :337-351 */
    }
  } else {
    uint32_t x = a.v.u, y = b.v.u;
    switch (ft) {
      case Plus: r.v.u = x + y; break;
      case Minus: r.v.u = x - y; break;
      case Multiply: r.v.u = x * y; break;
      case Divide: r.v.u = x / y; break;
      case Mod: r.v.u = x % y; break;
      case BitwiseAnd: r.v.u = x & y; break;
      case BitwiseOr: r.v.u = x | y; break;
      case BitwiseXor: r.v.u = x ^ y; break;
      default: r.v.u = x - x % y; break;
    }
  }
  return r;
}

/* ------------------------------------------------------------------------------------------
 * outputs — query/transform.hpp:112-171 and the three *_transform.cu binders
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const OutputVector *out; /* NULL => filter predicate */
  uint8_t *pred;
  const uint32_t *indexVector;
  const uint32_t *baseCounts;
  Kind kind; /* value kind the functor result converts to (K_NONE for 8-byte float) */
  int width; /* bytes per stored element */
  bool f64;  /* Float64 measure */
} Sink;

static const char *bind_sink(Sink *s, const OutputVector *out, uint8_t *pred,
                             const uint32_t *indexVector, const uint32_t *baseCounts) {
  memset(s, 0, sizeof(*s));
  s->out = out;
  s->pred = pred;
  s->indexVector = indexVector;
  s->baseCounts = baseCounts;
  if (!out) { s->kind = K_BOOL; s->width = 1; return NULL; }
  enum DataType t;
  switch (out->Type) {
    case ScratchSpaceOutput:
      t = out->Vector.ScratchSpace.DataType;
      if (!(t == Int32 || t == Uint32 || t == Float32 || t == Int64 || t == UUID || t == GeoPoint))
        return "Unsupported data type for ScratchSpaceOutput";
      break;
    case MeasureOutput:
      t = out->Vector.Measure.DataType;
      if (t == Float64) { s->f64 = true; s->kind = K_F32; s->width = 8; return NULL; }
      if (!(t == Int32 || t == Uint32 || t == Float32 || t == Int64))
        return "Unsupported data type for MeasureOutput";
      break;
    case DimensionOutput:
      t = out->Vector.Dimension.DataType;
      if (t == Uint64 || t == Float64) return "Unsupported data type for DimensionOutput";
      break;
    default:
      return "Unsupported output vector type";
  }
  s->kind = kind_of_datatype(t);
  s->width = step_in_bytes(t);
  if (t == Bool) s->width = 1;
  return NULL;
}

/* identity of an aggregate, as the measure type — query/utils.hpp:169-184 (quirk:
 * AGGR_MAX_FLOAT uses FLT_MIN) */
static void identity_bytes(enum AggregateFunction agg, enum DataType t, uint8_t out[8]) {
  double d = 0;
  int64_t l = 0;
  bool fl = false;
  switch (agg) {
    case AGGR_MIN_UNSIGNED: l = (int64_t)UINT32_MAX; d = (double)UINT32_MAX; break;
    case AGGR_MIN_SIGNED: l = INT32_MAX; d = INT32_MAX; break;
    case AGGR_MIN_FLOAT: fl = true; d = FLT_MAX; break;
    case AGGR_MAX_SIGNED: l = INT32_MIN; d = INT32_MIN; break;
    case AGGR_MAX_FLOAT: fl = true; d = FLT_MIN; break;
    default: break;
  }
  memset(out, 0, 8);
  switch (t) {
    case Int32: { int32_t x = fl ? (int32_t)d : (int32_t)l; memcpy(out, &x, 4); break; }
    case Uint32: { uint32_t x = fl ? (uint32_t)d : (uint32_t)l; memcpy(out, &x, 4); break; }
    case Float32: { float x = fl ? (float)d : (float)l; memcpy(out, &x, 4); break; }
    case Int64: { int64_t x = fl ? (int64_t)d : l; memcpy(out, &x, 8); break; }
    case Float64: { double x = fl ? d : (double)l; memcpy(out, &x, 8); break; }
    default: break;
  }
}

static void store_typed(uint8_t *dst, enum DataType t, Val r) {
  switch (t) {
    case Bool: { uint8_t x = convert(r, K_BOOL).v.b; *dst = x; break; }
    case Int8: { int8_t x = (int8_t)convert(r, K_I32).v.i; memcpy(dst, &x, 1); break; }
    case Uint8: { uint8_t x = (uint8_t)convert(r, K_U32).v.u; *dst = x; break; }
    case Int16: { int16_t x = (int16_t)convert(r, K_I32).v.i; memcpy(dst, &x, 2); break; }
    case Uint16: { uint16_t x = (uint16_t)convert(r, K_U32).v.u; memcpy(dst, &x, 2); break; }
    case Int32: { int32_t x = convert(r, K_I32).v.i; memcpy(dst, &x, 4); break; }
    case Uint32: { uint32_t x = convert(r, K_U32).v.u; memcpy(dst, &x, 4); break; }
    case Float32: { float x = convert(r, K_F32).v.f; memcpy(dst, &x, 4); break; }
    case Int64: { int64_t x = convert(r, K_I64).v.l; memcpy(dst, &x, 8); break; }
    case UUID: { UUIDT x = r.k == K_UUID ? r.v.uuid : (UUIDT){0, 0}; memcpy(dst, &x, 16); break; }
    case GeoPoint: { GeoPointT x = r.k == K_GEO ? r.v.geo : (GeoPointT){0, 0}; memcpy(dst, &x, 8); break; }
    default: break;
  }
}

/* narrow-type conversions of the reference go straight from the functor's value type to the
 * output type (e.g. float -> uint8).  For the integer kinds our two-step path through
 * int32/uint32 is value-identical; for float inputs we convert directly. */
static void store_narrow_from_float(uint8_t *dst, enum DataType t, float f) {
  switch (t) {
    case Int8: { int8_t x = (int8_t)f; memcpy(dst, &x, 1); break; }
    case Uint8: { uint8_t x = (uint8_t)f; *dst = x; break; }
    case Int16: { int16_t x = (int16_t)f; memcpy(dst, &x, 2); break; }
    case Uint16: { uint16_t x = (uint16_t)f; memcpy(dst, &x, 2); break; }
    default: break;
  }
}

static void sink_store(const Sink *s, int i, Val r) {
  if (!s->out) { /* filter: functor.hpp:903-915 keeps the VALUE, ignores validity */
    s->pred[i] = convert(r, K_BOOL).v.b ? 1 : 0;
    return;
  }
  const OutputVector *o = s->out;
  switch (o->Type) {
    case ScratchSpaceOutput: { /* iterator.hpp:600-614 */
      const ScratchSpaceVector *v = &o->Vector.ScratchSpace;
      store_typed(v->Values + (size_t)s->width * i, v->DataType, r);
      v->Values[v->NullsOffset + (size_t)i] = r.ok ? 1 : 0;
      return;
    }
    case DimensionOutput: { /* iterator.hpp:570-584 */
      const DimensionOutputVector *v = &o->Vector.Dimension;
      uint8_t *dst = v->DimValues + (size_t)s->width * i;
      if (r.k == K_F32 && (v->DataType == Int8 || v->DataType == Uint8 ||
                           v->DataType == Int16 || v->DataType == Uint16))
        store_narrow_from_float(dst, v->DataType, r.v.f);
      else
        store_typed(dst, v->DataType, r);
      v->DimNulls[i] = r.ok ? 1 : 0;
      return;
    }
    case MeasureOutput: { /* iterator.hpp:616-727 */
      const MeasureOutputVector *v = &o->Vector.Measure;
      uint8_t *dst = (uint8_t *)v->Values + (size_t)s->width * i;
      enum AggregateFunction agg = v->AggFunc;
      bool isSum = agg >= AGGR_SUM_UNSIGNED && agg <= AGGR_SUM_FLOAT;
      bool isAvg = agg == AGGR_AVG_FLOAT;
      uint32_t count = 1;
      if ((isSum || isAvg) && s->baseCounts) {
        uint32_t idx = s->indexVector[i];
        count = s->baseCounts[idx + 1] - s->baseCounts[idx];
      }
      if (!r.ok) { identity_bytes(agg, v->DataType, dst); return; }
      if (isAvg) { /* {float avg, u32 count} packed in the first 8 bytes */
        float f;
        if (s->f64) f = (float)(double)convert(r, K_F32).v.f;
        else if (v->DataType == Int64) f = (float)convert(r, K_I64).v.l;
        else if (v->DataType == Int32) f = (float)convert(r, K_I32).v.i;
        else if (v->DataType == Uint32) f = (float)convert(r, K_U32).v.u;
        else f = convert(r, K_F32).v.f;
        memcpy(dst, &f, 4);
        memcpy(dst + 4, &count, 4);
        return;
      }
      switch (v->DataType) {
        case Int32: { int32_t x = (int32_t)((uint32_t)convert(r, K_I32).v.i * count); memcpy(dst, &x, 4); break; }
        case Uint32: { uint32_t x = convert(r, K_U32).v.u * count; memcpy(dst, &x, 4); break; }
        case Float32: { float x = convert(r, K_F32).v.f * (float)count; memcpy(dst, &x, 4); break; }
        case Int64: { int64_t x = (int64_t)((uint64_t)convert(r, K_I64).v.l * (uint64_t)count); memcpy(dst, &x, 8); break; }
        case Float64: {
          double x;
          if (r.k == K_F32) x = (double)r.v.f;
          else if (r.k == K_I64) x = (double)r.v.l;
          else if (r.k == K_U32) x = (double)r.v.u;
          else if (r.k == K_BOOL) x = r.v.b ? 1.0 : 0.0;
          else x = (double)r.v.i;
          x = x * (double)count;
          memcpy(dst, &x, 8);
          break;
        }
        default: break;
      }
      return;
    }
  }
}

/* kind the functor's result_type carries for a sink (thrust::tuple<O,bool>) */
static Kind sink_result_kind(const Sink *s) {
  if (!s->out) return K_BOOL;
  return s->kind;
}

/* ------------------------------------------------------------------------------------------
 * generic drivers
 * ---------------------------------------------------------------------------------------- */
static const char *run_array(const InputVector *in, const InputVector *rhs, Sink *s, int n, int ft);

static const char *run_unary(const InputVector *in, Sink *s, const uint32_t *idx, int n,
                             const uint32_t *baseCounts, uint32_t startCount, int ft) {
  Operand a;
  if (in->Type == ArrayVectorPartyInput) return run_array(in, NULL, s, n, ft);
  const char *e = bind_operand(&a, in, idx, baseCounts, startCount, true);
  if (e) return e;
  Kind ok = sink_result_kind(s);
  for (int i = 0; i < n; i++) {
    Val x = load_operand(&a, i);
    Val r = unary_apply(ft, x, ok);
    sink_store(s, i, r);
  }
  return NULL;
}

static const char *run_binary(const InputVector *l, const InputVector *r_, Sink *s,
                              const uint32_t *idx, int n, const uint32_t *baseCounts,
                              uint32_t startCount, int ft) {
  Operand a, b;
  if (l->Type == ArrayVectorPartyInput) return run_array(l, r_, s, n, ft);
  const char *e = bind_operand(&a, l, idx, baseCounts, startCount, true);
  if (e) return e;
  e = bind_operand(&b, r_, idx, baseCounts, startCount, false);
  if (e) return e;
  if (a.kind == K_I64) return "int64 data type is only supported in UnaryTransform";
  if (a.kind == K_GEO && !(r_->Type == ConstantInput && b.kind == K_GEO))
    return "Unsupported data type when value type of first input iterator is GeoPoint";
  if (a.kind == K_UUID && !(r_->Type == ConstantInput && b.kind == K_UUID))
    return "Unsupported data type when value type of first input iterator is UUID";
  if (a.kind != K_GEO && a.kind != K_UUID && (b.kind == K_GEO || b.kind == K_UUID))
    return "Unsupported data type combination";
  Kind I = common_kind(a.kind, b.kind);
  Kind ok = sink_result_kind(s);
  for (int i = 0; i < n; i++) {
    Val x = convert(load_operand(&a, i), I);
    Val y = convert(load_operand(&b, i), I);
    Val r = binary_apply(ft, x, y, ok);
    sink_store(s, i, r);
  }
  return NULL;
}

/* ------------------------------------------------------------------------------------------
 * array columns — query/iterator.hpp:377-451 (ArrayVectorPartyIterator), query/functor.hpp:468-640
 * (ArrayLength / ArrayElementAt / ArrayContains), binding rules query/binder.hpp:385-426, :458-560.
 * An array column is [offset u32, length u32] x Length followed by the values; one array value is
 * [length u32][elements][validity bits].  The iterator is NOT zipped with the index vector: output
 * position i reads array i (binder.hpp:392-394 binds the bare iterator).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const uint8_t *value; /* NULL: empty (valid) or null (invalid) array */
  bool ok;
} ArrayRef;

static ArrayRef array_at(const ArrayVectorPartySlice *a, int i) {
  const uint8_t *base = a->OffsetLengthVector;
  const uint8_t *valuePtr = base + 8 * (size_t)a->Length - a->ValueOffsetAdj;
  uint32_t offset, length;
  memcpy(&offset, base + 8 * (size_t)i, 4);
  memcpy(&length, base + 8 * (size_t)i + 4, 4);
  ArrayRef r;
  if (length == 0) { r.value = NULL; r.ok = offset != 0; return r; }
  r.value = valuePtr + offset;
  r.ok = true;
  return r;
}

/* element j of an array value as a Val of the element's storage kind */
static Val array_element(const uint8_t *elems, enum DataType t, int j) {
  Val r;
  memset(&r, 0, sizeof(r));
  r.ok = true;
  r.k = kind_of_datatype(t);
  switch (t) {
    case Bool: r.v.b = elems[j] != 0; break;
    case Int8: r.v.i = (int8_t)elems[j]; break;
    case Uint8: r.v.u = elems[j]; break;
    case Int16: { int16_t x; memcpy(&x, elems + 2 * j, 2); r.v.i = x; break; }
    case Uint16: { uint16_t x; memcpy(&x, elems + 2 * j, 2); r.v.u = x; break; }
    case Int32: memcpy(&r.v.i, elems + 4 * j, 4); break;
    case Uint32: memcpy(&r.v.u, elems + 4 * j, 4); break;
    case Float32: memcpy(&r.v.f, elems + 4 * j, 4); break;
    case Int64: memcpy(&r.v.l, elems + 8 * j, 8); break;
    case UUID: memcpy(&r.v.uuid, elems + 16 * j, 16); break;
    case GeoPoint: memcpy(&r.v.geo, elems + 8 * j, 8); break;
    default: break;
  }
  return r;
}

static Val null_of(Kind k) {
  Val r;
  memset(&r, 0, sizeof(r));
  r.k = k;
  return r;
}

/* result of one array functor at position i; `ok_kind` = value type O of the sink's functor result;
 * c = constant second operand (binary functors), NULL for ArrayLength */
static Val array_apply(const ArrayVectorPartySlice *a, int i, int ft, bool binary, const ConstantVector *c, enum DataType odt) {
  const Kind ok_kind = kind_of_datatype(odt) == K_NONE ? K_F32 : kind_of_datatype(odt); /* kind of a null result */
  const enum DataType t = a->DataType;
  const int w = step_in_bytes(t);
  const Kind ek = kind_of_datatype(t);
  ArrayRef ar = array_at(a, i);
  if (!binary) { /* functor.hpp:700-723: only ArrayLength, and only into uint32 */
    if (ft != ArrayLength || odt != Uint32) return null_of(ok_kind);
    Val r = null_of(K_U32);
    if (!ar.ok) return r;
    r.ok = true;
    if (ar.value) memcpy(&r.v.u, ar.value, 4);
    return r;
  }
  const bool wide = ek == K_UUID || ek == K_GEO;
  if (ft == ArrayContains) { /* functor.hpp:610-640: result type bool only */
    if (odt != Bool) return null_of(ok_kind);
    if (wide ? (ek == K_UUID ? c->DataType != ConstUUID : (c->DataType != ConstGeoPoint && c->DataType != ConstUUID))
             : (c->DataType != ConstInt && c->DataType != ConstFloat))
      return null_of(K_BOOL); /* (binder.hpp:483-558 lets ConstInt through for wide arrays: generic functor, null) */
    Val r = null_of(K_BOOL);
    if (!ar.ok) return r;
    r.ok = true;
    if (!ar.value) return r;
    int32_t len;
    memcpy(&len, ar.value, 4);
    if (len <= 0) return r;
    const uint8_t *elems = ar.value + 4, *valid = elems + ((size_t)w * 8 * len + 7) / 8;
    for (int j = 0; j < len; j++) {
      if (!get_bit(valid, (uint32_t)j)) continue;
      Val e = array_element(elems, t, j);
      bool eq;
      if (ek == K_UUID) eq = e.v.uuid.p1 == c->Value.UUIDVal.p1 && e.v.uuid.p2 == c->Value.UUIDVal.p2;
      else if (ek == K_GEO) /* the constant travels in the upper half of SimpleIterator<GeoPointT>'s 64-bit pointer
                             * (iterator.hpp:484-516), which keeps its first 4 bytes only: Long arrives as 0 */
        eq = e.v.geo.Lat == c->Value.GeoPointVal.Lat && e.v.geo.Long == 0.0f;
      else { /* val = static_cast<input_type>(constant); equals(val, element) */
        const bool cf = c->DataType == ConstFloat;
        switch (t) {
          case Bool: eq = (cf ? c->Value.FloatVal != 0.0f : c->Value.IntVal != 0) == e.v.b; break;
          case Int8: eq = (cf ? (int8_t)c->Value.FloatVal : (int8_t)c->Value.IntVal) == (int8_t)e.v.i; break;
          case Uint8: eq = (cf ? (uint8_t)c->Value.FloatVal : (uint8_t)c->Value.IntVal) == (uint8_t)e.v.u; break;
          case Int16: eq = (cf ? (int16_t)c->Value.FloatVal : (int16_t)c->Value.IntVal) == (int16_t)e.v.i; break;
          case Uint16: eq = (cf ? (uint16_t)c->Value.FloatVal : (uint16_t)c->Value.IntVal) == (uint16_t)e.v.u; break;
          case Int32: eq = (cf ? (int32_t)c->Value.FloatVal : c->Value.IntVal) == e.v.i; break;
          case Uint32: eq = (cf ? (uint32_t)c->Value.FloatVal : (uint32_t)c->Value.IntVal) == e.v.u; break;
          case Float32: eq = (cf ? c->Value.FloatVal : (float)c->Value.IntVal) == e.v.f; break;
          case Int64: eq = (cf ? (int64_t)c->Value.FloatVal : (int64_t)c->Value.IntVal) == e.v.l; break;
          default: eq = false; break;
        }
      }
      if (eq) { r.v.b = true; return r; }
    }
    return r;
  }
  if (ft == ArrayElementAt) { /* functor.hpp:515-571: index must be ConstInt; UUID / GeoPoint only into themselves */
    if (c->DataType != ConstInt) return null_of(ok_kind);
    if ((ek == K_UUID) != (odt == UUID) || (ek == K_GEO) != (odt == GeoPoint)) return null_of(ok_kind);
    Val zero = null_of(wide ? ek : ok_kind);
    if (!ar.ok || !ar.value) return zero;
    uint32_t ulen;
    memcpy(&ulen, ar.value, 4);
    int index = c->Value.IntVal;
    if ((index >= 0 && ulen <= (uint32_t)index) || (index < 0 && ulen < (uint32_t)(-index))) return zero;
    const int len = (int)ulen;
    if (index < 0) index = len + index;
    if (len == 0 || index >= len || index < 0) return zero;
    const uint8_t *elems = ar.value + 4, *valid = elems + ((size_t)w * 8 * len + 7) / 8;
    if (!get_bit(valid, (uint32_t)index)) return zero;
    return array_element(elems, t, index); /* the sink applies static_cast<O>(element) */
  }
  return null_of(ok_kind);
}

static const char *run_array(const InputVector *in, const InputVector *rhs, Sink *s, int n, int ft) {
  const ArrayVectorPartySlice *a = &in->Vector.ArrayVP;
  if (kind_of_datatype(a->DataType) == K_NONE) return "Unsupported data type for ArrayVectorPartyInput";
  if (rhs && rhs->Type != ConstantInput)
    return "Unsupported data type when value type of first input iterator is ArrayVP Iterator";
  if (rhs) { /* binder.hpp:469-558: which constants an array of this element type binds with */
    const Kind ek = kind_of_datatype(a->DataType);
    const int ct = (int)rhs->Vector.Constant.DataType;
    bool okc = ek == K_UUID ? (ct == ConstInt || ct == ConstUUID)
               : ek == K_GEO ? (ct == ConstInt || ct == ConstGeoPoint || ct == ConstUUID)
                             : (ct == ConstInt || ct == ConstFloat);
    if (!okc) return "Unsupported data type when value type of first input iterator is ArrayVP Iterator";
  }
  /* O = value type of the sink's iterator: the stored data type (bool for a filter's predicate) */
  enum DataType odt = Bool;
  if (s->out)
    odt = s->out->Type == ScratchSpaceOutput ? s->out->Vector.ScratchSpace.DataType
          : s->out->Type == DimensionOutput ? s->out->Vector.Dimension.DataType : s->out->Vector.Measure.DataType;
  for (int i = 0; i < n; i++)
    sink_store(s, i, array_apply(a, i, ft, rhs != NULL, rhs ? &rhs->Vector.Constant : NULL, odt));
  return NULL;
}

/* stable in-place compaction — query/filter.cu:209-253 (thrust::remove_if on
 * zip(counting, indexVector, recordIDs...)) */
static int compact(uint32_t *idx, const uint8_t *pred, int n, RecordID **rids, int nForeign) {
  int w = 0;
  for (int i = 0; i < n; i++) {
    if (pred[i]) {
      idx[w] = idx[i];
      for (int t = 0; t < nForeign; t++) rids[t][w] = rids[t][i];
      w++;
    }
  }
  return w;
}

#define HANDLE_OK(x) ((CGoCallResHandle){(void *)(intptr_t)(x), NULL})
#define HANDLE_ERR(msg) ((CGoCallResHandle){NULL, dup_err(msg)})

/* ------------------------------------------------------------------------------------------
 * exported entry points
 * ---------------------------------------------------------------------------------------- */
CGoCallResHandle InitIndexVector(uint32_t *indexVector, uint32_t start, int indexVectorLength,
                                 void *cudaStream, int device) { /* algorithm.cu:22-41 */
  (void)cudaStream; (void)device;
  for (int i = 0; i < indexVectorLength; i++) indexVector[i] = start + (uint32_t)i;
  return HANDLE_OK(0);
}

CGoCallResHandle UnaryFilter(InputVector input, uint32_t *indexVector, uint8_t *predicateVector,
                             int indexVectorLength, RecordID **recordIDVectors,
                             int numForeignTables, uint32_t *baseCounts, uint32_t startCount,
                             enum UnaryFunctorType functorType, void *cudaStream, int device) {
  (void)cudaStream; (void)device; /* filter.cu:130-165 */
  if (numForeignTables < 0 || numForeignTables > 8) return HANDLE_ERR("only support up to 8 foreign tables");
  Sink s;
  bind_sink(&s, NULL, predicateVector, indexVector, baseCounts);
  const char *e = run_unary(&input, &s, indexVector, indexVectorLength, baseCounts, startCount,
                            (int)functorType);
  if (e) return HANDLE_ERR(e);
  return HANDLE_OK(compact(indexVector, predicateVector, indexVectorLength, recordIDVectors,
                           numForeignTables));
}

CGoCallResHandle BinaryFilter(InputVector lhs, InputVector rhs, uint32_t *indexVector,
                              uint8_t *predicateVector, int indexVectorLength,
                              RecordID **recordIDVectors, int numForeignTables,
                              uint32_t *baseCounts, uint32_t startCount,
                              enum BinaryFunctorType functorType, void *cudaStream, int device) {
  (void)cudaStream; (void)device; /* filter.cu:167-204 */
  if (numForeignTables < 0 || numForeignTables > 8) return HANDLE_ERR("only support up to 8 foreign tables");
  Sink s;
  bind_sink(&s, NULL, predicateVector, indexVector, baseCounts);
  const char *e = run_binary(&lhs, &rhs, &s, indexVector, indexVectorLength, baseCounts,
                             startCount, (int)functorType);
  if (e) return HANDLE_ERR(e);
  return HANDLE_OK(compact(indexVector, predicateVector, indexVectorLength, recordIDVectors,
                           numForeignTables));
}

CGoCallResHandle UnaryTransform(InputVector input, OutputVector output, uint32_t *indexVector,
                                int indexVectorLength, uint32_t *baseCounts, uint32_t startCount,
                                enum UnaryFunctorType functorType, void *cudaStream, int device) {
  (void)cudaStream; (void)device; /* transform.cu:21-52 */
  Sink s;
  const char *e = bind_sink(&s, &output, NULL, indexVector, baseCounts);
  if (e) return HANDLE_ERR(e);
  e = run_unary(&input, &s, indexVector, indexVectorLength, baseCounts, startCount,
                (int)functorType);
  if (e) return HANDLE_ERR(e);
  return HANDLE_OK(indexVectorLength);
}

CGoCallResHandle BinaryTransform(InputVector lhs, InputVector rhs, OutputVector output,
                                 uint32_t *indexVector, int indexVectorLength,
                                 uint32_t *baseCounts, uint32_t startCount,
                                 enum BinaryFunctorType functorType, void *cudaStream,
                                 int device) {
  (void)cudaStream; (void)device; /* transform.cu:54-86 */
  Sink s;
  const char *e = bind_sink(&s, &output, NULL, indexVector, baseCounts);
  if (e) return HANDLE_ERR(e);
  e = run_binary(&lhs, &rhs, &s, indexVector, indexVectorLength, baseCounts, startCount,
                 (int)functorType);
  if (e) return HANDLE_ERR(e);
  return HANDLE_OK(indexVectorLength);
}

/* hash_lookup.cu:70-157 + functor.hpp:1173-1266 */
CGoCallResHandle HashLookup(InputVector input, RecordID *output, uint32_t *indexVector,
                            int indexVectorLength, uint32_t *baseCounts, uint32_t startCount,
                            CuckooHashIndex hashIndex, void *cudaStream, int device) {
  (void)cudaStream; (void)device;
  Operand a;
  const char *e = bind_operand(&a, &input, indexVector, baseCounts, startCount, true);
  if (e) return HANDLE_ERR(e);
  if (a.kind == K_GEO) return HANDLE_ERR("Unsupported data type for HashLookup");
  const int keyBytes = hashIndex.keyBytes;
  const int bucketBytes = HASH_BUCKET_SIZE * (8 + keyBytes + 1);
  const uint8_t *stash = hashIndex.buckets + (size_t)bucketBytes * hashIndex.numBuckets;
  const int offSig = HASH_BUCKET_SIZE * 8, offKey = offSig + HASH_BUCKET_SIZE;
  for (int i = 0; i < indexVectorLength; i++) {
    Val t = load_operand(&a, i);
    RecordID rid = {0, 0};
    if (t.ok) {
      uint8_t key[16] = {0};
      memcpy(key, &t.v, sizeof(t.v) < 16 ? sizeof(t.v) : 16);
      bool found = false;
      for (int h = 0; h < hashIndex.numHashes && !found; h++) {
        uint32_t hv = oracle_murmur3_32(key, keyBytes, hashIndex.seeds[h]);
        const uint8_t *bucket = hashIndex.buckets + (size_t)(hv % (uint32_t)hashIndex.numBuckets) * bucketBytes;
        uint8_t sig = (uint8_t)(hv >> 24);
        if (sig < 1) sig = 1;
        for (int j = 0; j < HASH_BUCKET_SIZE; j++) {
          if (bucket[offSig + j] == sig && memcmp(bucket + offKey + j * keyBytes, key, keyBytes) == 0) {
            memcpy(&rid, bucket + 8 * j, 8);
            found = true;
            break;
          }
        }
      }
      for (int j = 0; j < HASH_STASH_SIZE && !found; j++) {
        if (stash[offSig + j] != 0 && memcmp(stash + offKey + j * keyBytes, key, keyBytes) == 0) {
          memcpy(&rid, stash + 8 * j, 8);
          found = true;
        }
      }
    }
    output[i] = rid;
  }
  return HANDLE_OK(indexVectorLength);
}

/* ---- dimension rows -------------------------------------------------------------------- */
typedef struct {
  int numDims;
  int valueBytes; /* sum of widths */
  int rowBytes;   /* valueBytes + numDims */
  int width[MAX_DIMENSIONS * 4];
  size_t valueOff[MAX_DIMENSIONS * 4]; /* byte offset of dim d's value vector, per capacity */
} DimLayout;

/* iterator.hpp:955-971 / query/common/dimval.go:122-144 */
static void dim_layout(const uint8_t numDimsPerWidth[NUM_DIM_WIDTH], DimLayout *L) {
  memset(L, 0, sizeof(*L));
  int d = 0;
  size_t off = 0; /* in units of `capacity` bytes */
  for (int w = 0; w < NUM_DIM_WIDTH; w++) {
    int bytes = 1 << (NUM_DIM_WIDTH - 1 - w);
    for (int j = 0; j < numDimsPerWidth[w]; j++) {
      L->width[d] = bytes;
      L->valueOff[d] = off;
      off += bytes;
      d++;
    }
  }
  L->numDims = d;
  L->valueBytes = (int)off;
  L->rowBytes = (int)off + d;
}

/* pack row `index` as [values...][validity bytes...] — iterator.hpp:984-1024 */
static void pack_dim_row(const uint8_t *dimValues, int capacity, const DimLayout *L,
                         uint32_t index, uint8_t row[64]) {
  memset(row, 0, 64);
  const uint8_t *nulls = dimValues + (size_t)L->valueBytes * capacity;
  int o = 0;
  for (int d = 0; d < L->numDims; d++) {
    memcpy(row + o, dimValues + L->valueOff[d] * capacity + (size_t)L->width[d] * index,
           L->width[d]);
    o += L->width[d];
    row[L->valueBytes + d] = nulls[(size_t)d * capacity + index];
  }
}

static void copy_dim_row(const uint8_t *in, int inCap, uint8_t *out, int outCap,
                         const DimLayout *L, uint32_t inIdx, uint32_t outIdx) {
  for (int d = 0; d < L->numDims; d++)
    memcpy(out + L->valueOff[d] * outCap + (size_t)L->width[d] * outIdx,
           in + L->valueOff[d] * inCap + (size_t)L->width[d] * inIdx, L->width[d]);
  const uint8_t *inNulls = in + (size_t)L->valueBytes * inCap;
  uint8_t *outNulls = out + (size_t)L->valueBytes * outCap;
  for (int d = 0; d < L->numDims; d++)
    outNulls[(size_t)d * outCap + outIdx] = inNulls[(size_t)d * inCap + inIdx];
}

/* sort_reduce.cu:118-133 */
typedef struct { uint64_t h; uint32_t idx; } HashIdx;

static void merge_sort(HashIdx *a, HashIdx *tmp, int n) { /* stable */
  if (n < 2) return;
  int m = n / 2;
  merge_sort(a, tmp, m);
  merge_sort(a + m, tmp, n - m);
  int i = 0, j = m, k = 0;
  while (i < m && j < n) tmp[k++] = (a[j].h < a[i].h) ? a[j++] : a[i++];
  while (i < m) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, sizeof(HashIdx) * n);
}

CGoCallResHandle Sort(DimensionVector keys, int length, void *cudaStream, int device) {
  (void)cudaStream; (void)device;
  if (length <= 0) return HANDLE_OK(0);
  DimLayout L;
  dim_layout(keys.NumDimsPerDimWidth, &L);
  HashIdx *a = (HashIdx *)malloc(sizeof(HashIdx) * (size_t)length * 2);
  if (!a) return HANDLE_ERR("out of memory");
  for (int i = 0; i < length; i++) {
    uint8_t row[64];
    uint64_t out[2];
    pack_dim_row(keys.DimValues, keys.VectorCapacity, &L, keys.IndexVector[i], row);
    oracle_murmur3_128(row, L.rowBytes, 0, out);
    a[i].h = out[0];
    a[i].idx = keys.IndexVector[i];
  }
  merge_sort(a, a + length, length);
  for (int i = 0; i < length; i++) {
    keys.HashValues[i] = a[i].h;
    keys.IndexVector[i] = a[i].idx;
  }
  free(a);
  return HANDLE_OK(0);
}

/* one aggregation step on raw value bytes — sort_reduce.cu:170-216, functor.hpp:1414-1436 */
static const char *agg_combine(enum AggregateFunction agg, int valueBytes, uint8_t *acc,
                               const uint8_t *v) {
#define COMBINE(T, EXPR) { T x, y; memcpy(&x, acc, sizeof(T)); memcpy(&y, v, sizeof(T)); x = (EXPR); memcpy(acc, &x, sizeof(T)); return NULL; }
  switch (agg) {
    case AGGR_SUM_UNSIGNED:
      if (valueBytes == 4) COMBINE(uint32_t, x + y) else COMBINE(uint64_t, x + y)
    case AGGR_SUM_SIGNED:
      if (valueBytes == 4) COMBINE(uint32_t, x + y) else COMBINE(uint64_t, x + y)
    case AGGR_SUM_FLOAT:
      if (valueBytes == 4) COMBINE(float, x + y) else COMBINE(double, x + y)
    case AGGR_MIN_UNSIGNED: COMBINE(uint32_t, y < x ? y : x)
    case AGGR_MIN_SIGNED: COMBINE(int32_t, y < x ? y : x)
    case AGGR_MIN_FLOAT: COMBINE(float, y < x ? y : x)
    case AGGR_MAX_UNSIGNED: COMBINE(uint32_t, x < y ? y : x)
    case AGGR_MAX_SIGNED: COMBINE(int32_t, x < y ? y : x)
    case AGGR_MAX_FLOAT: COMBINE(float, x < y ? y : x)
    case AGGR_AVG_FLOAT: {
      uint64_t lhs, rhs;
      memcpy(&lhs, acc, 8);
      memcpy(&rhs, v, 8);
      uint32_t lc = (uint32_t)(lhs >> 32), rc = (uint32_t)(rhs >> 32);
      uint32_t total = lc + rc;
      uint64_t res = 0;
      if (total != 0) {
        float lf, rf;
        memcpy(&lf, &lhs, 4);
        memcpy(&rf, &rhs, 4);
        float f = lf / total * lc + rf / total * rc;
        uint32_t fb;
        memcpy(&fb, &f, 4);
        res = ((uint64_t)total << 32) | fb;
      }
      memcpy(acc, &res, 8);
      return NULL;
    }
    default:
      return "Unsupported aggregation function type";
  }
#undef COMBINE
}

static int agg_value_bytes(enum AggregateFunction agg, int valueBytes) {
  switch (agg) {
    case AGGR_SUM_UNSIGNED: case AGGR_SUM_SIGNED: case AGGR_SUM_FLOAT:
      return valueBytes == 4 ? 4 : 8;
    case AGGR_AVG_FLOAT: return 8;
    default: return 4;
  }
}

/* sort_reduce.cu:135-249 */
CGoCallResHandle Reduce(DimensionVector inputKeys, uint8_t *inputValues,
                        DimensionVector outputKeys, uint8_t *outputValues, int valueBytes,
                        int length, enum AggregateFunction aggFunc, void *cudaStream,
                        int device) {
  (void)cudaStream; (void)device;
  if (aggFunc != AGGR_AVG_FLOAT && (aggFunc < AGGR_SUM_UNSIGNED || aggFunc > AGGR_MAX_FLOAT))
    return HANDLE_ERR("Unsupported aggregation function type");
  int vb = agg_value_bytes(aggFunc, valueBytes);
  int groups = 0;
  for (int i = 0; i < length; i++) {
    uint32_t idx = inputKeys.IndexVector[i];
    const uint8_t *v = inputValues + (size_t)vb * idx;
    if (i == 0 || inputKeys.HashValues[i] != inputKeys.HashValues[i - 1]) {
      outputKeys.IndexVector[groups] = idx;
      memcpy(outputValues + (size_t)vb * groups, v, vb);
      groups++;
    } else {
      const char *e = agg_combine(aggFunc, valueBytes, outputValues + (size_t)vb * (groups - 1), v);
      if (e) return HANDLE_ERR(e);
    }
  }
  /* gather representatives; BOTH strides use inputKeys.VectorCapacity (sort_reduce.cu:234-239) */
  DimLayout L;
  dim_layout(inputKeys.NumDimsPerDimWidth, &L);
  for (int g = 0; g < groups; g++)
    copy_dim_row(inputKeys.DimValues, inputKeys.VectorCapacity, outputKeys.DimValues,
                 inputKeys.VectorCapacity, &L, outputKeys.IndexVector[g], (uint32_t)g);
  return HANDLE_OK(groups);
}

/* hash_reduction.cu:183-391 with the HOST map of concurrent_unordered_map.hpp:81-135.
 * Group identity = 32-bit murmur3 of the packed row; the first row (input order) carrying a
 * hash is the group's representative; a new group starts from a value-initialised 0, not from
 * the aggregate's identity (reference quirk).  Output order here is first-appearance order; the
 * reference's is std::unordered_map iteration order — the ABI leaves it unspecified. */
CGoCallResHandle HashReduce(DimensionVector inputKeys, uint8_t *inputValues,
                            DimensionVector outputKeys, uint8_t *outputValues, int valueBytes,
                            int length, enum AggregateFunction aggFunc, void *cudaStream,
                            int device) {
  (void)cudaStream; (void)device;
  if (length <= 0) return HANDLE_OK(0);
  if (aggFunc != AGGR_AVG_FLOAT && (aggFunc < AGGR_SUM_UNSIGNED || aggFunc > AGGR_MAX_FLOAT))
    return HANDLE_ERR("Unsupported aggregation function type");
  int vb = agg_value_bytes(aggFunc, valueBytes);
  DimLayout L;
  dim_layout(inputKeys.NumDimsPerDimWidth, &L);
  size_t cap = 16;
  while (cap < (size_t)length * 2) cap <<= 1;
  int32_t *slots = (int32_t *)malloc(sizeof(int32_t) * cap); /* slot -> group or -1 */
  uint32_t *ghash = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)length);
  uint32_t *grow = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)length);
  if (!slots || !ghash || !grow) { free(slots); free(ghash); free(grow); return HANDLE_ERR("out of memory"); }
  memset(slots, 0xff, sizeof(int32_t) * cap);
  int groups = 0;
  for (int i = 0; i < length; i++) {
    uint8_t row[64];
    pack_dim_row(inputKeys.DimValues, inputKeys.VectorCapacity, &L, (uint32_t)i, row);
    uint32_t h = oracle_murmur3_32(row, L.rowBytes, 0);
    size_t s = h & (cap - 1);
    int g = -1;
    for (;;) {
      if (slots[s] < 0) break;
      if (ghash[slots[s]] == h) { g = slots[s]; break; }
      s = (s + 1) & (cap - 1);
    }
    const uint8_t *v = inputValues + (size_t)vb * i;
    if (g < 0) {
      g = groups++;
      slots[s] = g;
      ghash[g] = h;
      grow[g] = (uint32_t)i;
      memset(outputValues + (size_t)vb * g, 0, vb); /* m_map[key] default-constructs 0 */
    }
    /* m_map[key] = op(m_map[key], value) */
    agg_combine(aggFunc, valueBytes, outputValues + (size_t)vb * g, v);
  }
  int outGroups = 0;
  for (int g = 0; g < groups; g++) {
    /* key == 0 (hash 0 at row 0) is the map's "unused" marker and is dropped on extraction
     * (hash_reduction.cu:146-149) */
    if (ghash[g] == 0 && grow[g] == 0) continue;
    if (outGroups != g) memmove(outputValues + (size_t)vb * outGroups, outputValues + (size_t)vb * g, vb);
    copy_dim_row(inputKeys.DimValues, inputKeys.VectorCapacity, outputKeys.DimValues,
                 inputKeys.VectorCapacity, &L, grow[g], (uint32_t)outGroups);
    outGroups++;
  }
  free(slots); free(ghash); free(grow);
  return HANDLE_OK(outGroups);
}

/* sort_reduce.cu:252-314 */
CGoCallResHandle Expand(DimensionVector inputKeys, DimensionVector outputKeys,
                        uint32_t *baseCounts, uint32_t *indexVector, int indexVectorLen,
                        int outputOccupiedLen, void *cudaStream, int device) {
  (void)cudaStream; (void)device;
  DimLayout L;
  dim_layout(inputKeys.NumDimsPerDimWidth, &L);
  uint64_t total = 0;
  for (int i = 0; i < indexVectorLen; i++)
    total += baseCounts[indexVector[i] + 1] - baseCounts[indexVector[i]];
  int room = outputKeys.VectorCapacity - outputOccupiedLen;
  int outLen = total < (uint64_t)room ? (int)total : room;
  int w = 0;
  /* the expanded row j reads input dim row i, where i is the POSITION in indexVector
   * (scatter_if of counting_iterator + max-scan, sort_reduce.cu:271-288) */
  for (int i = 0; i < indexVectorLen && w < outLen; i++) {
    uint32_t c = baseCounts[indexVector[i] + 1] - baseCounts[indexVector[i]];
    for (uint32_t k = 0; k < c && w < outLen; k++, w++)
      copy_dim_row(inputKeys.DimValues, inputKeys.VectorCapacity, outputKeys.DimValues,
                   outputKeys.VectorCapacity, &L, (uint32_t)i, (uint32_t)(outputOccupiedLen + w));
  }
  return HANDLE_OK(outLen + outputOccupiedLen);
}

/* ---- HyperLogLog (query/hll.cu:62-290) -------------------------------------------------------
 * State carried between batches by the host (query/aql_processor.go:717-723 swaps [0]<->[1] of the
 * dimension / measure / hash vectors; the two index vectors are re-initialised per batch,
 * query/aql_batchexecutor.go:221-224): rows [0, prevResultSize) of prev{Dim,Hash,Values} are the
 * merged (dim hash | register, hll value) entries of the earlier batches, sorted by (hash asc,
 * value desc); rows [prevResultSize, +curBatchSize) of prevDim hold the dimension rows of the
 * current batch, curValuesOut[0, curBatchSize) their hll values (rho << 16 | register), and
 * curDimOut.IndexVector[i] the dimension row of entry i. */
typedef struct { uint64_t h; uint32_t idx; uint32_t val; } HllEntry;

static void hll_merge_sort(HllEntry *a, HllEntry *tmp, int n) { /* stable, key = h */
  if (n < 2) return;
  int m = n / 2;
  hll_merge_sort(a, tmp, m);
  hll_merge_sort(a + m, tmp, n - m);
  int i = 0, j = m, k = 0;
  while (i < m && j < n) tmp[k++] = (a[j].h < a[i].h) ? a[j++] : a[i++];
  while (i < m) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, sizeof(HllEntry) * (size_t)n);
}

/* HLLMergeComparator (functor.hpp:1316-1328): hash ascending, ties by value descending */
static bool hll_less(uint64_t h1, uint32_t v1, uint64_t h2, uint32_t v2) {
  return h1 == h2 ? v1 > v2 : h1 < h2;
}

CGoCallResHandle deviceMalloc(void **devPtr, size_t size);
CGoCallResHandle deviceMemset(void *devPtr, int value, size_t count);

CGoCallResHandle HyperLogLog(DimensionVector prevDimOut, DimensionVector curDimOut,
                             uint32_t *prevValuesOut, uint32_t *curValuesOut, int prevResultSize,
                             int curBatchSize, bool isLastBatch, uint8_t **hllVectorPtr,
                             size_t *hllVectorSizePtr, uint16_t **hllDimRegIDCountPtr,
                             void *cudaStream, int device) {
  (void)cudaStream; (void)device;
  DimLayout L;
  dim_layout(curDimOut.NumDimsPerDimWidth, &L);
  const int P = prevResultSize, n = curBatchSize;

  /* 1. sortCurrentBatch (hll.cu:70-88): key = dim hash with its low 16 bits replaced by the
   *    register id (HLLHashFunctor, functor.hpp:1299-1305); stable sort of (index, value) */
  HllEntry *e = (HllEntry *)malloc(sizeof(HllEntry) * (size_t)(n > 0 ? n : 1) * 2);
  if (!e) return HANDLE_ERR("out of memory");
  for (int i = 0; i < n; i++) {
    uint8_t row[64];
    uint64_t out[2];
    pack_dim_row(prevDimOut.DimValues, curDimOut.VectorCapacity, &L, curDimOut.IndexVector[i], row);
    oracle_murmur3_128(row, L.rowBytes, 0, out);
    e[i].h = (out[0] & 0xFFFFFFFFFFFF0000ull) | (curValuesOut[i] & 0x3FFF);
    e[i].idx = curDimOut.IndexVector[i];
    e[i].val = curValuesOut[i];
  }
  hll_merge_sort(e, e + n, n);
  for (int i = 0; i < n; i++) {
    curDimOut.HashValues[i] = e[i].h;
    curDimOut.IndexVector[i] = e[i].idx;
    curValuesOut[i] = e[i].val;
  }

  /* 2. reduceCurrentBatch (hll.cu:211-231): runs of equal keys -> (first index, max value),
   *    appended behind the previous results */
  int c = 0;
  for (int i = 0; i < n;) {
    int j = i;
    uint32_t best = e[i].val;
    while (j + 1 < n && e[j + 1].h == e[i].h) { j++; if (e[j].val > best) best = e[j].val; }
    prevDimOut.HashValues[P + c] = e[i].h;
    prevDimOut.IndexVector[P + c] = e[i].idx;
    prevValuesOut[P + c] = best;
    c++;
    i = j + 1;
  }
  free(e);

  /* 3. merge (hll.cu:191-209): stable merge of previous and current entries into cur* */
  {
    int i = 0, j = P, k = 0;
    const int endB = P + c;
    while (i < P && j < endB) {
      int take = hll_less(prevDimOut.HashValues[j], prevValuesOut[j], prevDimOut.HashValues[i], prevValuesOut[i]) ? j++ : i++;
      curDimOut.HashValues[k] = prevDimOut.HashValues[take];
      curValuesOut[k] = prevValuesOut[take];
      curDimOut.IndexVector[k] = prevDimOut.IndexVector[take];
      k++;
    }
    for (; i < P; i++, k++) {
      curDimOut.HashValues[k] = prevDimOut.HashValues[i];
      curValuesOut[k] = prevValuesOut[i];
      curDimOut.IndexVector[k] = prevDimOut.IndexVector[i];
    }
    for (; j < endB; j++, k++) {
      curDimOut.HashValues[k] = prevDimOut.HashValues[j];
      curValuesOut[k] = prevValuesOut[j];
      curDimOut.IndexVector[k] = prevDimOut.IndexVector[j];
    }
  }
  int resSize = P + c;

  /* 4. makeHLLVector (hll.cu:233-254, 108-166), last batch only */
  if (isLastBatch && resSize > 0) {
    const uint64_t *H = curDimOut.HashValues;
    /* dimension heads: the first 48 bits of the key differ (HLLDimNotEqualFunctor) */
    int dims = 0;
    for (int i = 0; i < resSize; i++)
      if (i == 0 || (H[i] >> 16) != (H[i - 1] >> 16)) dims++;
    uint16_t *regCount = NULL;
    CGoCallResHandle h = deviceMalloc((void **)&regCount, (size_t)dims * sizeof(uint16_t));
    if (h.pStrErr) return h;
    uint64_t *offsets = (uint64_t *)calloc((size_t)dims + 1, sizeof(uint64_t));
    if (!offsets) return HANDLE_ERR("out of memory");
    /* registers per dimension = distinct keys of the run (HLLRegIDHeadFlagIterator) */
    int d = -1;
    uint32_t count = 0;
    for (int i = 0; i < resSize; i++) {
      if (i == 0 || (H[i] >> 16) != (H[i - 1] >> 16)) {
        if (d >= 0) regCount[d] = (uint16_t)count;
        d++;
        count = 0;
        curDimOut.IndexVector[d] = curDimOut.IndexVector[i]; /* remove_if on the dim heads */
      }
      if (i == 0 || H[i] != H[i - 1]) count++;
    }
    regCount[d] = (uint16_t)count;
    for (int k = 0; k < dims; k++) /* HLLDimByteCountFunctor (functor.hpp:1331-1340) */
      offsets[k + 1] = offsets[k] + (regCount[k] < HLL_DENSE_THRESHOLD ? (uint64_t)regCount[k] * 4 : HLL_DENSE_SIZE);
    uint8_t *vec = NULL;
    h = deviceMalloc((void **)&vec, offsets[dims]);
    if (h.pStrErr) { free(offsets); return h; }
    deviceMemset(vec, 0, offsets[dims]);
    /* CopyHLLFunctor (functor.hpp:1351-1374) through HLLValueOutputIterator (iterator.hpp:1197-1257) */
    d = -1;
    uint32_t regInDim = 0;
    for (int i = 0; i < resSize; i++) {
      if (i == 0 || (H[i] >> 16) != (H[i - 1] >> 16)) { d++; regInDim = 0; }
      if (i == 0 || H[i] != H[i - 1]) {
        const uint32_t value = curValuesOut[i];
        const uint16_t regID = (uint16_t)(value & 0x3FFF);
        const uint8_t rho = (uint8_t)((uint8_t)((value >> 16) & 0xFF) + 1);
        if (regCount[d] < HLL_DENSE_THRESHOLD) {
          const uint32_t w = (uint32_t)rho << 16 | regID;
          memcpy(vec + offsets[d] + (uint64_t)regInDim * 4, &w, 4);
        } else {
          vec[offsets[d] + regID] = rho;
        }
        regInDim++;
      }
    }
    *hllVectorPtr = vec;
    *hllVectorSizePtr = (size_t)offsets[dims];
    *hllDimRegIDCountPtr = regCount;
    free(offsets);
    resSize = dims;
  }

  /* 5. copyDim (hll.cu:169-187): gather the dimension rows of the surviving entries; both
   *    strides are the INPUT capacity, like Reduce */
  for (int i = 0; i < resSize; i++)
    copy_dim_row(prevDimOut.DimValues, prevDimOut.VectorCapacity, curDimOut.DimValues,
                 prevDimOut.VectorCapacity, &L, curDimOut.IndexVector[i], (uint32_t)i);
  return HANDLE_OK(resSize);
}

/* ---- geo intersection (query/geo_intersects.cu, query/iterator.hpp:1260-1452) -------------------
 * GeoShapeBatch.LatLongs = [latitudes f32 x N][longitudes f32 x N][shape index u8 x N]; a polygon
 * is a run of points with the same shape index, (FLT_MAX, FLT_MAX) separates its rings.  Entry i
 * of the index vector owns TotalWords predicate words; bit s is toggled once per polygon edge the
 * horizontal ray from the point crosses (even-odd rule). */

/* the point of entry i and its validity */
static bool geo_point_at(const InputVector *points, const uint32_t *indexVector, int i, GeoPointT *out) {
  memset(out, 0, sizeof(*out));
  if (points->Type == VectorPartyInput) { /* VectorPartyIterator<GeoPointT>, iterator.hpp:291-352 */
    const VectorPartySlice *vp = &points->Vector.VP;
    const uint32_t row = indexVector[i];
    *out = ((const GeoPointT *)(vp->BasePtr + vp->ValuesOffset))[row];
    if (vp->ValuesOffset == 0) return true;
    return get_bit(vp->BasePtr, row + vp->StartingIndex);
  }
  /* RecordIDJoinIterator<GeoPointT>, iterator.hpp:911-930 */
  const ForeignColumnVector *f = &points->Vector.ForeignVP;
  const RecordID rid = f->RecordIDs[i];
  if (rid.batchID && (rid.batchID - f->BaseBatchID < f->NumBatches - 1 ||
                      rid.index < (uint32_t)f->NumRecordsInLastBatch)) {
    const VectorPartySlice *vp = &f->Batches[rid.batchID - f->BaseBatchID];
    if (vp->BasePtr == NULL) {
      *out = f->DefaultValue.Value.GeoPointVal;
      return f->DefaultValue.HasDefault;
    }
    *out = ((const GeoPointT *)(vp->BasePtr + vp->ValuesOffset))[rid.index];
    if (vp->ValuesOffset == 0) return true;
    return get_bit(vp->BasePtr, rid.index + vp->StartingIndex);
  }
  return false;
}

/* GeoPredicateIterator (iterator.hpp:1263-1315): first set bit as an int8 — shapes 128..255 wrap to
 * negative values, which every consumer reads as "no shape" */
static int8_t geo_first_shape(const uint32_t *words, int totalWords) {
  for (int w = 0; w < totalWords; w++)
    for (int b = 0; b < 32; b++)
      if ((words[w] >> b) & 1) return (int8_t)(w * 32 + b);
  return -1;
}

CGoCallResHandle GeoBatchIntersects(GeoShapeBatch geoShapeBatch, InputVector points,
                                    uint32_t *indexVector, int indexVectorLength,
                                    uint32_t startCount, RecordID **recordIDVectors,
                                    int numForeignTables, uint32_t *outputPredicate, bool inOrOut,
                                    void *cudaStream, int device) {
  (void)startCount; (void)cudaStream; (void)device;
  if (points.Type == VectorPartyInput) {
    if (points.Vector.VP.DataType != GeoPoint)
      return HANDLE_ERR("only geo point column are allowed in geo_intersects");
    if (points.Vector.VP.BasePtr == NULL) return HANDLE_OK(0);
  } else if (points.Type == ForeignColumnInput) {
    if (points.Vector.ForeignVP.DataType != GeoPoint)
      return HANDLE_ERR("only geo point column are allowed in geo_intersects");
  } else {
    return HANDLE_ERR("Unsupported data type for geo intersection contexts");
  }
  const int N = geoShapeBatch.TotalNumPoints, W = geoShapeBatch.TotalWords;
  const float *lats = (const float *)geoShapeBatch.LatLongs;
  const float *longs = lats + N;
  const uint8_t *shape = geoShapeBatch.LatLongs + (size_t)N * 8;
  /* calculateBatchIntersection (geo_intersects.cu:246-276): every (entry, polygon point) pair */
  for (int i = 0; i < indexVectorLength; i++) {
    uint32_t *pred = outputPredicate + (size_t)i * W;
    GeoPointT pt;
    const bool ok = geo_point_at(&points, indexVector, i, &pt);
    for (int p = 0; p < N; p++) { /* GeoBatchIntersectIterator::dereference, iterator.hpp:1356-1421 */
      if (p >= N - 1) continue;             /* last point: no edge starts here */
      if (shape[p] != shape[p + 1]) continue; /* last point of a shape */
      if (!ok) { /* a null point: the first edge writes the verdict, nobody toggles */
        if (p == 0)
          for (int w = 0; w < W; w++) pred[w] = !inOrOut;
        continue;
      }
      const float lat1 = lats[p], lat2 = lats[p + 1];
      if (lat1 < FLT_MAX && lat2 < FLT_MAX) {
        const float long1 = longs[p], long2 = longs[p + 1];
        if (((long1 > pt.Long) != (long2 > pt.Long)) &&
            (pt.Lat < (lat2 - lat1) * (pt.Long - long1) / (long2 - long1) + lat1))
          pred[shape[p] / 32] ^= (1u << (shape[p] % 32));
      }
    }
  }
  if (numForeignTables < 0 || numForeignTables > 8) return HANDLE_ERR("only support up to 8 foreign tables");
  /* GeoRemoveFilter (geo_intersects.cu:214-239): drop entry i when inOrOut == "in no shape" */
  int k = 0;
  for (int i = 0; i < indexVectorLength; i++) {
    const bool none = geo_first_shape(outputPredicate + (size_t)i * W, W) < 0;
    if (inOrOut == none) continue;
    indexVector[k] = indexVector[i];
    for (int t = 0; t < numForeignTables; t++) recordIDVectors[t][k] = recordIDVectors[t][i];
    k++;
  }
  return HANDLE_OK(k);
}

/* write_geo_shape_dim (geo_intersects.cu:318-337): the first intersected shape of every entry that
 * is inside some shape, in entry order, as a uint8 dimension with validity 1 */
CGoCallResHandle WriteGeoShapeDim(int shapeTotalWords, DimensionOutputVector dimOut,
                                  int indexVectorLengthBeforeGeo, uint32_t *outputPredicate,
                                  void *cudaStream, int device) {
  (void)cudaStream; (void)device;
  int k = 0;
  for (int i = 0; i < indexVectorLengthBeforeGeo; i++) {
    const int8_t s = geo_first_shape(outputPredicate + (size_t)i * (uint8_t)shapeTotalWords, (uint8_t)shapeTotalWords);
    if (s < 0) continue;
    dimOut.DimValues[k] = (uint8_t)s;
    dimOut.DimNulls[k] = 1;
    k++;
  }
  return HANDLE_OK(0);
}

CGoCallResHandle BootstrapDevice(void) { return HANDLE_OK(0); }
