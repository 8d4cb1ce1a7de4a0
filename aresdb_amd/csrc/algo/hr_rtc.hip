// Per-plan specialised scan / merge kernels of the fused HashReduce, compiled at run time with hiprtc.
//
// The generic hr_fused_scan_kernel (hr_kernels.hpp) interprets the plan: operation codes, kinds and
// constants arrive as kernel arguments and every expression is dispatched per quad.  That costs
// ~210 VALU + ~80 SALU instructions per row, 266 KB of code and several hundred spilled SGPRs — the
// kernel is bound by instruction issue and instruction fetch, not by HBM.  Here the host writes the
// plan's SHAPE out as straight-line code and hiprtc compiles it for the device's architecture.
//
// What is a literal and what is an argument.  Literals (they select instructions): functors, kinds, null
// mask, column slots, aggregate, widening, partition bits, record format — and DIVISORS (the time bucket
// `Floor(ts, 3600)`: the compiler strength-reduces the division; DESIGN.md 3 measured that this is where the
// gain of specialisation is).  Kernel ARGUMENTS: every comparison constant and every + / - / x constant
// (`Args::k`).  AresDB queries carry per-query `from` / `to` time-filter constants
// (query/common/time_filter.go:371-397): a new time range runs the kernel that is already loaded.
//
// Compilation never sits on a query's critical path: a kernel that is not loaded yet is compiled on a
// background thread (ARES_RTC_ASYNC=0: inline) while the caller goes on with the generic kernels — which
// stay the fallback at every step —, code objects are kept in an on-disk cache keyed on (architecture,
// hiprtc version, source) so that a restarted process loads instead of compiling, and the in-memory cache is
// bounded (least recently used shapes are unloaded).
//
// Three scans are generated:
//   * DIRECT, compact lines — high-cardinality queries (more groups than an LDS table holds): every surviving
//     row becomes an 8-byte record {carried measure, (hash << partBits) | row bits}, counting-sorted by
//     partition in LDS; only whole aligned 128-byte lines of 14 records + two 8-byte headers (the low row
//     bits) leave the CU.  9.14 bytes per record instead of 16: the record round trip is what bounds this
//     query shape (the scan moved 1.72x its algorithmic bytes with 16-byte records).
//   * DIRECT, 16-byte records {row, hash, value lo, value hi} in lines of 8 — the vector-sourced scan
//     (HashReduce on materialised dimension / measure vectors, 8-byte values) and batches whose per-workgroup
//     chunk does not fit the compact row field.
//   * TABLE — low-cardinality queries: each workgroup aggregates its rows in an LDS hash table without a
//     barrier in the loop and emits one record per group at the end (region A, read by the generic merge);
//     rows that find the table full spill as single records.
// Write path facts behind the DIRECT kernels (tools/ubench_scatter.hip, profiles/r2_ubench_write_path*.txt):
// reading and hashing the columns runs at 6 TB/s, but a CU retires only one scattered small store per ~4.5
// cycles and HBM write time follows the number of 64-byte write requests — hence the LDS sort and whole lines.
//
// Supported shapes (everything else: generic kernel) — exactly the fast paths of eval_quad /
// compare_tile in fast_eval.hpp, so results are bit-identical:
//   * columns of kind int32 / uint32 / float32 (modes 1 and 2);
//   * dimension / measure: a bare column, or an integer column (Divide | Mod | Floor | Plus | Minus |
//     Multiply) a valid integer constant, stored without a value conversion;
//   * filters: a column compared (==, !=, <, <=, >, >=) with a valid constant in the common kind;
//   * the measure carried as 4 bytes (fused_carry).
// hiprtc is loaded with dlopen: a host without it simply keeps the generic kernel.  ARES_RTC=0: off.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <cerrno>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <fstream>
#include <functional>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "hash_reduce_lds.hpp"
#include "hr_kernels.hpp"
#include "hr_rtc.hpp"

namespace ares {

namespace {

// ---- the minimum of the hiprtc API, resolved at run time ---------------------------------------------
typedef struct _hiprtcProgram *RtcProgram;
struct RtcApi {
  int (*create)(RtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
  int (*compile)(RtcProgram, int, const char **) = nullptr;
  int (*logSize)(RtcProgram, size_t *) = nullptr;
  int (*log)(RtcProgram, char *) = nullptr;
  int (*codeSize)(RtcProgram, size_t *) = nullptr;
  int (*code)(RtcProgram, char *) = nullptr;
  int (*destroy)(RtcProgram *) = nullptr;
  int (*version)(int *, int *) = nullptr;
  bool ok = false;
};
const RtcApi &rtc_api() {
  static const RtcApi api = [] {
    RtcApi a;
    const char *e = getenv("ARES_RTC");
    if (e && e[0] == '0') return a;
    void *h = dlopen("libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libhiprtc.so.7", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return a;
    a.create = reinterpret_cast<decltype(a.create)>(dlsym(h, "hiprtcCreateProgram"));
    a.compile = reinterpret_cast<decltype(a.compile)>(dlsym(h, "hiprtcCompileProgram"));
    a.logSize = reinterpret_cast<decltype(a.logSize)>(dlsym(h, "hiprtcGetProgramLogSize"));
    a.log = reinterpret_cast<decltype(a.log)>(dlsym(h, "hiprtcGetProgramLog"));
    a.codeSize = reinterpret_cast<decltype(a.codeSize)>(dlsym(h, "hiprtcGetCodeSize"));
    a.code = reinterpret_cast<decltype(a.code)>(dlsym(h, "hiprtcGetCode"));
    a.destroy = reinterpret_cast<decltype(a.destroy)>(dlsym(h, "hiprtcDestroyProgram"));
    a.version = reinterpret_cast<decltype(a.version)>(dlsym(h, "hiprtcVersion"));
    a.ok = a.create && a.compile && a.logSize && a.log && a.codeSize && a.code && a.destroy;
    return a;
  }();
  return api;
}

// host twin of cvt32 (device_model.hpp) for constants
uint32_t host_cvt32(uint32_t bits, int from, int to) {
  if (from == to) return bits;
  auto asf = [](uint32_t b) { float f; memcpy(&f, &b, 4); return f; };
  auto fb = [](float f) { uint32_t b; memcpy(&b, &f, 4); return b; };
  switch (to) {
    case K_BOOL: return from == K_F32 ? (asf(bits) != 0.0f) : (bits != 0u);
    case K_I32: return from == K_F32 ? static_cast<uint32_t>(static_cast<int32_t>(asf(bits))) : bits;
    case K_U32: return from == K_F32 ? static_cast<uint32_t>(asf(bits)) : bits;
    default: return from == K_I32 ? fb(static_cast<float>(static_cast<int32_t>(bits))) : fb(static_cast<float>(bits));
  }
}

bool int_kind(int k) { return k == K_I32 || k == K_U32; }
bool col_kind(int k) { return k == K_I32 || k == K_U32 || k == K_F32; }

std::string hex(uint32_t v) {
  char b[16];
  snprintf(b, sizeof(b), "0x%08xu", v);
  return b;
}

// slots of Args::k (run-time constants): filter f -> f, dimension d -> kFusedFilters + d, measure -> the last
constexpr int kNumConsts = kFusedFilters + kFusedDims + 1;
int const_slot_filter(int f) { return f; }
int const_slot_dim(int d) { return kFusedFilters + d; }
int const_slot_measure() { return kFusedFilters + kFusedDims; }
std::string const_name(int slot) { return "a.k[" + std::to_string(slot) + "]"; }

// value expression of one element: writes `r` (result bits) given `v` (stored bits) and `okb` (0/1);
// `kc` names the run-time constant of the expression.  Returns false when the shape is outside the fast
// paths of eval_quad.
bool gen_value(const FastOperands &f, std::ostringstream &o, const char *v, const char *okb, const char *r, const std::string &kc) {
  if (!col_kind(f.akind)) return false;
  const bool intKinds = f.akind != K_F32 && f.I != K_F32 && f.akind != K_BOOL;
  if (f.arity == 1) {
    if (!(f.akind == f.I || intKinds)) return false;
    o << "      " << r << " = " << v << ";\n";  // a null bare column keeps its stored bits (functor.hpp:345-351)
    return true;
  }
  if (f.arity != 2 || !intKinds || !int_kind(f.I) || !int_kind(f.bkind) || !f.bok) return false;
  const uint32_t y = f.bbits;  // cvt32 between the integer kinds keeps the bits
  if (f.divLike) {  // the divisor is a literal: the compiler turns the division into multiply + shift
    const bool sgn = f.I == K_I32;
    const uint32_t mag = (sgn && static_cast<int32_t>(y) < 0) ? 0u - y : y;
    const bool yneg = sgn && static_cast<int32_t>(y) < 0;
    o << "      {\n";
    if (sgn) o << "        const bool xneg = (i32)" << v << " < 0; const u32 ax = xneg ? 0u - " << v << " : " << v << ";\n";
    else o << "        const u32 ax = " << v << ";\n";
    // fast_divmod: d = 0 -> q = r = 0; d = 1 -> q = x, r = 0
    if (mag == 0) o << "        const u32 q = 0u, m = 0u;\n";
    else if (mag == 1) o << "        const u32 q = ax, m = 0u;\n";
    else o << "        const u32 q = ax / " << hex(mag) << ", m = ax % " << hex(mag) << ";\n";
    if (sgn) {
      o << "        const u32 sq = (xneg != " << (yneg ? "true" : "false") << ") ? 0u - q : q;\n";
      o << "        const u32 sm = xneg ? 0u - m : m;\n";
    } else {
      o << "        const u32 sq = q, sm = m;\n";
    }
    o << "        " << r << " = " << (f.functor == Divide ? "sq" : f.functor == Mod ? "sm" : std::string(v) + " - sm") << ";\n";
    o << "        if (!" << okb << ") " << r << " = 0u;\n      }\n";
    return true;
  }
  if (f.functor == Plus || f.functor == Minus || f.functor == Multiply) {
    o << "      " << r << " = " << okb << " ? (" << v << (f.functor == Plus ? " + " : f.functor == Minus ? " - " : " * ") << kc
      << ") : 0u;\n";
    return true;
  }
  return false;
}
// what Args::k holds for a value expression
uint32_t value_const(const FastOperands &f) { return f.bbits; }

// keep bit of one element for one filter; the constant (converted to the common kind by the host) is `kc`
bool gen_compare(const FastOperands &f, std::ostringstream &o, const char *v, const char *okb, const char *keep, const std::string &kc) {
  if (!col_kind(f.akind) || f.arity != 2) return false;
  const bool sameBits = f.akind == f.I || (f.akind != K_F32 && f.I != K_F32 && f.akind != K_BOOL);
  if (!sameBits || !f.bok) return false;
  if (f.functor < Equal || f.functor > GreaterThanOrEqual) return false;
  if (!(f.I == K_F32 || f.I == K_I32 || f.I == K_U32)) return false;
  const char *op = f.functor == Equal ? "==" : f.functor == NotEqual ? "!=" : f.functor == LessThan ? "<"
                   : f.functor == LessThanOrEqual ? "<=" : f.functor == GreaterThan ? ">" : ">=";
  if (f.I == K_F32) o << "      " << keep << " &= (" << okb << " && (__uint_as_float(" << v << ") " << op << " __uint_as_float(" << kc << "))) ? 1u : 0u;\n";
  else if (f.I == K_I32) o << "      " << keep << " &= (" << okb << " && ((i32)" << v << " " << op << " (i32)" << kc << ")) ? 1u : 0u;\n";
  else o << "      " << keep << " &= (" << okb << " && (" << v << " " << op << " " << kc << ")) ? 1u : 0u;\n";
  return true;
}
uint32_t compare_const(const FastOperands &f) { return host_cvt32(f.bbits, f.bkind, f.I); }

bool plain_store(int rk, int outKind) { return rk == outKind || (rk != K_F32 && outKind != K_F32 && rk != K_BOOL); }

// ARES_HR_PHASES=1: the generated kernels time-stamp their phases (diagnostics; a different source text, so
// a separate cache entry)
static bool phases_enabled() {
  static const bool on = [] {
    const char *e = getenv("ARES_HR_PHASES");
    return e && e[0] == '1';
  }();
  return on;
}

// ARES_HR_SCAN_TILES=1|2: tiles of 4096 rows per pass of the compact scan (kernel_body_compact / kernel_body_compact2)
static int scan_tiles() {
  static const int n = [] {
    const char *e = getenv("ARES_HR_SCAN_TILES");
    return e && e[0] == '2' ? 2 : 1;
  }();
  return n;
}

// ARES_HR_NT: non-temporal loads for the columns (read once; 4 = loads only, the default) and / or non-temporal stores for the
// record lines (written once, read by another kernel; 2 = stores only, 3 or 1 = both, 0 = neither).  Round 3 measured the
// pair 4 % SLOWER than neither; round 4 measured them one at a time (profiles/r4_experiments.md): loads only, scan
// 0.392 -> 0.379 ms per 64 Mi rows with the merge 0.280 -> 0.287; stores only, scan 0.397, merge 0.273.
static int nt_mode() {  // bit 0: column loads, bit 1: line stores ("1" of round 3 = both = 3)
  static const int v = [] {
    const char *e = getenv("ARES_HR_NT");
    const int x = e ? atoi(e) : 4;  // default: streaming column loads, ordinary line stores
    return x == 1 ? 3 : x == 4 ? 1 : x;
  }();
  return v;
}
static bool nt_enabled() { return (nt_mode() & 1) != 0; }
static bool nt_stores_enabled() { return (nt_mode() & 2) != 0; }

// ARES_HR_SCAN_OPT: bit mask of compact-scan variants under measurement (a different source text per value, so they
// can be timed against each other in one process tree: tools/gpu_r3_ab.sh, profiles/r3_experiments.md)
//   1  a partition that completes no line in a tile scatters its records straight into its remainder
//   2  murmur's h * 5 + c as shift-add + add instead of the v_mad_u64_u32 the compiler picks
//   4  compact scan: the 13 slots behind a partition's last line are moved to its remainder by two lanes (sixteen wavefronts) instead of one
static unsigned scan_opt() {
  static const unsigned v = [] {
    const char *e = getenv("ARES_HR_SCAN_OPT");
    return e && e[0] ? static_cast<unsigned>(atoi(e)) : 3u;
  }();
  return v;
}

struct RtcArgs {  // mirrors `struct Args` of the generated source (args_text below): pointers, 8-byte, then 4-byte fields
  const uint32_t *vals[kFusedCols];
  const uint8_t *nulls[kFusedCols];
  uint32_t *recB;
  uint32_t *countsB;
  uint32_t *overflow;
  uint64_t *phases;
  uint4 *recA;
  uint32_t *cursorsA;
  uint64_t capA;
  uint32_t bitOff[kFusedCols];
  uint32_t rowBase;
  int length;
  uint32_t capB;
  uint32_t chunkTiles;
  uint32_t k[kNumConsts];
  uint32_t pad;
};
static_assert(sizeof(RtcArgs) % 8 == 0, "Args is passed as one buffer");

std::string args_text() {
  std::ostringstream o;
  o << "struct Args { const u32 *vals[" << kFusedCols << "]; const u8 *nulls[" << kFusedCols << "]; u32 *recB; u32 *countsB; u32 *overflow; u64 *phases;\n"
       "              uint4 *recA; u32 *cursorsA; u64 capA; u32 bitOff[" << kFusedCols << "]; u32 rowBase; int length; u32 capB; u32 chunkTiles;\n"
       "              u32 k[" << kNumConsts << "]; u32 pad; };\n";
  return o.str();
}

const char *kPrelude =
    "typedef unsigned int u32; typedef unsigned long long u64; typedef unsigned char u8; typedef unsigned short u16; typedef int i32; typedef long long i64;\n"
    "struct __attribute__((packed, aligned(1))) PU32x4 { u32 v[4]; };\n"
    "struct __attribute__((packed, aligned(1))) PU32 { u32 v; };\n"
    "struct __attribute__((packed, aligned(1))) PU16 { u16 v; };\n"
    "typedef u32 U4 __attribute__((ext_vector_type(4)));\n"
    "typedef U4 U4a __attribute__((aligned(4)));\n"
    "struct __attribute__((packed, aligned(1))) PU32x2 { u32 v[2]; };\n"
    "typedef u32 U2 __attribute__((ext_vector_type(2)));\n"
    "typedef U2 U2a __attribute__((aligned(2)));\n"
    "typedef u32 U1a __attribute__((aligned(1)));\n"
    "__device__ __forceinline__ u32 rotl(u32 x, int r) { return (x << r) | (x >> (32 - r)); }\n"
    "__device__ __forceinline__ u32 mix(u32 h, u32 k) { k *= 0xcc9e2d51u; k = rotl(k, 15) * 0x1b873593u; h ^= k; return TIMES5(rotl(h, 13)) + 0xe6546b64u; }\n";

// h * 5 + c: the compiler folds it into one v_mad_u64_u32 (a 64-bit, slow-rate multiply-add); a shift-add and an add are
// two full-rate instructions (ARES_HR_SCAN_OPT bit 2)
std::string times5_text() {
  return scan_opt() & 2u ? "__device__ __forceinline__ unsigned int times5(unsigned int r) { unsigned int t; asm(\"v_lshl_add_u32 %0, %1, 2, %1\" : \"=v\"(t) : \"v\"(r)); return t; }\n"
                           "#define TIMES5(r) times5(r)\n"
                         : "#define TIMES5(r) ((r) * 5u)\n";
}

void phase_macros(std::ostringstream &o) {
  if (phases_enabled())
    o << "#define PH_DECL u64 phT[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u64 phLast = __builtin_readcyclecounter();\n"
         "#define PH(k) { const u64 now = __builtin_readcyclecounter(); phT[k] += now - phLast; phLast = now; }\n"
         "#define PH_OUT if (threadIdx.x == 0u) for (int k = 0; k < 8; k++) a.phases[(u64)blockIdx.x * 8u + k] = phT[k];\n";
  else
    o << "#define PH_DECL\n#define PH(k)\n#define PH_OUT\n";
}

// value of a carried measure (hr::widen_value)
bool gen_widen(std::ostringstream &o, const hr::Widen &w) {
  o << "__device__ __forceinline__ u64 widen(u32 raw) {\n";
  if (w.mode == 0) o << "  return raw;\n";
  else if (w.dtype == Float64)
    o << (w.rk == K_F32 ? "  return (u64)__double_as_longlong((double)__uint_as_float(raw));\n"
          : w.rk == K_I32 ? "  return (u64)__double_as_longlong((double)(i32)raw);\n"
                          : "  return (u64)__double_as_longlong((double)raw);\n");
  else
    o << (w.rk == K_F32 ? "  return (u64)(i64)__uint_as_float(raw);\n" : w.rk == K_I32 ? "  return (u64)(i64)(i32)raw;\n" : "  return (u64)(i64)raw;\n");
  o << "}\n";
  return true;
}

// the aggregate on an LDS slot (hr::lds_aggregate)
bool gen_agg(std::ostringstream &o, const AggSpec &a) {
  o << "__device__ __forceinline__ void agg(u64 *slot, u64 bits) {\n";
  switch (a.vtype) {
    case V_F64: o << "  __hip_atomic_fetch_add(reinterpret_cast<double *>(slot), __longlong_as_double((long long)bits), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"; break;
    case V_U64: case V_I64: o << "  __hip_atomic_fetch_add(slot, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"; break;
    case V_F32: o << "  __hip_atomic_fetch_add(reinterpret_cast<float *>(slot), __uint_as_float((u32)bits), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"; break;
    case V_U32:
      o << "  __hip_atomic_fetch_" << (a.op == OP_SUM ? "add" : a.op == OP_MIN ? "min" : "max")
        << "(reinterpret_cast<u32 *>(slot), (u32)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
      break;
    case V_I32:
      o << "  __hip_atomic_fetch_" << (a.op == OP_SUM ? "add" : a.op == OP_MIN ? "min" : "max")
        << "(reinterpret_cast<i32 *>(slot), (i32)(u32)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n";
      break;
    default: return false;
  }
  o << "}\n";
  char ident[32];
  snprintf(ident, sizeof(ident), "0x%016llxull", static_cast<unsigned long long>(a.identity));
  o << "#define IDENT " << ident << "\n";
  return true;
}

// ---- DIRECT scan, 16-byte records in lines of 8 ------------------------------------------------------
// `Raw`, `load_full`, `load_tail` and `eval4(R, a, i0, hh, cv, cw, alive)` are already in `o`; `fourth` is the
// record's fourth word.  One 1024-lane workgroup per CU walks 4096-row tiles (tile = blockIdx + k * grid).
// Records are counting-sorted by partition in LDS and ONLY whole lines of 8 records leave the CU, each written
// by 8 adjacent lanes with one store; the < 8 records a partition has left over stay in LDS and go first in
// the next tile's lines.  Streams are private to the workgroup: no global atomics.
// part: the partition of row j's hash (default: its top PB bits)
static void kernel_body_lines16(std::ostringstream &o, const char *fourth, const char *entry = "hr_scan_rtc",
                                const char *part = "(PB ? hh[j] >> (32 - (PB ? PB : 1)) : 0u)") {
  phase_macros(o);
  o << "#define T 4096u\n"
       "__device__ __forceinline__ u32 lane_up(u32 v, u32 lane, u32 off) { return (u32)__builtin_amdgcn_ds_bpermute((int)((lane - off) << 2), (int)v); }\n"
       "extern \"C\" __global__ void __launch_bounds__(1024) " << entry << "(Args a) {\n"
       "  __shared__ uint4 sRec[T];\n"            // the tile's records, sorted by partition
       "  __shared__ uint4 sLeft[NP * 7u];\n"     // up to 7 records per partition waiting for a full line
       "  __shared__ u32 sCount[2][NP];\n"
       "  __shared__ u32 sStart[NP], sLeftN[NP], sCursor[NP];\n"
       "  __shared__ u32 sLines[(T + NP * 7u) / 8u + 1u];\n"
       "  __shared__ u32 sWave[16];\n"
       "  __shared__ u32 sTotalLines;\n"
       "  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;\n"
       "  for (u32 p = tid; p < NP; p += 1024u) { sCount[0][p] = 0u; sCount[1][p] = 0u; sLeftN[p] = 0u; sCursor[p] = 0u; }\n"
       "  __syncthreads();\n"
       "  uint4 *myB = reinterpret_cast<uint4 *>(a.recB) + (u64)blockIdx.x * NP * a.capB;\n"
       "  const u32 numTiles = ((u32)a.length + T - 1u) / T;\n"
       "  u32 tile = blockIdx.x, par = 0u;\n"
       "  Raw R;\n"
       "  PH_DECL\n"
       "  load_tile(R, a, tile * T + tid * 4u);\n"
       "  while (tile < numTiles) {\n"
       "    u32 i0 = tile * T + tid * 4u;\n"          // eval4p moves it to the first row the lane's registers hold
       "    u32 hh[4], cv[4], cw[4], alive[4], rank[4];\n"
       "    const u32 next = tile + gridDim.x;\n"
       "    eval4p(R, a, i0, hh, cv, cw, alive, next * T + tid * 4u);\n"
       "    u32 *cnt = sCount[par];\n"
       "#pragma unroll\n"
       "    for (int j = 0; j < 4; j++) {\n"
       "      rank[j] = 0u;\n"
       "      if (alive[j]) rank[j] = __hip_atomic_fetch_add(&cnt[" << part << "], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "    }\n"
       "    __syncthreads();\n"
       "    PH(0)\n"
       // exclusive scans of (new records, whole lines) per partition, packed in one word
       "    u32 myCount = 0u, myLeft = 0u;\n"
       "    if (tid < NP) { myCount = cnt[tid]; myLeft = sLeftN[tid]; }\n"
       "    const u32 myHave = myCount + myLeft, myLines = myHave >> 3;\n"
       "    const u32 packed = (myCount << 16) | myLines;\n"
       "    u32 incl = packed;\n"
       "#pragma unroll\n"
       "    for (u32 off = 1u; off < 64u; off <<= 1) { const u32 t = lane_up(incl, lane, off); if (lane >= off) incl += t; }\n"
       "    if (lane == 63u) sWave[wave] = incl;\n"
       "    __syncthreads();\n"
       "    u32 before = 0u;\n"
       "#pragma unroll\n"
       "    for (u32 w = 0u; w < (NP + 63u) / 64u; w++) { const u32 t = sWave[w]; before += w < wave ? t : 0u; }\n"
       "    const u32 excl = before + incl - packed;\n"
       "    const u32 myStart = excl >> 16, myLineStart = excl & 0xFFFFu;\n"
       "    if (tid < NP) {\n"
       "      sStart[tid] = myStart;\n"
       "      for (u32 c = 0u; c < myLines; c++) sLines[myLineStart + c] = tid | (c << 9);\n"
       "      if (tid == NP - 1u) sTotalLines = myLineStart + myLines;\n"
       "    }\n"
       "    __syncthreads();\n"
       "    PH(1)\n"
       "#pragma unroll\n"
       "    for (int j = 0; j < 4; j++)\n"
       "      if (alive[j]) sRec[sStart[" << part << "] + rank[j]] = make_uint4(a.rowBase + i0 + j, hh[j], cv[j], " << fourth << ");\n"
       "    __syncthreads();\n"
       "    PH(2)\n"
       // whole lines: 8 adjacent lanes write the 8 records of one aligned 128-byte line with one store;
       // four lines per lane are in flight (the LDS look-ups of a line depend on one another)
       "    const u32 totalLines = sTotalLines;\n"
       "    for (u32 L0 = tid >> 3; L0 < totalLines; L0 += 512u) {\n"
       "      const u32 q = tid & 7u;\n"
       "      u32 e[4], lf[4], st[4], cu[4];\n"
       "#pragma unroll\n"
       "      for (u32 j = 0u; j < 4u; j++) { const u32 L = L0 + j * 128u; e[j] = L < totalLines ? sLines[L] : 0u; }\n"
       "#pragma unroll\n"
       "      for (u32 j = 0u; j < 4u; j++) { const u32 p = e[j] & 511u; lf[j] = sLeftN[p]; st[j] = sStart[p]; cu[j] = sCursor[p]; }\n"
       "      uint4 rec[4];\n"
       "#pragma unroll\n"
       "      for (u32 j = 0u; j < 4u; j++) {\n"
       "        const u32 p = e[j] & 511u, idx = (e[j] >> 9) * 8u + q;\n"
       "        const uint4 *src = idx < lf[j] ? sLeft + p * 7u + idx : sRec + ((st[j] + idx - lf[j]) & (T - 1u));\n"
       "        rec[j] = *src;\n"
       "      }\n"
       "#pragma unroll\n"
       "      for (u32 j = 0u; j < 4u; j++) {\n"
       "        const u32 p = e[j] & 511u, at = cu[j] + (e[j] >> 9) * 8u + q;\n"
       "        if (L0 + j * 128u < totalLines && at < a.capB) myB[(u64)p * a.capB + at] = rec[j];\n"
       "      }\n"
       "    }\n"
       "    __syncthreads();\n"
       "    PH(3)\n"
       // what is left of each partition (< 8 records) moves to its LDS remainder; cursors advance
       "    if (tid < NP) {\n"
       "      const u32 rem = myHave & 7u;\n"
       "      const u32 n = myLines ? rem : myCount;\n"
       "      const uint4 *src = sRec + (myLines ? myStart + myLines * 8u - myLeft : myStart);\n"
       "      uint4 *dst = sLeft + tid * 7u + (myLines ? 0u : myLeft);\n"
       "      uint4 t[7];\n"
       "#pragma unroll\n"
       "      for (u32 k = 0u; k < 7u; k++) t[k] = src[k < n ? k : 0u];\n"  // loads first, then stores: one LDS round trip
       "#pragma unroll\n"
       "      for (u32 k = 0u; k < 7u; k++) if (k < n) dst[k] = t[k];\n"
       "      sLeftN[tid] = rem;\n"
       "      u32 cur = sCursor[tid] + myLines * 8u;\n"
       "      if (cur > a.capB) { *a.overflow = 1u; cur = a.capB; }\n"
       "      sCursor[tid] = cur;\n"
       "      cnt[tid] = 0u;\n"  // this counter set is used again two tiles from now
       "    }\n"
       "    par ^= 1u;\n"
       "    tile = next;\n"
       "    PH(4)\n"
       "  }\n"
       "  __syncthreads();\n"
       "  PH(5)\n"
       // the remainders go out as one last line each, padded with null records (row = ~0) the merge skips
       "  for (u32 p = tid >> 3; p < NP; p += 128u) {\n"
       "    const u32 left = sLeftN[p], j = tid & 7u, cur = sCursor[p];\n"
       "    const bool fits = cur + 8u <= a.capB;\n"
       "    if (left && fits) myB[(u64)p * a.capB + cur + j] = j < left ? sLeft[p * 7u + j] : make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);\n"
       "    if (left && !fits) *a.overflow = 1u;\n"
       "    if (j == 0u) a.countsB[(u64)blockIdx.x * NP + p] = (left && fits) ? cur + 8u : cur;\n"
       "  }\n"
       "  PH(6)\n"
       "  PH_OUT\n"
       "}\n";
}

// ---- DIRECT scan, compact lines ------------------------------------------------------------------------
// (format: hr::Workspace::lineRecords == 14.)  Workgroup g scans the contiguous chunk of a.chunkTiles tiles that
// starts at tile g * a.chunkTiles, so that a row is identified by (stream, row within the chunk): 9 of those bits
// travel in the line's header, the rest in the low PB bits of the record's hash word — the PB partition bits of
// the hash are implied by the stream.  The host guarantees chunkTiles * 4096 <= 1 << (PB + 9).
// LDS: records as 8-byte units + a 2-byte array of low row bits; a line is written by 16 adjacent lanes (8 bytes
// each: lanes 0 and 8 the headers), LPL lines per lane in flight; the header of a half-line is the OR of its
// seven lanes' shifted row bits (three DPP steps inside the 8-lane group).
static void kernel_body_compact(std::ostringstream &o) {
  phase_macros(o);
  const bool direct = scan_opt() & 1u;
  const bool pairCopy = direct && (scan_opt() & 4u);  // a partition's 13 slots behind its last line are moved by TWO lanes
  o << (nt_stores_enabled() ? "#define STORE_LINE(p, v) __builtin_nontemporal_store((u64)(v), (p))\n" : "#define STORE_LINE(p, v) (*(p) = (v))\n");
  o << "#define T 4096u\n#define LR 14u\n#define LEFT 13u\n#define LPL 5u\n"
       "__device__ __forceinline__ u32 lane_up(u32 v, u32 lane, u32 off) { return (u32)__builtin_amdgcn_ds_bpermute((int)((lane - off) << 2), (int)v); }\n"
       // OR over the 8 lanes of a half-line: xor 1, xor 2 (quad permutes), then the mirrored quad (row_half_mirror)
       "__device__ __forceinline__ u32 or8(u32 v) {\n"
       "  v |= (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);\n"
       "  v |= (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);\n"
       "  v |= (u32)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);\n"
       "  return v;\n"
       "}\n"
       // inclusive scan over the wavefront: Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8; lanes without a
       // source keep the 0), then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3
       "__device__ __forceinline__ u32 wave_incl_scan(u32 v) {\n"
       "  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);\n"
       "  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);\n"
       "  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);\n"
       "  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);\n"
       "  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);\n"
       "  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);\n"
       "  return v;\n"
       "}\n"
       "extern \"C\" __global__ void __launch_bounds__(1024) hr_scan_rtc(Args a) {\n"
       "  __shared__ u64 sRec[T + NP * LEFT];\n"   // the tile's records sorted by partition, then up to 13 records per partition waiting for a full line
       "  __shared__ u16 sLo[T + NP * LEFT];\n"    // their low 9 row bits
       "  __shared__ u32 sCount[2][NP];\n"
       "  __shared__ u32 sStart[NP];\n"            // where the tile's records of a partition go
    << (pairCopy ? "  __shared__ u32 sMove[NP];\n" : "")  // first of the 13 slots behind a partition's last line of this tile (~0: no line)
    << ""
       "  __shared__ uint2 sLines[(T + NP * LEFT) / LR + 2u];\n"   // {partition | line of the tile << 9 | leftovers << 18, first slot | stream cursor << 13}
       "  __shared__ u32 sWave[16];\n"
       "  __shared__ u32 sTotalLines;\n"
       "  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;\n"
       "  for (u32 p = tid; p < NP; p += 1024u) { sCount[0][p] = 0u; sCount[1][p] = 0u; }\n"
       "  u32 myLeftN = 0u, myCursor = 0u;\n"    // thread p < NP keeps partition p's leftover count and stream cursor in registers
       "  __syncthreads();\n"
       "  u64 *myB = reinterpret_cast<u64 *>(a.recB) + (u64)blockIdx.x * NP * a.capB * 16u;\n"  // capB: lines per stream
       "  const u32 numTiles = ((u32)a.length + T - 1u) / T;\n"
       "  const u32 firstTile = blockIdx.x * a.chunkTiles;\n"
       "  const u32 endTile = firstTile + a.chunkTiles < numTiles ? firstTile + a.chunkTiles : numTiles;\n"
       "  u32 tile = firstTile, par = 0u;\n"
       "  Raw R;\n"
       "  PH_DECL\n"
       "  load_tile(R, a, tile * T + tid * 4u);\n"
       // the lane's place in a line: 16 lanes per line, lanes 0 and 8 carry the two headers
       "  const u32 q = tid & 15u, r8 = q & 7u;\n"
       "  const u32 kk = (q >> 3) * 7u + (r8 ? r8 - 1u : 0u);\n"  // record of the line this lane carries
       "  const u32 sh = r8 ? 9u * (r8 - 1u) : 0u;\n"
       "  while (tile < endTile) {\n"
       "    u32 i0 = tile * T + tid * 4u;\n"          // eval4p moves it to the first row the lane's registers hold
       "    u32 hh[4], cv[4], cw[4], alive[4], rank[4];\n"
       "    const u32 next = tile + 1u;\n"
       "    eval4p(R, a, i0, hh, cv, cw, alive, next * T + tid * 4u);\n"
       "    const u32 rc0 = i0 - firstTile * T;\n"    // row within the chunk
       "    u32 *cnt = sCount[par];\n"
       "#pragma unroll\n"
       "    for (int j = 0; j < 4; j++) {\n"
       "      rank[j] = 0u;\n"
       "      if (alive[j]) rank[j] = __hip_atomic_fetch_add(&cnt[hh[j] >> (32 - PB)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "    }\n"
       "    __syncthreads();\n"
       "    PH(0)\n"
       // exclusive scans of (new records, whole lines) per partition, packed in one word
       "    u32 myCount = 0u, myLeft = 0u;\n"
       "    if (tid < NP) { myCount = cnt[tid]; myLeft = myLeftN; }\n"
       "    const u32 myHave = myCount + myLeft, myLines = myHave / LR;\n"
       "    const u32 packed = (myCount << 16) | myLines;\n"
       "    const u32 incl = wave_incl_scan(packed);\n"
       "    if (lane == 63u) sWave[wave] = incl;\n"
       "    __syncthreads();\n"
       "    u32 before = 0u;\n"
       "#pragma unroll\n"
       "    for (u32 w = 0u; w < (NP + 63u) / 64u; w++) { const u32 t = sWave[w]; before += w < wave ? t : 0u; }\n"
       "    const u32 excl = before + incl - packed;\n"
       "    const u32 myStart = excl >> 16, myLineStart = excl & 0xFFFFu;\n"
       "    if (tid < NP) {\n"
    // a partition that completes no line in this tile takes its records straight into its remainder
    << (direct ? "      sStart[tid] = myLines ? myStart : T + tid * LEFT + myLeft;\n" : "      sStart[tid] = myStart;\n")
    << (pairCopy ? "      sMove[tid] = myLines ? myStart + myLines * LR - myLeft : 0xFFFFFFFFu;\n" : "")
    << "      for (u32 c = 0u; c < myLines; c++) sLines[myLineStart + c] = make_uint2(tid | (c << 9) | (myLeft << 18), myStart | (myCursor << 13));\n"
       "      if (tid == NP - 1u) sTotalLines = myLineStart + myLines;\n"
       "    }\n"
       "    __syncthreads();\n"
       "    PH(1)\n"
       "#pragma unroll\n"
       "    for (int j = 0; j < 4; j++)\n"
       "      if (alive[j]) {\n"
       "        const u32 at = sStart[hh[j] >> (32 - PB)] + rank[j], rc = rc0 + (u32)j;\n"
       "        sRec[at] = ((u64)((hh[j] << PB) | (rc >> 9)) << 32) | cv[j];\n"
       "        sLo[at] = (u16)(rc & 511u);\n"
       "      }\n"
       "    __syncthreads();\n"
       "    PH(2)\n"
       "    const u32 totalLines = sTotalLines;\n"
       "    for (u32 L0 = tid >> 4; L0 < totalLines; L0 += 64u * LPL) {\n"
       "      u32 e[LPL], lf[LPL], st[LPL], cu[LPL], lo[LPL];\n"
       "      u64 rec[LPL];\n"
       "#pragma unroll\n"
       "      for (u32 j = 0u; j < LPL; j++) {\n"
       "        const u32 L = L0 + j * 64u;\n"
       "        const uint2 w = L < totalLines ? sLines[L] : make_uint2(0u, 0u);\n"
       "        e[j] = w.x & 0x3FFFFu; lf[j] = w.x >> 18; st[j] = w.y & 0x1FFFu; cu[j] = w.y >> 13;\n"
       "      }\n"
       "#pragma unroll\n"
       "      for (u32 j = 0u; j < LPL; j++) {\n"
       "        const u32 p = e[j] & 511u, idx = (e[j] >> 9) * LR + kk;\n"
       "        const u32 at = idx < lf[j] ? T + p * LEFT + idx : st[j] + idx - lf[j];\n"
       "        rec[j] = sRec[at];\n"
       "        lo[j] = sLo[at];\n"
       "      }\n"
       "#pragma unroll\n"
       "      for (u32 j = 0u; j < LPL; j++) {\n"
       "        const u64 mine = r8 ? (u64)lo[j] << sh : 0ull;\n"
       "        const u64 hdr = ((u64)or8((u32)(mine >> 32)) << 32) | or8((u32)mine);\n"
       "        const u32 p = e[j] & 511u, line = cu[j] + (e[j] >> 9);\n"
       "        if (L0 + j * 64u < totalLines && line < a.capB) STORE_LINE(&myB[((u64)p * a.capB + line) * 16u + q], r8 ? rec[j] : hdr);\n"
       "      }\n"
       "    }\n"
       "    __syncthreads();\n"
       "    PH(3)\n"
       // what is left of each partition (< 14 records) moves to its LDS remainder; cursors advance
       "    if (tid < NP) {\n"
       "      const u32 rem = myHave - myLines * LR;\n"
    << (pairCopy ? ""  // (moved by all lanes, below)
       : direct ?  // only after a line: the 13 slots behind the last line go to the remainder as they are (those past `rem` are never read)
                 "      if (myLines) {\n"
                 "        const u32 from = myStart + myLines * LR - myLeft, to = T + tid * LEFT;\n"
                 "        u64 t[LEFT]; u16 tl[LEFT];\n"
                 "#pragma unroll\n"
                 "        for (u32 k = 0u; k < LEFT; k++) { t[k] = sRec[from + k]; tl[k] = sLo[from + k]; }\n"
                 "#pragma unroll\n"
                 "        for (u32 k = 0u; k < LEFT; k++) { sRec[to + k] = t[k]; sLo[to + k] = tl[k]; }\n"
                 "      }\n"
               : "      const u32 n = myLines ? rem : myCount;\n"
                 "      const u32 from = myLines ? myStart + myLines * LR - myLeft : myStart;\n"
                 "      const u32 to = T + tid * LEFT + (myLines ? 0u : myLeft);\n"
                 "      u64 t[LEFT]; u16 tl[LEFT];\n"
                 "#pragma unroll\n"
                 "      for (u32 k = 0u; k < LEFT; k++) { const u32 s = from + (k < n ? k : 0u); t[k] = sRec[s]; tl[k] = sLo[s]; }\n"
                 "#pragma unroll\n"
                 "      for (u32 k = 0u; k < LEFT; k++) if (k < n) { sRec[to + k] = t[k]; sLo[to + k] = tl[k]; }\n")
    << "      myLeftN = rem;\n"
       "      u32 cur = myCursor + myLines;\n"
       "      if (cur > a.capB) { *a.overflow = 1u; cur = a.capB; }\n"
       "      myCursor = cur;\n"
       "      cnt[tid] = 0u;\n"  // this counter set is used again two tiles from now
       "    }\n"
    << (pairCopy ?  // lanes 2p and 2p + 1 move slots 0..6 and 7..12 of partition p: sixteen wavefronts share what eight did
                   "    if (tid < 2u * NP) {\n"
                   "      const u32 p = tid >> 1, k0 = (tid & 1u) * 7u, from = sMove[p];\n"
                   "      if (from != 0xFFFFFFFFu) {\n"
                   "        const u32 to = T + p * LEFT;\n"
                   "        u64 t[7]; u16 tl[7];\n"
                   "#pragma unroll\n"
                   "        for (u32 k = 0u; k < 7u; k++) { const u32 kk2 = k0 + k < LEFT ? k0 + k : 0u; t[k] = sRec[from + kk2]; tl[k] = sLo[from + kk2]; }\n"
                   "#pragma unroll\n"
                   "        for (u32 k = 0u; k < 7u; k++) if (k0 + k < LEFT) { sRec[to + k0 + k] = t[k]; sLo[to + k0 + k] = tl[k]; }\n"
                   "      }\n"
                   "    }\n"
                 : "")
    << "    par ^= 1u;\n"
       "    tile = next;\n"
       "    PH(4)\n"
       "  }\n"
       "  __syncthreads();\n"
       "  if (tid < NP) sLines[tid] = make_uint2(myLeftN, myCursor);\n"
       "  __syncthreads();\n"
       "  PH(5)\n"
       // the remainders go out as one last, partly filled line each; countsB holds the exact number of records
       "  for (u32 p = tid >> 4; p < NP; p += 64u) {\n"
       "    const uint2 m = sLines[p];\n"
       "    const u32 left = m.x, cur = m.y;\n"
       "    const bool fits = cur < a.capB, has = r8 && kk < left;\n"
       "    const u64 rec = has ? sRec[T + p * LEFT + kk] : 0ull;\n"
       "    const u64 mine = has ? (u64)sLo[T + p * LEFT + kk] << sh : 0ull;\n"
       "    const u64 hdr = ((u64)or8((u32)(mine >> 32)) << 32) | or8((u32)mine);\n"
       "    if (left && fits) STORE_LINE(&myB[((u64)p * a.capB + cur) * 16u + q], r8 ? rec : hdr);\n"
       "    if (left && !fits) *a.overflow = 1u;\n"
       "    if (q == 0u) a.countsB[(u64)blockIdx.x * NP + p] = cur * LR + ((left && fits) ? left : 0u);\n"
       "  }\n"
       "  PH(6)\n"
       "  PH_OUT\n"
       "}\n";
}

// ---- TABLE scan ----------------------------------------------------------------------------------------
// Low-cardinality queries: every workgroup aggregates its rows in an LDS hash table (key = hash << 32 | lowest
// row, 8-byte value) and emits one 16-byte record per group {row, hash, value} into region A at the end — the
// layout hr::flush_table writes and hr::merge_body reads.  No barrier inside the loop: the wavefronts run free,
// two tiles per wavefront in flight (two register buffers, each refilled column by column while it is evaluated:
// with two or three columns a single tile per wavefront leaves too few bytes in flight to cover HBM latency).
// The table is the specialised merge's: buckets of four keys (two 16-byte LDS reads).  A row first looks at its
// home bucket with straight-line code — it meets its group there nearly always once the groups exist: one LDS
// atomic more —; rows that do not are queued per wavefront in LDS and taken through the general probe loop 64 at a
// time, every lane busy.  Once the table holds LIMIT groups a row whose group finds no slot is written as a single
// record straight away (one global cursor reservation): always correct, slow when frequent — the host sends
// queries with that many groups to the DIRECT kernels.
// Two tiles per pass (ARES_HR_SCAN_TILES=2): the per-partition work of a pass — the scan over the partitions' counts, the
// leftover bookkeeping, the barriers — is the same whether 7 or 14 records per partition arrive (4096 rows x 0.9 / 512
// partitions = 7.2: half a line), so two tiles are evaluated and counted one after the other (the second tile's rows in
// registers of their own) and sorted, lined up and written out together.  LDS: 161 KB of the 160 KiB.
static void kernel_body_compact2(std::ostringstream &o) {
  phase_macros(o);
  const bool direct = scan_opt() & 1u;
  o << (nt_stores_enabled() ? "#define STORE_LINE(p, v) __builtin_nontemporal_store((u64)(v), (p))\n" : "#define STORE_LINE(p, v) (*(p) = (v))\n");
  o << "#define T 4096u\n#define T2 8192u\n#define LR 14u\n#define LEFT 13u\n#define LPL 5u\n"
       "__device__ __forceinline__ u32 lane_up(u32 v, u32 lane, u32 off) { return (u32)__builtin_amdgcn_ds_bpermute((int)((lane - off) << 2), (int)v); }\n"
       // OR over the 8 lanes of a half-line: xor 1, xor 2 (quad permutes), then the mirrored quad (row_half_mirror)
       "__device__ __forceinline__ u32 or8(u32 v) {\n"
       "  v |= (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);\n"
       "  v |= (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);\n"
       "  v |= (u32)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);\n"
       "  return v;\n"
       "}\n"
       // inclusive scan over the wavefront: Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8; lanes without a
       // source keep the 0), then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3
       "__device__ __forceinline__ u32 wave_incl_scan(u32 v) {\n"
       "  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);\n"
       "  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);\n"
       "  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);\n"
       "  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);\n"
       "  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);\n"
       "  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);\n"
       "  return v;\n"
       "}\n"
       "extern \"C\" __global__ void __launch_bounds__(1024) hr_scan_rtc(Args a) {\n"
       "  __shared__ u64 sRec[T2 + NP * LEFT];\n"   // the tile's records sorted by partition, then up to 13 records per partition waiting for a full line
       "  __shared__ u16 sLo[T2 + NP * LEFT];\n"    // their low 9 row bits
       "  __shared__ u32 sCount[2][NP];\n"
       "  __shared__ u32 sStart[NP];\n"            // where the tile's records of a partition go
       "  __shared__ uint2 sLines[(T2 + NP * LEFT) / LR + 2u];\n"   // {partition | line of the tile << 9 | leftovers << 18, first slot | stream cursor << 13}
       "  __shared__ u32 sWave[16];\n"
       "  __shared__ u32 sTotalLines;\n"
       "  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;\n"
       "  for (u32 p = tid; p < NP; p += 1024u) { sCount[0][p] = 0u; sCount[1][p] = 0u; }\n"
       "  u32 myLeftN = 0u, myCursor = 0u;\n"    // thread p < NP keeps partition p's leftover count and stream cursor in registers
       "  __syncthreads();\n"
       "  u64 *myB = reinterpret_cast<u64 *>(a.recB) + (u64)blockIdx.x * NP * a.capB * 16u;\n"  // capB: lines per stream
       "  const u32 numTiles = ((u32)a.length + T - 1u) / T;\n"
       "  const u32 firstTile = blockIdx.x * a.chunkTiles;\n"
       "  const u32 endTile = firstTile + a.chunkTiles < numTiles ? firstTile + a.chunkTiles : numTiles;\n"
       "  u32 tile = firstTile, par = 0u;\n"
       "  Raw R;\n"
       "  PH_DECL\n"
       "  load_tile(R, a, tile * T + tid * 4u);\n"
       // the lane's place in a line: 16 lanes per line, lanes 0 and 8 carry the two headers
       "  const u32 q = tid & 15u, r8 = q & 7u;\n"
       "  const u32 kk = (q >> 3) * 7u + (r8 ? r8 - 1u : 0u);\n"  // record of the line this lane carries
       "  const u32 sh = r8 ? 9u * (r8 - 1u) : 0u;\n"
       "  while (tile < endTile) {\n"
       "    u32 i0 = tile * T + tid * 4u;\n"          // eval4p moves it to the first row the lane's registers hold
       "    u32 hh[4], cv[4], cw[4], alive[4], rank[4];\n"
       "    u32 hhB[4], cvB[4], aliveB[4], rankB[4];\n"
       "    const u32 next = tile + 2u;\n"
       "    eval4p(R, a, i0, hh, cv, cw, alive, (tile + 1u) * T + tid * 4u);\n"
       "    const u32 rc0 = i0 - firstTile * T;\n"    // row within the chunk
       "    u32 *cnt = sCount[par];\n"
       "#pragma unroll\n"
       "    for (int j = 0; j < 4; j++) {\n"
       "      rank[j] = 0u;\n"
       "      if (alive[j]) rank[j] = __hip_atomic_fetch_add(&cnt[hh[j] >> (32 - PB)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "    }\n"
       // the pass's second tile (none: the chunk's last pass holds one tile — its rows are the next workgroup's)
       "    u32 i0B = (tile + 1u) * T + tid * 4u;\n"
       "    const bool second = tile + 1u < endTile;\n"
       "    eval4p(R, a, i0B, hhB, cvB, cw, aliveB, next * T + tid * 4u);\n"
       "    const u32 rc0B = i0B - firstTile * T;\n"
       "#pragma unroll\n"
       "    for (int j = 0; j < 4; j++) {\n"
       "      rankB[j] = 0u;\n"
       "      aliveB[j] = second ? aliveB[j] : 0u;\n"
       "      if (aliveB[j]) rankB[j] = __hip_atomic_fetch_add(&cnt[hhB[j] >> (32 - PB)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "    }\n"
       "    __syncthreads();\n"
       "    PH(0)\n"
       // exclusive scans of (new records, whole lines) per partition, packed in one word
       "    u32 myCount = 0u, myLeft = 0u;\n"
       "    if (tid < NP) { myCount = cnt[tid]; myLeft = myLeftN; }\n"
       "    const u32 myHave = myCount + myLeft, myLines = myHave / LR;\n"
       "    const u32 packed = (myCount << 16) | myLines;\n"
       "    const u32 incl = wave_incl_scan(packed);\n"
       "    if (lane == 63u) sWave[wave] = incl;\n"
       "    __syncthreads();\n"
       "    u32 before = 0u;\n"
       "#pragma unroll\n"
       "    for (u32 w = 0u; w < (NP + 63u) / 64u; w++) { const u32 t = sWave[w]; before += w < wave ? t : 0u; }\n"
       "    const u32 excl = before + incl - packed;\n"
       "    const u32 myStart = excl >> 16, myLineStart = excl & 0xFFFFu;\n"
       "    if (tid < NP) {\n"
    // a partition that completes no line in this tile takes its records straight into its remainder
    << (direct ? "      sStart[tid] = myLines ? myStart : T2 + tid * LEFT + myLeft;\n" : "      sStart[tid] = myStart;\n")
    << "      for (u32 c = 0u; c < myLines; c++) sLines[myLineStart + c] = make_uint2(tid | (c << 9) | (myLeft << 19), myStart | (myCursor << 13));\n"
       "      if (tid == NP - 1u) sTotalLines = myLineStart + myLines;\n"
       "    }\n"
       "    __syncthreads();\n"
       "    PH(1)\n"
       "#pragma unroll\n"
       "    for (int j = 0; j < 4; j++)\n"
       "      if (alive[j]) {\n"
       "        const u32 at = sStart[hh[j] >> (32 - PB)] + rank[j], rc = rc0 + (u32)j;\n"
       "        sRec[at] = ((u64)((hh[j] << PB) | (rc >> 9)) << 32) | cv[j];\n"
       "        sLo[at] = (u16)(rc & 511u);\n"
       "      }\n"
       "#pragma unroll\n"
       "    for (int j = 0; j < 4; j++)\n"
       "      if (aliveB[j]) {\n"
       "        const u32 at = sStart[hhB[j] >> (32 - PB)] + rankB[j], rc = rc0B + (u32)j;\n"
       "        sRec[at] = ((u64)((hhB[j] << PB) | (rc >> 9)) << 32) | cvB[j];\n"
       "        sLo[at] = (u16)(rc & 511u);\n"
       "      }\n"
       "    __syncthreads();\n"
       "    PH(2)\n"
       "    const u32 totalLines = sTotalLines;\n"
       "    for (u32 L0 = tid >> 4; L0 < totalLines; L0 += 64u * LPL) {\n"
       "      u32 e[LPL], lf[LPL], st[LPL], cu[LPL], lo[LPL];\n"
       "      u64 rec[LPL];\n"
       "#pragma unroll\n"
       "      for (u32 j = 0u; j < LPL; j++) {\n"
       "        const u32 L = L0 + j * 64u;\n"
       "        const uint2 w = L < totalLines ? sLines[L] : make_uint2(0u, 0u);\n"
       "        e[j] = w.x & 0x7FFFFu; lf[j] = w.x >> 19; st[j] = w.y & 0x1FFFu; cu[j] = w.y >> 13;\n"
       "      }\n"
       "#pragma unroll\n"
       "      for (u32 j = 0u; j < LPL; j++) {\n"
       "        const u32 p = e[j] & 511u, idx = (e[j] >> 9) * LR + kk;\n"
       "        const u32 at = idx < lf[j] ? T2 + p * LEFT + idx : st[j] + idx - lf[j];\n"
       "        rec[j] = sRec[at];\n"
       "        lo[j] = sLo[at];\n"
       "      }\n"
       "#pragma unroll\n"
       "      for (u32 j = 0u; j < LPL; j++) {\n"
       "        const u64 mine = r8 ? (u64)lo[j] << sh : 0ull;\n"
       "        const u64 hdr = ((u64)or8((u32)(mine >> 32)) << 32) | or8((u32)mine);\n"
       "        const u32 p = e[j] & 511u, line = cu[j] + (e[j] >> 9);\n"
       "        if (L0 + j * 64u < totalLines && line < a.capB) STORE_LINE(&myB[((u64)p * a.capB + line) * 16u + q], r8 ? rec[j] : hdr);\n"
       "      }\n"
       "    }\n"
       "    __syncthreads();\n"
       "    PH(3)\n"
       // what is left of each partition (< 14 records) moves to its LDS remainder; cursors advance
       "    if (tid < NP) {\n"
       "      const u32 rem = myHave - myLines * LR;\n"
    << (direct ?  // only after a line: the 13 slots behind the last line go to the remainder as they are (those past `rem` are never read)
                 "      if (myLines) {\n"
                 "        const u32 from = myStart + myLines * LR - myLeft, to = T2 + tid * LEFT;\n"
                 "        u64 t[LEFT]; u16 tl[LEFT];\n"
                 "#pragma unroll\n"
                 "        for (u32 k = 0u; k < LEFT; k++) { t[k] = sRec[from + k]; tl[k] = sLo[from + k]; }\n"
                 "#pragma unroll\n"
                 "        for (u32 k = 0u; k < LEFT; k++) { sRec[to + k] = t[k]; sLo[to + k] = tl[k]; }\n"
                 "      }\n"
               : "      const u32 n = myLines ? rem : myCount;\n"
                 "      const u32 from = myLines ? myStart + myLines * LR - myLeft : myStart;\n"
                 "      const u32 to = T2 + tid * LEFT + (myLines ? 0u : myLeft);\n"
                 "      u64 t[LEFT]; u16 tl[LEFT];\n"
                 "#pragma unroll\n"
                 "      for (u32 k = 0u; k < LEFT; k++) { const u32 s = from + (k < n ? k : 0u); t[k] = sRec[s]; tl[k] = sLo[s]; }\n"
                 "#pragma unroll\n"
                 "      for (u32 k = 0u; k < LEFT; k++) if (k < n) { sRec[to + k] = t[k]; sLo[to + k] = tl[k]; }\n")
    << "      myLeftN = rem;\n"
       "      u32 cur = myCursor + myLines;\n"
       "      if (cur > a.capB) { *a.overflow = 1u; cur = a.capB; }\n"
       "      myCursor = cur;\n"
       "      cnt[tid] = 0u;\n"  // this counter set is used again two tiles from now
       "    }\n"
       "    par ^= 1u;\n"
       "    tile = next;\n"
       "    PH(4)\n"
       "  }\n"
       "  __syncthreads();\n"
       "  if (tid < NP) sLines[tid] = make_uint2(myLeftN, myCursor);\n"
       "  __syncthreads();\n"
       "  PH(5)\n"
       // the remainders go out as one last, partly filled line each; countsB holds the exact number of records
       "  for (u32 p = tid >> 4; p < NP; p += 64u) {\n"
       "    const uint2 m = sLines[p];\n"
       "    const u32 left = m.x, cur = m.y;\n"
       "    const bool fits = cur < a.capB, has = r8 && kk < left;\n"
       "    const u64 rec = has ? sRec[T2 + p * LEFT + kk] : 0ull;\n"
       "    const u64 mine = has ? (u64)sLo[T2 + p * LEFT + kk] << sh : 0ull;\n"
       "    const u64 hdr = ((u64)or8((u32)(mine >> 32)) << 32) | or8((u32)mine);\n"
       "    if (left && fits) STORE_LINE(&myB[((u64)p * a.capB + cur) * 16u + q], r8 ? rec : hdr);\n"
       "    if (left && !fits) *a.overflow = 1u;\n"
       "    if (q == 0u) a.countsB[(u64)blockIdx.x * NP + p] = cur * LR + ((left && fits) ? left : 0u);\n"
       "  }\n"
       "  PH(6)\n"
       "  PH_OUT\n"
       "}\n";
}

// ---- TABLE scan ----------------------------------------------------------------------------------------
// Low-cardinality queries: every workgroup aggregates its rows in an LDS hash table (key = hash << 32 | lowest
// row, 8-byte value) and emits one 16-byte record per group {row, hash, value} into region A at the end — the
// layout hr::flush_table writes and hr::merge_body reads.  No barrier inside the loop: the wavefronts run free,
// two tiles per wavefront in flight (two register buffers, each refilled column by column while it is evaluated:
// with two or three columns a single tile per wavefront leaves too few bytes in flight to cover HBM latency).
// The table is the specialised merge's: buckets of four keys (two 16-byte LDS reads).  A row first looks at its
// home bucket with straight-line code — it meets its group there nearly always once the groups exist: one LDS
// atomic more —; rows that do not are queued per wavefront in LDS and taken through the general probe loop 64 at a
// time, every lane busy.  Once the table holds LIMIT groups a row whose group finds no slot is written as a single
// record straight away (one global cursor reservation): always correct, slow when frequent — the host sends
// queries with that many groups to the DIRECT kernels.
static void kernel_body_table(std::ostringstream &o) {
  // table: 32-bit keys (the hash; 0xFFFFFFFF = empty — a row whose hash IS that value travels alone), the groups'
  // lowest rows and their values in arrays of their own: a probe is one 16-byte LDS read and four 32-bit compares
  o << "#define T 4096u\n#define SLOTS " << hr::kSlots << "u\n#define BUCKETS (SLOTS / 4u)\n#define LIMIT " << (hr::kSlots * 3 / 4)
    << "u\n#define EMPTY 0xFFFFFFFFu\n#define QCAP 128u\n"
       "struct Probe { u32 b, slot; bool done, spill; };\n"
       "__device__ __forceinline__ void probe_round(u32 *sKeys, u32 *sClaims, Probe &q, u32 h) {\n"
       "  const uint4 k = *reinterpret_cast<const uint4 *>(sKeys + 4u * q.b);\n"
       "  const bool e0 = k.x == EMPTY, e1 = k.y == EMPTY, e2 = k.z == EMPTY, e3 = k.w == EMPTY;\n"
       "  const bool m0 = k.x == h, m1 = k.y == h, m2 = k.z == h, m3 = k.w == h;\n"
       "  const bool anyM = m0 | m1 | m2 | m3, anyE = e0 | e1 | e2 | e3;\n"
       "  const u32 mi = m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u, ei = e0 ? 0u : e1 ? 1u : e2 ? 2u : 3u;\n"
       "  const bool active = !q.done, hit = active & anyM;\n"
       "  q.slot = hit ? 4u * q.b + mi : q.slot;\n"
       "  bool claimed = false;\n"
       "  if (active & !anyM & anyE) {\n"  // a group this workgroup has not seen yet
       "    if (__hip_atomic_load(sClaims, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= LIMIT) {\n"
       "      q.spill = true; claimed = true;\n"  // the table is full enough: the row travels alone
       "    } else {\n"
       "      u32 expected = EMPTY;\n"
       "      if (__hip_atomic_compare_exchange_strong(sKeys + 4u * q.b + ei, &expected, h, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {\n"
       "        __hip_atomic_fetch_add(sClaims, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "        q.slot = 4u * q.b + ei; claimed = true;\n"
       "      }\n"  // lost the slot: the same bucket again next round (the winner may be this very group)
       "    }\n"
       "  }\n"
       "  q.b = (active & !anyM & !anyE) ? (q.b + 1u) & (BUCKETS - 1u) : q.b;\n"
       "  q.done = q.done | hit | claimed;\n"
       "}\n"
       // one row through the general probe loop
       "__device__ __forceinline__ void insert(const Args &a, u32 *sKeys, u32 *sRows, u64 *sVals, u32 *sClaims, u32 row, u32 h, u32 carried) {\n"
       "  const u64 value = widen(carried);\n"
       "  Probe q; q.b = h & (BUCKETS - 1u); q.slot = 0u; q.done = false; q.spill = h == EMPTY;\n"
       "  for (u32 tries = 0u; tries < BUCKETS + 8u && !q.done && !q.spill; tries++) probe_round(sKeys, sClaims, q, h);\n"
       "  if (q.done && !q.spill) {\n"
       "    __hip_atomic_fetch_min(sRows + q.slot, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "    agg(sVals + q.slot, value);\n"
       "  } else {\n"
       "    const u32 p = PB ? h >> (32 - (PB ? PB : 1)) : 0u;\n"
       "    const u64 at = atomicAdd(a.cursorsA + p, 1u);\n"
       "    if (at < a.capA) a.recA[(u64)p * a.capA + at] = make_uint4(row, h, (u32)value, (u32)(value >> 32));\n"
       "    else *a.overflow = 1u;\n"
       "  }\n"
       "}\n"
       "__device__ __forceinline__ void drain(const Args &a, u32 *queue, u32 first, u32 count, u32 lane, u32 *sKeys, u32 *sRows, u64 *sVals, u32 *sClaims) {\n"
       "  asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n"
       "  if (lane < count) {\n"
       "    const u32 e = 3u * (first + lane);\n"
       "    insert(a, sKeys, sRows, sVals, sClaims, queue[e], queue[e + 1u], queue[e + 2u]);\n"
       "  }\n"
       "  asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n"
       "}\n"
       // one row, round one: the home bucket, straight-line
       "__device__ __forceinline__ void row_one(const Args &a, bool valid, u32 row, u32 h, u32 carried, u32 lane, u32 *sKeys, u32 *sRows, u64 *sVals, u32 *sClaims, u32 *queue, u32 &qn) {\n"
       "  const u32 b = h & (BUCKETS - 1u);\n"
       "  const uint4 k = *reinterpret_cast<const uint4 *>(sKeys + 4u * b);\n"
       "  const bool m0 = k.x == h, m1 = k.y == h, m2 = k.z == h, m3 = k.w == h;\n"
       "  const bool hit = valid && h != EMPTY && (m0 || m1 || m2 || m3);\n"
       "  if (hit) {\n"
       "    const u32 slot = 4u * b + (m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u);\n"
       "    __hip_atomic_fetch_min(sRows + slot, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "    agg(sVals + slot, widen(carried));\n"
       "  }\n"
       "  const bool pend = valid && !hit;\n"
       "  const u64 m = __ballot(pend);\n"
       "  if (m) {\n"
       "    if (pend) {\n"
       "      const u32 e = 3u * (qn + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)));\n"
       "      queue[e] = row; queue[e + 1u] = h; queue[e + 2u] = carried;\n"
       "    }\n"
       "    qn += (u32)__popcll(m);\n"
       "    if (qn >= 64u) { qn -= 64u; drain(a, queue, qn, 64u, lane, sKeys, sRows, sVals, sClaims); }\n"
       "  }\n"
       "}\n"
       "extern \"C\" __global__ void __launch_bounds__(1024) hr_scan_rtc(Args a) {\n"
       "  __shared__ __attribute__((aligned(16))) u32 sKeys[SLOTS];\n"
       "  __shared__ u32 sRows[SLOTS];\n"
       "  __shared__ u64 sVals[SLOTS];\n"
       "  __shared__ u32 sQueue[16u * QCAP * 3u];\n"
       "  __shared__ u32 sPartCount[NP], sPartBase[NP];\n"
       "  __shared__ u32 sClaims;\n"
       "  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;\n"
       "  for (u32 s = tid; s < SLOTS; s += 1024u) { sKeys[s] = EMPTY; sRows[s] = 0xFFFFFFFFu; sVals[s] = IDENT; }\n"
       "  for (u32 p = tid; p < NP; p += 1024u) sPartCount[p] = 0u;\n"
       "  if (tid == 0u) sClaims = 0u;\n"
       "  __syncthreads();\n"
       "  const u32 numTiles = ((u32)a.length + T - 1u) / T, G = gridDim.x;\n"
       "  u32 tile = blockIdx.x, qn = 0u;\n"
       "  u32 *queue = sQueue + wave * (QCAP * 3u);\n"
       "  Raw R0, R1;\n"
       "  load_tile(R0, a, tile * T + tid * 4u);\n"
       "  load_tile(R1, a, (tile + G) * T + tid * 4u);\n"
       "#define TILE_STEP(R)                                                                                   \\\n"
       "  {                                                                                                    \\\n"
       "    u32 i0 = tile * T + tid * 4u;                                                                      \\\n"
       "    const u32 next = tile + 2u * G;                                                                    \\\n"
       "    u32 hh[4], cv[4], cw[4], alive[4];                                                                 \\\n"
       "    eval4p(R, a, i0, hh, cv, cw, alive, next * T + tid * 4u);                                          \\\n"
       "    const u32 row0 = a.rowBase + i0;                                                                   \\\n"
       "    row_one(a, alive[0] != 0u, row0, hh[0], cv[0], lane, sKeys, sRows, sVals, &sClaims, queue, qn);    \\\n"
       "    row_one(a, alive[1] != 0u, row0 + 1u, hh[1], cv[1], lane, sKeys, sRows, sVals, &sClaims, queue, qn); \\\n"
       "    row_one(a, alive[2] != 0u, row0 + 2u, hh[2], cv[2], lane, sKeys, sRows, sVals, &sClaims, queue, qn); \\\n"
       "    row_one(a, alive[3] != 0u, row0 + 3u, hh[3], cv[3], lane, sKeys, sRows, sVals, &sClaims, queue, qn); \\\n"
       "    tile += G;                                                                                         \\\n"
       "  }\n"
       "  while (tile < numTiles) {\n"
       "    TILE_STEP(R0)\n"
       "    if (tile >= numTiles) break;\n"
       "    TILE_STEP(R1)\n"
       "  }\n"
       "  if (qn) drain(a, queue, 0u, qn, lane, sKeys, sRows, sVals, &sClaims);\n"
       "  __syncthreads();\n"
       // flush (hr::flush_table): counting sort of the entries by partition, one cursor reservation per partition
       "  u32 rank[SLOTS / 1024u];\n"
       "#pragma unroll\n"
       "  for (u32 k = 0u; k < SLOTS / 1024u; k++) {\n"
       "    const u32 key = sKeys[tid + k * 1024u];\n"
       "    rank[k] = 0u;\n"
       "    if (key != EMPTY) rank[k] = __hip_atomic_fetch_add(&sPartCount[PB ? key >> (32 - (PB ? PB : 1)) : 0u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "  }\n"
       "  __syncthreads();\n"
       "  for (u32 p = tid; p < NP; p += 1024u) { const u32 c = sPartCount[p]; if (c) sPartBase[p] = atomicAdd(a.cursorsA + p, c); }\n"
       "  __syncthreads();\n"
       "#pragma unroll\n"
       "  for (u32 k = 0u; k < SLOTS / 1024u; k++) {\n"
       "    const u32 s = tid + k * 1024u;\n"
       "    const u32 key = sKeys[s];\n"
       "    if (key == EMPTY) continue;\n"
       "    const u32 p = PB ? key >> (32 - (PB ? PB : 1)) : 0u;\n"
       "    const u64 at = (u64)sPartBase[p] + rank[k], v = sVals[s];\n"
       "    if (at < a.capA) a.recA[(u64)p * a.capA + at] = make_uint4(sRows[s], key, (u32)v, (u32)(v >> 32));\n"
       "    else *a.overflow = 1u;\n"
       "  }\n"
       "}\n";
}

// SCAN_SORT64: the Sort + Reduce path (sort_reduce_fused.hip) — 16-byte line records {row, hash64 >> 32, carried measure, (u32)hash64}
// keyed by murmur3_x64_128 of the packed row (what Sort hashes: query/sort_reduce.cu:118-133), partitioned by the TOP bits of the
// 64-bit hash so that a partition is a contiguous range of the sorted order; plan.measure.col < 0: a constant measure
// (COUNT(*) is SUM over the literal 1) — no measure column is read, records carry Args::k's measure slot
enum ScanKind { SCAN_LINES16 = 0, SCAN_COMPACT = 1, SCAN_TABLE = 2, SCAN_SORT64 = 3 };

// ---- dimension slots of 1, 2 or 4 bytes -------------------------------------------------------------------
// The dimension vector holds, for each dimension in descending width order, capacity x width value bytes, then one
// validity byte vector per dimension (dim_layout.hpp); the row that is hashed is [values][validity bytes], every field
// naturally aligned.  With the widths known when the source is written, the packed row becomes a list of 32-bit words,
// each the OR of the fields that fall into it.
struct SlotLayout {
  int nd = 0, valueBytes = 0;
  int width[kFusedDims] = {4, 4, 4, 4, 4, 4, 4, 4}, off[kFusedDims] = {};
  bool all4 = true, ok = true;
};
SlotLayout slot_layout(const FusedPlanD &plan, int nd) {
  SlotLayout L;
  L.nd = nd;
  int prev = 4;
  for (int d = 0; d < nd && d < kFusedDims; d++) {
    const int w = fused_dim_width(plan, d);
    L.ok = L.ok && (w == 4 || w == 2 || w == 1) && w <= prev;  // descending: fields never straddle a word
    prev = w;
    L.width[d] = w;
    L.off[d] = L.valueBytes;
    L.valueBytes += w;
    L.all4 = L.all4 && w == 4;
  }
  // (the all-4-byte shortcuts pack the validity bytes of up to four dimensions into one word: beyond, the general word list)
  L.all4 = L.all4 && nd <= 4;
  return L;
}
// murmur3_x86_32 (seed 0) of the packed row: `val(d)` names the dimension's value (already truncated to its width),
// `okb(d)` its validity (0 / 1); writes the statements that leave the hash in `out`
void gen_row_hash(std::ostringstream &o, const SlotLayout &L, const std::function<std::string(int)> &val,
                  const std::function<std::string(int)> &okb, const std::string &out, const char *indent) {
  const int total = L.valueBytes + L.nd, words = (total + 3) / 4;
  std::vector<std::string> w(static_cast<size_t>(words));
  auto add = [&](int byteOff, const std::string &e) {
    std::string &x = w[static_cast<size_t>(byteOff / 4)];
    const int sh = 8 * (byteOff % 4);
    const std::string term = sh ? "(" + e + " << " + std::to_string(sh) + ")" : e;
    x = x.empty() ? term : x + " | " + term;
  };
  for (int d = 0; d < L.nd; d++) add(L.off[d], val(d));
  for (int d = 0; d < L.nd; d++) add(L.valueBytes + d, okb(d));
  o << indent << "{\n" << indent << "  u32 g = 0u;\n";
  for (int k = 0; k < total / 4; k++) o << indent << "  g = mix(g, " << w[static_cast<size_t>(k)] << ");\n";
  if (total % 4) o << indent << "  { u32 k = (" << w[static_cast<size_t>(words - 1)] << ") * 0xcc9e2d51u; k = rotl(k, 15) * 0x1b873593u; g ^= k; }\n";
  o << indent << "  g ^= " << total << "u; g ^= g >> 16; g *= 0x85ebca6bu; g ^= g >> 13; g *= 0xc2b2ae35u; g ^= g >> 16;\n"
    << indent << "  " << out << " = g;\n" << indent << "}\n";
}
// lo64(murmur3_x64_128) (seed 0) of the packed row — Murmur128Stream of dim_layout.hpp, query/utils.cu:157-241 — with the
// same naming of values and validity bits as gen_row_hash; writes the statements that leave the hash in the u64 `out`
void gen_row_hash64(std::ostringstream &o, const SlotLayout &L, const std::function<std::string(int)> &val,
                    const std::function<std::string(int)> &okb, const std::string &out, const char *indent) {
  const int total = L.valueBytes + L.nd, words = (total + 3) / 4;
  std::vector<std::string> w(static_cast<size_t>(words));
  auto add = [&](int byteOff, const std::string &e) {
    std::string &x = w[static_cast<size_t>(byteOff / 4)];
    const int sh = 8 * (byteOff % 4);
    const std::string term = sh ? "(" + e + " << " + std::to_string(sh) + ")" : e;
    x = x.empty() ? term : x + " | " + term;
  };
  for (int d = 0; d < L.nd; d++) add(L.off[d], val(d));
  for (int d = 0; d < L.nd; d++) add(L.valueBytes + d, okb(d));
  auto lane64 = [&](int firstWord) {  // 8 row bytes from 32-bit word `firstWord` on, as a u64 expression ("" = none left)
    if (firstWord >= words) return std::string();
    std::string e = "(u64)(" + w[static_cast<size_t>(firstWord)] + ")";
    if (firstWord + 1 < words) e += " | ((u64)(" + w[static_cast<size_t>(firstWord + 1)] + ") << 32)";
    return e;
  };
  const std::string in = indent;
  o << in << "{\n" << in << "  u64 g1 = 0ull, g2 = 0ull, q1, q2;\n";
  const int blocks = total / 16;
  for (int b = 0; b < blocks; b++) {
    o << in << "  q1 = " << lane64(4 * b) << "; q2 = " << lane64(4 * b + 2) << ";\n"
      << in << "  q1 *= MC1; q1 = rotl64(q1, 31); q1 *= MC2; g1 ^= q1; g1 = rotl64(g1, 27); g1 += g2; g1 = g1 * 5ull + 0x52dce729ull;\n"
      << in << "  q2 *= MC2; q2 = rotl64(q2, 33); q2 *= MC1; g2 ^= q2; g2 = rotl64(g2, 31); g2 += g1; g2 = g2 * 5ull + 0x38495ab5ull;\n";
  }
  const int tail = total % 16;
  if (tail > 8) o << in << "  q2 = " << lane64(4 * blocks + 2) << "; q2 *= MC2; q2 = rotl64(q2, 33); q2 *= MC1; g2 ^= q2;\n";
  if (tail > 0) o << in << "  q1 = " << lane64(4 * blocks) << "; q1 *= MC1; q1 = rotl64(q1, 31); q1 *= MC2; g1 ^= q1;\n";
  o << in << "  g1 ^= " << total << "ull; g2 ^= " << total << "ull; g1 += g2; g2 += g1;\n"
    << in << "  g1 = fmix64(g1); g2 = fmix64(g2); g1 += g2;\n"
    << in << "  " << out << " = g1;\n" << in << "}\n";
}
const char *kPrelude64 =
    "#define MC1 0x87c37b91114253d5ull\n#define MC2 0x4cf5ad432745937full\n"
    "__device__ __forceinline__ u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }\n"
    "__device__ __forceinline__ u64 fmix64(u64 k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k; }\n";
// mask that truncates a 32-bit value to a slot's width
const char *width_mask(int w) { return w == 4 ? "" : w == 2 ? " & 0xFFFFu" : " & 0xFFu"; }
// dimension d of row `row` of a dimension vector at `base` (capacity `cap`), zero-extended; and its validity byte
std::string slot_load(const SlotLayout &L, int d, const char *base, const char *cap, const std::string &row) {
  const std::string at = std::string(base) + " + (u64)" + std::to_string(L.off[d]) + " * " + cap + " + " + std::to_string(L.width[d]) + "ull * " + row;
  return L.width[d] == 4 ? "*reinterpret_cast<const u32 *>(" + at + ")" : L.width[d] == 2 ? "(u32)*reinterpret_cast<const u16 *>(" + at + ")" : "(u32)*(" + at + ")";
}
std::string slot_store(const SlotLayout &L, int d, const char *base, const char *cap, const std::string &row, const std::string &v) {
  const std::string at = std::string(base) + " + (u64)" + std::to_string(L.off[d]) + " * " + cap + " + " + std::to_string(L.width[d]) + "ull * " + row;
  return L.width[d] == 4 ? "*reinterpret_cast<u32 *>(" + at + ") = " + v + ";" : L.width[d] == 2 ? "*reinterpret_cast<u16 *>(" + at + ") = (u16)(" + v + ");" : "*(" + at + ") = (u8)(" + v + ");";
}
// element `idx` of a source column of `step` bytes per value, widened to 32 bits (sign-extended for int kinds)
std::string column_elem(int step, bool sgn, const std::string &base, const std::string &idx) {
  if (step == 4) return base + "[" + idx + "]";
  if (step == 2) return sgn ? "(u32)(i32)reinterpret_cast<const short *>(" + base + ")[" + idx + "]" : "(u32)reinterpret_cast<const u16 *>(" + base + ")[" + idx + "]";
  return sgn ? "(u32)(i32)reinterpret_cast<const signed char *>(" + base + ")[" + idx + "]" : "(u32)reinterpret_cast<const u8 *>(" + base + ")[" + idx + "]";
}
// which column slot's stored kind is signed (decides the widening of a narrow column): the expressions that read it say
bool column_signed(const FusedPlanD &plan, int nd, int c) {
  for (int d = 0; d < nd; d++)
    if (plan.dims[d].col == c) return plan.dims[d].f.akind == K_I32;
  if (plan.measure.col == c) return plan.measure.f.akind == K_I32;
  for (int k = 0; k < plan.numFilters && k < kFusedFilters; k++)
    if (plan.filters[k].col == c) return plan.filters[k].f.akind == K_I32;
  return false;
}

// the whole kernel source for `plan`; empty when the plan is outside the supported shapes.  `agg` / `widen`
// are read for SCAN_TABLE only.
std::string generate(const FusedPlanD &plan, int nd, int partBits, uint32_t nullMask, ScanKind kind, const AggSpec *agg = nullptr,
                     const hr::Widen *widen = nullptr) {
  if (nd < 1 || nd > kFusedDims || plan.numCols > kFusedCols) return "";
  if (kind == SCAN_COMPACT && partBits < 3) return "";
  const SlotLayout SL = slot_layout(plan, nd);
  if (!SL.ok) return "";
  for (int c = 0; c < plan.numCols; c++) {
    const int st = fused_col_step(plan, c);
    if (!(st == 4 || st == 2 || st == 1)) return "";
  }
  std::ostringstream o;
  const int nc = plan.numCols;
  const bool sort64 = kind == SCAN_SORT64;
  const bool constMeasure = sort64 && plan.measure.col < 0;  // the records carry Args::k's measure slot as it is
  if (plan.measure.col < 0 && !sort64) return "";
  const int firstFilterCol = constMeasure ? nd : nd + 1;  // column slots: dimension d -> d, measure -> nd (if any), then the filters' own
  o << times5_text() << kPrelude << (sort64 ? kPrelude64 : "") << args_text()
    << "#define NC " << nc << "\n#define ND " << nd << "\n#define PB " << partBits << "\n#define NP " << (1 << partBits) << "\n"
       "struct Raw { u32 v[NC][4]; u32 win[NC]; };\n";
  // ---- loads, one column at a time.  Always the full 16 bytes + the 16-bit validity window, from a row index clamped
  // to length - 4: no guarded variant, hence no branch at load time (a branch around a load makes the compiler copy
  // the loaded registers at the join — and wait for the load right there).  The one quad of the shard that straddles
  // its end is shifted into place when its tile is evaluated; lanes past the end hold rows that do not take part.
  o << "__device__ __forceinline__ u32 clampi(const Args &a, u32 i0) { const u32 lim = a.length >= 4 ? (u32)a.length - 4u : 0u; return i0 < lim ? i0 : lim; }\n";
  for (int c = 0; c < nc; c++) {
    o << "__device__ __forceinline__ void load_col" << c << "(Raw &r, const Args &a, u32 i0c) {\n";
    const int step = fused_col_step(plan, c);
    if (step != 4) {  // a quad of a 2- / 1-byte column is 8 / 4 bytes: one load, widened in registers (query/iterator.hpp:146-165)
      const bool sgn = column_signed(plan, nd, c);
      const char *ptr = step == 2 ? "reinterpret_cast<const u16 *>(a.vals[" : "reinterpret_cast<const u8 *>(a.vals[";
      if (step == 2) {
        if (nt_enabled()) o << "  const U2 t = __builtin_nontemporal_load(reinterpret_cast<const U2a *>(" << ptr << c << "]) + i0c));\n  const u32 t0 = t.x, t1 = t.y;\n";
        else o << "  const PU32x2 t = *reinterpret_cast<const PU32x2 *>(" << ptr << c << "]) + i0c); const u32 t0 = t.v[0], t1 = t.v[1];\n";
        if (sgn) o << "  r.v[" << c << "][0] = (u32)((i32)(t0 << 16) >> 16); r.v[" << c << "][1] = (u32)((i32)t0 >> 16); r.v[" << c << "][2] = (u32)((i32)(t1 << 16) >> 16); r.v[" << c << "][3] = (u32)((i32)t1 >> 16);\n";
        else o << "  r.v[" << c << "][0] = t0 & 0xFFFFu; r.v[" << c << "][1] = t0 >> 16; r.v[" << c << "][2] = t1 & 0xFFFFu; r.v[" << c << "][3] = t1 >> 16;\n";
      } else {
        if (nt_enabled()) o << "  const u32 t0 = __builtin_nontemporal_load(reinterpret_cast<const U1a *>(" << ptr << c << "]) + i0c));\n";
        else o << "  const u32 t0 = reinterpret_cast<const PU32 *>(" << ptr << c << "]) + i0c)->v;\n";
        if (sgn) o << "  r.v[" << c << "][0] = (u32)((i32)(t0 << 24) >> 24); r.v[" << c << "][1] = (u32)((i32)(t0 << 16) >> 24); r.v[" << c << "][2] = (u32)((i32)(t0 << 8) >> 24); r.v[" << c << "][3] = (u32)((i32)t0 >> 24);\n";
        else o << "  r.v[" << c << "][0] = t0 & 0xFFu; r.v[" << c << "][1] = (t0 >> 8) & 0xFFu; r.v[" << c << "][2] = (t0 >> 16) & 0xFFu; r.v[" << c << "][3] = t0 >> 24;\n";
      }
    } else
    if (nt_enabled())
      o << "  const U4 t = __builtin_nontemporal_load(reinterpret_cast<const U4a *>(a.vals[" << c << "] + i0c)); r.v[" << c << "][0] = t.x; r.v[" << c
        << "][1] = t.y; r.v[" << c << "][2] = t.z; r.v[" << c << "][3] = t.w;\n";
    else
      o << "  const PU32x4 t = *reinterpret_cast<const PU32x4 *>(a.vals[" << c << "] + i0c); r.v[" << c << "][0] = t.v[0]; r.v[" << c
        << "][1] = t.v[1]; r.v[" << c << "][2] = t.v[2]; r.v[" << c << "][3] = t.v[3];\n";
    if (nullMask & (1u << c))
      o << "  r.win[" << c << "] = reinterpret_cast<const PU16 *>(a.nulls[" << c << "] + ((i0c + a.bitOff[" << c << "]) >> 3))->v;\n";
    else
      o << "  r.win[" << c << "] = 0xFFFFu;\n";
    o << "}\n";
  }
  o << "__device__ __forceinline__ void load_tile(Raw &r, const Args &a, u32 i0) {\n  const u32 i0c = clampi(a, i0);\n";
  for (int c = 0; c < nc; c++) o << "  load_col" << c << "(r, a, i0c);\n";
  o << "}\n";
  // ---- evaluate + hash one quad (hash, carried measure bits and "takes part" of its four rows) AND issue
  // the next tile's loads, column by column, as soon as a column's registers are dead: a single register buffer, yet
  // loads are in flight during the whole evaluation instead of only after it (the kernel is HBM-bound: with the loads
  // issued after the evaluation the read pipe idled for a third of every tile).  Scheduling barriers pin each load
  // behind the last use of the registers it refills.  i0n: this lane's first row in the next tile (past the last tile the
  // clamp turns the prefetch into one cache line per column: no branch needed); partial: this tile holds the shard's
  // end (wave-uniform).
  {
    const std::string bar = "  __builtin_amdgcn_sched_barrier(0);\n";
    auto prefetch = [&](int c) { o << bar << "  load_col" << c << "(r, a, i0nc);\n" << bar; };
    o << "__device__ __forceinline__ void eval4p(Raw &r, const Args &a, u32 &i0, u32 (&hh)[4], u32 (&cv)[4], u32 (&cw)[4], u32 (&alive)[4], u32 i0n) {\n"
         "  u32 okc[NC];\n"
         "  cw[0] = cw[1] = cw[2] = cw[3] = 0u;\n"
         "  const u32 i0c = clampi(a, i0), i0nc = clampi(a, i0n);\n"
         // the registers hold rows i0c .. i0c + 3: i0's own rows except in the one quad that straddles the shard's end (loaded
         // `sh` rows early: its first `sh` rows belong to the lane before) and in the lanes past the end (sh = 4: no row).
         // The caller numbers the lane's rows from i0c: no register is moved
         "  const u32 sh = i0 - i0c < 4u ? i0 - i0c : 4u;\n"
         "  i0 = i0c;\n";
    for (int c = 0; c < nc; c++) {
      if (nullMask & (1u << c)) o << "  okc[" << c << "] = (r.win[" << c << "] >> ((i0c + a.bitOff[" << c << "]) & 7u)) & 0xFu;\n";
      else o << "  okc[" << c << "] = 0xFu;\n";
    }
    o << "#pragma unroll\n"
         "  for (int j = 0; j < 4; j++) {\n"
         "    u32 keep = ((u32)j >= sh && (int)(i0c + j) < a.length) ? 1u : 0u;\n";
    for (int k = 0; k < plan.numFilters; k++) {
      const FusedExpr &e = plan.filters[k];
      if (e.col < 0 || e.col >= nc) return "";
      o << "    {\n      const u32 v = r.v[" << e.col << "][j]; const u32 okb = (okc[" << e.col << "] >> j) & 1u;\n";
      if (!gen_compare(e.f, o, "v", "okb", "keep", const_name(const_slot_filter(k)))) return "";
      o << "    }\n";
    }
    o << "    alive[j] = keep;\n  }\n";
    for (int c = firstFilterCol; c < nc; c++) prefetch(c);  // columns only the filters read
    if (constMeasure) {
      o << "  cv[0] = cv[1] = cv[2] = cv[3] = " << const_name(const_slot_measure()) << ";\n";
    } else {  // measure: fused_carry
      const FusedExpr &e = plan.measure;
      if (e.col != nd) return "";
      o << "#pragma unroll\n  for (int j = 0; j < 4; j++) {\n"
           "    const u32 v = r.v[" << nd << "][j]; const u32 okb = (okc[" << nd << "] >> j) & 1u; u32 x;\n";
      if (!gen_value(e.f, o, "v", "okb", "x", const_name(const_slot_measure()))) return "";
      if (plan.measureWidth == 8) {
        if (plan.identity != 0) return "";
        o << "    cv[j] = okb ? x : 0u;\n";
      } else {
        const int target = plan.measureDtype == Int32 ? K_I32 : plan.measureDtype == Uint32 ? K_U32 : K_F32;
        if (!plain_store(e.f.rk, target)) return "";
        o << "    cv[j] = okb ? x : " << hex(static_cast<uint32_t>(plan.identity)) << ";\n";
      }
      o << "  }\n";
      prefetch(nd);
    }
    if (SL.all4 && !sort64) {
    o << "  u32 h[4] = {0u, 0u, 0u, 0u}, okbytes[4] = {0u, 0u, 0u, 0u};\n";
    for (int d = 0; d < nd; d++) {
      const FusedExpr &e = plan.dims[d];
      if (e.col != d || !plain_store(e.f.rk, e.outKind)) return "";
      o << "#pragma unroll\n  for (int j = 0; j < 4; j++) {\n"
           "    const u32 v = r.v[" << d << "][j]; const u32 okb = (okc[" << d << "] >> j) & 1u; u32 x;\n";
      if (!gen_value(e.f, o, "v", "okb", "x", const_name(const_slot_dim(d)))) return "";
      o << "    h[j] = mix(h[j], x); okbytes[j] |= okb << " << 8 * d << ";\n  }\n";
      prefetch(d);
    }
    // Murmur32Stream (dim_layout.hpp): the validity bytes are one more block when there are four of them,
    // otherwise the tail
    o << "#pragma unroll\n  for (int j = 0; j < 4; j++) {\n    u32 g = h[j];\n";
    if (nd == 4) o << "    g = mix(g, okbytes[j]);\n";
    else o << "    { u32 k = okbytes[j] * 0xcc9e2d51u; k = rotl(k, 15) * 0x1b873593u; g ^= k; }\n";
    o << "    g ^= " << 5 * nd << "u; g ^= g >> 16; g *= 0x85ebca6bu; g ^= g >> 13; g *= 0xc2b2ae35u; g ^= g >> 16;\n"
         "    hh[j] = g;\n  }\n}\n";
    } else {
      // narrow slots: the values are kept (truncated to their slot, as the dimension vector would hold them) until all
      // are known, then the packed row's words are put together
      for (int d = 0; d < nd; d++) o << "  u32 xv" << d << "[4], xo" << d << "[4];\n";
      for (int d = 0; d < nd; d++) {
        const FusedExpr &e = plan.dims[d];
        if (e.col != d || !plain_store(e.f.rk, e.outKind) || (SL.width[d] != 4 && !int_kind(e.f.rk))) return "";
        o << "#pragma unroll\n  for (int j = 0; j < 4; j++) {\n"
             "    const u32 v = r.v[" << d << "][j]; const u32 okb = (okc[" << d << "] >> j) & 1u; u32 x;\n";
        if (!gen_value(e.f, o, "v", "okb", "x", const_name(const_slot_dim(d)))) return "";
        o << "    xv" << d << "[j] = x" << width_mask(SL.width[d]) << "; xo" << d << "[j] = okb;\n  }\n";
        prefetch(d);
      }
      o << "#pragma unroll\n  for (int j = 0; j < 4; j++) {\n";
      if (sort64) {  // the record's second word holds the hash's upper half (its top bits choose the partition), the fourth the lower
        o << "    u64 h64;\n";
        gen_row_hash64(o, SL, [](int d) { return "xv" + std::to_string(d) + "[j]"; }, [](int d) { return "xo" + std::to_string(d) + "[j]"; },
                       "h64", "    ");
        o << "    hh[j] = (u32)(h64 >> 32); cw[j] = (u32)h64;\n";
      } else {
        gen_row_hash(o, SL, [](int d) { return "xv" + std::to_string(d) + "[j]"; }, [](int d) { return "xo" + std::to_string(d) + "[j]"; },
                     "hh[j]", "    ");
      }
      o << "  }\n}\n";
    }
  }
  if (kind == SCAN_TABLE) {
    if (!agg || !widen || !gen_widen(o, *widen) || !gen_agg(o, *agg)) return "";
    kernel_body_table(o);
  } else if (kind == SCAN_COMPACT) {
    // (two tiles per pass: the second tile's rows need 16 more registers — four dimensions at most)
    if (scan_tiles() == 2 && nd <= 4) kernel_body_compact2(o);
    else kernel_body_compact(o);
  } else {
    kernel_body_lines16(o, sort64 ? "cw[j]" : "0u", sort64 ? "sr_scan_rtc" : "hr_scan_rtc");
  }
  return o.str();
}

// The same kernel over rows [rowBase, rowBase + length) of a dimension vector of `nd` 4-byte dimensions
// (values per dimension, then one validity byte per row and dimension) and a measure vector of `vw`-byte
// values — what HashReduce is handed when the batch's transforms were launched (ARES_FUSE=0, plans the
// fused scan does not cover).  Args: vals[d] / nulls[d] = dimension d's values / validity bytes at
// rowBase, vals[nd] = the measures at rowBase.  Records carry the whole value: {row, hash, lo, hi}.
// sort64: the Sort + Reduce path over materialised vectors (sort_reduce_fused.hip): records {row, hash64 >> 32, the 4-byte value,
// (u32)hash64} keyed by lo64(murmur3_x64_128) of the packed row, partition = top bits of the 64-bit hash; up to eight dimensions.
// widths (sort64 only): the dimension slots' bytes in vector order (4 / 2 / 1, descending); null: all four bytes
std::string generate_vector(int nd, int vw, int partBits, bool sort64 = false, const int *widths = nullptr) {
  if (nd < 1 || nd > (sort64 ? kFusedDims : kGenericFusedDims) || (vw != 4 && vw != 8) || (sort64 && vw != 4)) return "";
  int width[kFusedDims] = {4, 4, 4, 4, 4, 4, 4, 4};
  bool narrow = false;
  for (int d = 0; widths && d < nd; d++) {
    if ((widths[d] != 4 && widths[d] != 2 && widths[d] != 1) || (d && widths[d] > widths[d - 1]) || !sort64) return "";
    width[d] = widths[d];
    narrow = narrow || widths[d] != 4;
  }
  std::ostringstream o;
  const int mq = vw / 4;
  o << times5_text() << kPrelude << (sort64 ? kPrelude64 : "") << args_text()
    << "#define ND " << nd << "\n#define MQ " << mq << "\n#define PB " << partBits << "\n#define NP " << (1 << partBits) << "\n"
       "struct Raw { u32 v[ND][4]; u32 ok[ND]; u32 m[MQ * 4]; };\n"
       "__device__ __forceinline__ void load_full(Raw &r, const Args &a, u32 i0) {\n";
  if (narrow) {  // (slot by slot: a 2-byte slot's four rows are one 8-byte load, a 1-byte slot's one 4-byte load)
    for (int d = 0; d < nd; d++) {
      if (width[d] == 4)
        o << "  { const PU32x4 t = *reinterpret_cast<const PU32x4 *>(a.vals[" << d << "] + i0); r.v[" << d << "][0] = t.v[0]; r.v[" << d
          << "][1] = t.v[1]; r.v[" << d << "][2] = t.v[2]; r.v[" << d << "][3] = t.v[3]; }\n";
      else if (width[d] == 2)
        o << "  { const PU32x2 t = *reinterpret_cast<const PU32x2 *>(reinterpret_cast<const u8 *>(a.vals[" << d << "]) + 2ull * i0); r.v[" << d
          << "][0] = t.v[0] & 0xFFFFu; r.v[" << d << "][1] = t.v[0] >> 16; r.v[" << d << "][2] = t.v[1] & 0xFFFFu; r.v[" << d << "][3] = t.v[1] >> 16; }\n";
      else
        o << "  { const u32 t = reinterpret_cast<const PU32 *>(reinterpret_cast<const u8 *>(a.vals[" << d << "]) + i0)->v; r.v[" << d
          << "][0] = t & 0xFFu; r.v[" << d << "][1] = (t >> 8) & 0xFFu; r.v[" << d << "][2] = (t >> 16) & 0xFFu; r.v[" << d << "][3] = t >> 24; }\n";
      o << "  r.ok[" << d << "] = reinterpret_cast<const PU32 *>(a.nulls[" << d << "] + i0)->v;\n";
    }
  } else {
  o << "#pragma unroll\n"
       "  for (int d = 0; d < ND; d++) {\n"
       "    const PU32x4 t = *reinterpret_cast<const PU32x4 *>(a.vals[d] + i0);\n"
       "    r.v[d][0] = t.v[0]; r.v[d][1] = t.v[1]; r.v[d][2] = t.v[2]; r.v[d][3] = t.v[3];\n"
       "    r.ok[d] = reinterpret_cast<const PU32 *>(a.nulls[d] + i0)->v;\n"
       "  }\n";
  }
  o << ""
       "#pragma unroll\n"
       "  for (int q = 0; q < MQ; q++) {\n"
       "    const PU32x4 t = *reinterpret_cast<const PU32x4 *>(a.vals[ND] + (u64)i0 * MQ + 4 * q);\n"
       "    r.m[4 * q] = t.v[0]; r.m[4 * q + 1] = t.v[1]; r.m[4 * q + 2] = t.v[2]; r.m[4 * q + 3] = t.v[3];\n"
       "  }\n"
       "}\n"
       "__device__ __forceinline__ void load_tail(Raw &r, const Args &a, u32 i0) {\n";
  if (narrow) {
    for (int d = 0; d < nd; d++) {
      const char *elem = width[d] == 4 ? "a.vals[%d][i0 + j]" : width[d] == 2 ? "(u32)reinterpret_cast<const u16 *>(a.vals[%d])[i0 + j]" : "(u32)reinterpret_cast<const u8 *>(a.vals[%d])[i0 + j]";
      char buf[128];
      snprintf(buf, sizeof(buf), elem, d);
      o << "  r.ok[" << d << "] = 0u;\n"
           "  for (int j = 0; j < 4; j++) {\n"
           "    const bool in = (int)(i0 + j) < a.length;\n"
           "    r.v[" << d << "][j] = in ? " << buf << " : 0u;\n"
           "    r.ok[" << d << "] |= in ? (u32)a.nulls[" << d << "][i0 + j] << (8 * j) : 0u;\n"
           "  }\n";
    }
  } else {
  o << "#pragma unroll\n"
       "  for (int d = 0; d < ND; d++) {\n"
       "    r.ok[d] = 0u;\n"
       "    for (int j = 0; j < 4; j++) {\n"
       "      const bool in = (int)(i0 + j) < a.length;\n"
       "      r.v[d][j] = in ? a.vals[d][i0 + j] : 0u;\n"
       "      r.ok[d] |= in ? (u32)a.nulls[d][i0 + j] << (8 * j) : 0u;\n"
       "    }\n"
       "  }\n";
  }
  o << "  for (int j = 0; j < 4; j++)\n"
       "    for (int q = 0; q < MQ; q++) r.m[j * MQ + q] = (int)(i0 + j) < a.length ? a.vals[ND][(u64)(i0 + j) * MQ + q] : 0u;\n"
       "}\n"
       // Murmur32Stream over the packed row (dim_layout.hpp): values, then the validity bytes — one more block
       // when there are four of them, otherwise the tail
       "__device__ __forceinline__ void eval4(const Raw &r, const Args &a, u32 i0, u32 (&hh)[4], u32 (&cv)[4], u32 (&cw)[4], u32 (&alive)[4]) {\n"
       "#pragma unroll\n"
       "  for (int j = 0; j < 4; j++) {\n"
       "    alive[j] = (int)(i0 + j) < a.length ? 1u : 0u;\n";
  if (sort64) {
    SlotLayout SL;
    SL.nd = nd;
    for (int d = 0; d < nd; d++) {
      SL.width[d] = width[d];
      SL.off[d] = SL.valueBytes;
      SL.valueBytes += width[d];
    }
    o << "    u64 h64;\n";
    gen_row_hash64(o, SL, [](int d) { return "r.v[" + std::to_string(d) + "][j]"; },
                   [](int d) { return "((r.ok[" + std::to_string(d) + "] >> (8 * j)) & 0xFFu)"; }, "h64", "    ");
    o << "    hh[j] = (u32)(h64 >> 32);\n"
         "    cv[j] = r.m[j];\n"
         "    cw[j] = (u32)h64;\n"
         "  }\n"
         "}\n";
  } else {
  o << "    u32 h = 0u, okbytes = 0u;\n"
       "#pragma unroll\n"
       "    for (int d = 0; d < ND; d++) { h = mix(h, r.v[d][j]); okbytes |= ((r.ok[d] >> (8 * j)) & 0xFFu) << (8 * d); }\n";
  if (nd == 4) o << "    h = mix(h, okbytes);\n";
  else o << "    { u32 k = okbytes * 0xcc9e2d51u; k = rotl(k, 15) * 0x1b873593u; h ^= k; }\n";
  o << "    h ^= " << 5 * nd << "u; h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;\n"
       "    hh[j] = h;\n"
       "    cv[j] = r.m[j * MQ];\n"
       "    cw[j] = MQ == 2 ? r.m[j * MQ + MQ - 1] : 0u;\n"
       "  }\n"
       "}\n";
  }
  o << "__device__ __forceinline__ void load_tile(Raw &r, const Args &a, u32 i0) { if ((int)(i0 + 3u) < a.length) load_full(r, a, i0); else if ((int)i0 < a.length) load_tail(r, a, i0); }\n"
       "__device__ __forceinline__ void eval4p(Raw &r, const Args &a, u32 &i0, u32 (&hh)[4], u32 (&cv)[4], u32 (&cw)[4], u32 (&alive)[4], u32 i0n) {\n"
       "  eval4(r, a, i0, hh, cv, cw, alive);\n"
       "  load_tile(r, a, i0n);\n"
       "}\n";
  // (sort64: the level-1 partition of the wide layout is the hash's top PB bits, or — a.pad set: a previous result whose row
  // hashes are not known is hashed again, rows in ascending hash order of which every tile falls into ONE such partition —
  // the LOW PB bits of the top-bits partition index (a.chunkTiles = 32 - total partition bits) XORed with a scramble of its
  // leading bits: the tiles one workgroup scans lie a multiple of a power of two apart, the plain low bits would repeat)
  if (sort64)
    kernel_body_lines16(o, "cw[j]", "sr_scan_rtc",
                        "(a.pad ? (((hh[j] >> a.chunkTiles) ^ ((((hh[j] >> a.chunkTiles) >> PB) * 0x9E3779B1u) >> 23)) & (NP - 1u)) "
                        ": (PB ? hh[j] >> (32 - (PB ? PB : 1)) : 0u))");
  else kernel_body_lines16(o, "cw[j]", "hr_scan_rtc");
  return o.str();
}


// ---- specialised merge ------------------------------------------------------------------------------
// One workgroup per partition, like merge_body<ND, true, 4> (hr_kernels.hpp) for the case the
// specialised scans produce: line records in region B only, previous groups (if any) read
// from their partition-grouped ranges, the whole hash range in one round.  What changes is the cost
// per record: the aggregate, the widening of the carried measure and the dimension expressions are
// literals (the generic kernel spends ~90 VALU + ~120 SALU instructions per record on dispatch), and
// the records a lane holds are probed together — their LDS key reads in flight, then one
// non-returning LDS atomic each for records that meet their group in the first slot (all of them,
// once the groups exist); only misses walk the probe loop.  A partition with more groups than the
// table holds raises a flag and the host runs the generic multi-round merge instead.
struct RtcMergeArgs {  // mirrors `struct MArgs` of the generated source
  const uint32_t *vals[kFusedCols];
  const uint8_t *nulls[kFusedCols];
  const uint32_t *recB;
  const uint32_t *countsB;
  const uint32_t *prevRanges;
  const uint8_t *prevDims;
  const uint8_t *prevValues;
  uint8_t *dimOut;
  uint8_t *outValues;
  uint32_t *outCount;
  uint32_t *outRanges;
  uint64_t prevCapacity, outCapacity;
  uint32_t bitOff[kFusedCols];
  uint32_t capB, streams, prevSize, chunkRows;
  uint64_t *phases;  // ARES_HR_PHASES=1: per-partition time stamps (diagnostics)
  uint32_t k[kNumConsts];
  uint32_t pad;
  const uint4 *recA;  // region A (read by kernels generated with `regionA` only)
  const uint32_t *cursorsA;
  uint64_t capA;
  // table images (kernels generated with `image`): kSlots uint4 per partition = [keys u32 x kSlots][positions u32 x kSlots]
  // [values u64 x kSlots]; one group count per partition; knownOut: leading rows of the output dimension vector that already
  // hold the query's groups
  const uint4 *imgIn;
  uint4 *imgOut;
  const uint32_t *imgInCount;
  uint32_t *imgOutCount;
  uint32_t *hostOut;  // the calling thread's mapped pinned slot (null: the host copies outCount back)
  uint32_t knownOut, pad2;
};
static_assert(sizeof(RtcMergeArgs) % 8 == 0, "MArgs is passed as one buffer");

// vectorVW = 0: records of the plan-sourced scan (4-byte carried measure, widened here; rows >= prevSize are
// source rows whose dimensions are re-evaluated from the plan's columns), 16-byte lines or — `compact` — compact
// lines.  vectorVW = 4 / 8: records of the vector-sourced scan (the whole value travels; every row, old or new,
// is a row of the input vectors).
// regionA: the partition's records also come from region A — 16-byte {row, hash, value} records, what the TABLE-mode
// scan emits (one per group and workgroup) — ahead of the region-B runs.
// image: the partition's LDS table persists between the HashReduce calls of a query (hash_reduce_lds.hip "table image"):
//   1  the kernel as above, and at the end it leaves its table in HBM — keys (NEWG cleared), the output position of every
//      group where the representative row stood, values: 128 KB per partition, coalesced stores;
//   2  the kernel STARTS from the previous call's image (coalesced loads instead of re-hashing and re-inserting every
//      previous group), takes the batch's records through it, and emits only what is new: dimension rows of the groups
//      first seen in this batch, appended behind the previous result (a group keeps its position for the life of the
//      query), the dimension rows the output vector has not seen yet copied over from the input vector, and the table
//      image again.  The measure vector is NOT written: it is defined by the image (materialised by
//      hr_image_values_kernel when somebody reads it).
std::string generate_merge(const FusedPlanD &plan, int nd, int partBits, const AggSpec &a, const hr::Widen &w, int vectorVW = 0,
                           bool compact = false, bool regionA = false, int image = 0) {
  if (nd < 1 || nd > kFusedDims) return "";
  if (image && vectorVW) return "";
  if (compact && (vectorVW || partBits < 3)) return "";
  if (partBits < 2) return "";  // the 32-bit table keys need two spare hash bits (small inputs: the generic merge)
  const SlotLayout SL = slot_layout(plan, nd);
  if (!SL.ok || (vectorVW && !SL.all4)) return "";
  std::ostringstream o;
  o << times5_text() << kPrelude
    << "struct MArgs { const u32 *vals[" << kFusedCols << "]; const u8 *nulls[" << kFusedCols << "]; const u32 *recB; const u32 *countsB;\n"
       "  const u32 *prevRanges; const u8 *prevDims; const u8 *prevValues; u8 *dimOut; u8 *outValues; u32 *outCount; u32 *outRanges;\n"
       "  u64 prevCapacity, outCapacity; u32 bitOff[" << kFusedCols << "]; u32 capB, streams, prevSize, chunkRows; u64 *phases; u32 k[" << kNumConsts << "]; u32 pad;\n"
       "  const uint4 *recA; const u32 *cursorsA; u64 capA;\n"
       "  const uint4 *imgIn; uint4 *imgOut; const u32 *imgInCount; u32 *imgOutCount; u32 *hostOut; u32 knownOut, pad2; };\n"
       // The call's result words (groups, region overflow, stale ranges, crowded partition) reach the host without a copy
       // command behind the kernel: the workgroup that finishes last writes them into the calling thread's mapped pinned slot
       // (a.hostOut; outCount[4] is the ticket).  The host only waits for the stream.
       "#define FINISH() { __syncthreads(); if (threadIdx.x == 0u && a.hostOut) { __threadfence(); \\\n"
       "  if (atomicAdd(a.outCount + 4, 1u) == gridDim.x - 1u) { __threadfence(); const volatile u32 *oc = a.outCount; \\\n"
       "    a.hostOut[0] = oc[0]; a.hostOut[1] = oc[1]; a.hostOut[2] = oc[2]; a.hostOut[3] = oc[3]; } } }\n"
       "#define ND " << nd << "\n#define VB " << SL.valueBytes << "\n#define PB " << partBits << "\n#define NP " << (1 << partBits) << "\n"
       "#define SLOTS " << hr::kSlots << "\n#define LIMIT " << hr::kMergeLimit << "u\n#define RANGEWORDS " << hr::kRangeWords
    << "\n#define MAXRANGES " << hr::kMaxRanges << "u\n"
       // Table keys are 32 bits: within a partition the top PB bits of every hash are the partition's number, so a
       // key keeps the other 32 - PB bits (HMASK) and uses two of the freed bits as flags — OCC (the slot is taken:
       // an empty slot is 0, "key matches" is one masked compare) and NEWG (the group was first seen in THIS call's
       // records: only then can a record lower the group's representative row — groups that come from the previous
       // result have rows below every row of the batch).  Four keys = one 16-byte LDS read per probe.
       "#define HMASK ((1u << (32 - PB)) - 1u)\n#define OCC 0x40000000u\n#define NEWG 0x80000000u\n";
  gen_widen(o, w);
  if (!gen_agg(o, a)) return "";
  if (phases_enabled())
    o << "#define STAMP(k) if (threadIdx.x == 0u) a.phases[(u64)blockIdx.x * 8u + (k)] = __builtin_amdgcn_s_memrealtime();\n";
  else
    o << "#define STAMP(k)\n";
  // The table is probed by buckets of four keys (32 bytes, two LDS reads): a record meets its group in
  // its home bucket ~93 % of the time at this load, so a wavefront rarely takes more than two or three
  // rounds — with one key per probe the longest probe sequence among 64 lanes paced every wave.  A
  // lane's records go through the rounds together: their bucket reads are in flight at once.
  // Claims only ever turn the LOWEST empty slot of a bucket into a key, so a hash cannot end up twice.
  o << "#define BUCKETS (SLOTS / 4)\n"
       "struct Probe { u32 b, slot; bool done, isNew; };\n"
       // One round for one record, written without branches except for the rare claim: the merge is bound by
       // instruction issue (divergent control flow costs ~6 scalar instructions per `if`), not by LDS or HBM.
       // want = the occupied key of the record's hash; claimKey = want, with NEWG for a record of the batch.
       "__device__ __forceinline__ void probe_round(u32 *sKeys, u32 *sClaimed, u32 *sOverflow, Probe &q, u32 want, u32 claimKey) {\n"
       "  const uint4 k = *reinterpret_cast<const uint4 *>(sKeys + 4u * q.b);\n"
       "  const bool e0 = k.x == 0u, e1 = k.y == 0u, e2 = k.z == 0u, e3 = k.w == 0u;\n"
       "  const bool m0 = (k.x & ~NEWG) == want, m1 = (k.y & ~NEWG) == want, m2 = (k.z & ~NEWG) == want, m3 = (k.w & ~NEWG) == want;\n"
       "  const bool anyM = m0 | m1 | m2 | m3, anyE = e0 | e1 | e2 | e3;\n"
       "  const u32 mi = m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u, ei = e0 ? 0u : e1 ? 1u : e2 ? 2u : 3u;\n"
       "  const u32 mk = m0 ? k.x : m1 ? k.y : m2 ? k.z : k.w;\n"
       "  const bool active = !q.done, hit = active & anyM;\n"
       "  q.slot = hit ? 4u * q.b + mi : q.slot;\n"
       "  q.isNew = hit ? (mk & NEWG) != 0u : q.isNew;\n"
       "  bool claimed = false;\n"
       "  if (active & !anyM & anyE) {\n"  // a group that is new in this partition: rare once the groups exist
       "    u32 expected = 0u;\n"
       "    if (__hip_atomic_compare_exchange_strong(sKeys + 4u * q.b + ei, &expected, claimKey, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {\n"
       "      if (__hip_atomic_fetch_add(sClaimed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= LIMIT)\n"
       "        __hip_atomic_store(sOverflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "      q.slot = 4u * q.b + ei; q.isNew = true; claimed = true;\n"  // (the claimer lowers the row too: rows start at ~0)
       "    }\n"  // lost the slot: the same bucket again next round (the winner may be this very group)
       "  }\n"
       "  q.b = (active & !anyM & !anyE) ? (q.b + 1u) & (BUCKETS - 1u) : q.b;\n"
       "  q.done = q.done | hit | claimed;\n"
       "}\n"
       // one record through the general probe loop; batch = a record of this call's batch (may found a NEWG group)
       "__device__ __forceinline__ void insert(u32 *sKeys, u32 *sRows, u64 *sVals, u32 *sClaimed, u32 *sOverflow, u32 row, u32 h, u64 value, bool batch) {\n"
       "  const u32 want = (h & HMASK) | OCC;\n"
       "  Probe q; q.b = h & (BUCKETS - 1u); q.slot = 0u; q.done = false; q.isNew = false;\n"
       "  for (u32 tries = 0u; tries < 4u * BUCKETS && !q.done; tries++) probe_round(sKeys, sClaimed, sOverflow, q, want, batch ? want | NEWG : want);\n"
       "  if (q.done) {\n"
       "    if (q.isNew || !batch) __hip_atomic_fetch_min(sRows + q.slot, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "    agg(sVals + q.slot, value);\n"
       "  } else {\n"
       "    __hip_atomic_store(sOverflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"  // table full: the round is void anyway
       "  }\n"
       "}\n";
  // Records arrive in segments of 64 sixteen-byte units (one per lane), four segments per register stage — of
  // one long run or of four short ones (small batches leave ~16 records per run: a stage per run would make the
  // merge a chain of dependent loads).  Round one looks at every record's home bucket with straight-line
  // code (no claim, no advance): a record whose group sits there — ~93 % once the groups exist — costs one
  // LDS atomic more.  The rest (the group lives further on, or is new) is queued per wavefront in LDS and
  // taken through the general probe loop 64 at a time, every lane busy: run per record where it occurs,
  // that loop would execute for a handful of lanes after nearly every segment.
  if (vectorVW == 8)  // four words per queued record: a smaller queue, drained from 32 entries on (LDS is full)
    o << "#define QCAP 96u\n#define QW 4u\n#define QDRAIN 32u\n#define VALB(z, w) ((((u64)(w)) << 32) | (z))\n";
  else if (vectorVW == 4)
    o << "#define QCAP 128u\n#define QW 3u\n#define QDRAIN 64u\n#define VALB(z, w) ((u64)(z))\n";
  else
    o << "#define QCAP 128u\n#define QW 3u\n#define QDRAIN 64u\n#define VALB(z, w) widen(z)\n";
  o << "struct Seg { const uint4 *ptr; u32 n, rem, rb; };\n"
       "struct Stage { uint4 r[4]; u32 n[4], rem[4], rb[4]; };\n"
       "__device__ __forceinline__ void drain(u32 *queue, u32 first, u32 count, u32 lane, u32 *sKeys, u32 *sRows, u64 *sVals, u32 *sClaimed, u32 *sOverflow) {\n"
       "  asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n"
       "  if (lane < count) {\n"
       "    const u32 e = QW * (first + lane);\n"
       "    const u32 row = queue[e], h = queue[e + 1u], z = queue[e + 2u], w = QW == 4u ? queue[e + QW - 1u] : 0u;\n"
       "    insert(sKeys, sRows, sVals, sClaimed, sOverflow, row, h, VALB(z, w), true);\n"
       "  }\n"
       "  asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n"
       "}\n"
       // one record in round one
       "__device__ __forceinline__ void consume_one(bool valid, u32 row, u32 h, u32 z, u32 w, u32 lane, u32 *sKeys, u32 *sRows, u64 *sVals, u32 *sClaimed, u32 *sOverflow, u32 *queue, u32 &qn) {\n"
       "  const u32 b = h & (BUCKETS - 1u), want = (h & HMASK) | OCC;\n"
       "  const uint4 k = *reinterpret_cast<const uint4 *>(sKeys + 4u * b);\n"
       "  const bool m0 = (k.x & ~NEWG) == want, m1 = (k.y & ~NEWG) == want, m2 = (k.z & ~NEWG) == want, m3 = (k.w & ~NEWG) == want;\n"
       "  const bool hit = valid && (m0 || m1 || m2 || m3);\n"
       "  if (hit) {\n"
       "    const u32 mi = m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u;\n"
       "    const u32 mk = m0 ? k.x : m1 ? k.y : m2 ? k.z : k.w;\n"
       "    if (mk & NEWG) __hip_atomic_fetch_min(sRows + 4u * b + mi, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "    agg(sVals + 4u * b + mi, VALB(z, w));\n"
       "  }\n"
       "  const bool pend = valid && !hit;\n"
       "  const u64 m = __ballot(pend);\n"
       "  if (m) {\n"
       "    if (pend) {\n"
       "      const u32 e = QW * (qn + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)));\n"
       "      queue[e] = row; queue[e + 1u] = h; queue[e + 2u] = z;\n"
       "      if (QW == 4u) queue[e + QW - 1u] = w;\n"
       "    }\n"
       "    qn += (u32)__popcll(m);\n"
       "    if (qn >= QDRAIN) { const u32 take = qn < 64u ? qn : 64u; qn -= take; drain(queue, qn, take, lane, sKeys, sRows, sVals, sClaimed, sOverflow); }\n"
       "  }\n"
       "}\n";
  if (compact)
    // a 16-byte unit holds two 8-byte slots of a line: lanes 8l .. 8l + 7 hold line l of the segment, the first
    // slot of lanes 8l and 8l + 4 is a header (low 9 row bits of the half-line's seven records)
    o << "__device__ __forceinline__ void consume(const Stage &s, u32 lane, u32 p, u32 *sKeys, u32 *sRows, u64 *sVals, u32 *sClaimed, u32 *sOverflow, u32 *queue, u32 &qn) {\n"
         "  const u32 pos = 2u * (lane & 3u);\n"                       // place of the unit's first slot within its half-line
         "  const u32 recBase = (lane >> 3) * 14u + 7u * ((lane >> 2) & 1u);\n"  // record number (within the segment) of the half-line's first record
         "#pragma unroll\n"
         "  for (int k = 0; k < 4; k++) {\n"
         "    const u64 hdr = ((u64)(u32)__builtin_amdgcn_mov_dpp((int)s.r[k].y, 0x00, 0xF, 0xF, true) << 32) | (u32)__builtin_amdgcn_mov_dpp((int)s.r[k].x, 0x00, 0xF, 0xF, true);\n"
         "#pragma unroll\n"
         "    for (u32 j = 0u; j < 2u; j++) {\n"
         "      const u32 ph = pos + j;\n"                             // 0 = the header slot
         "      const u32 k7 = ph ? ph - 1u : 0u;\n"
         "      const bool valid = lane < s.n[k] && ph != 0u && recBase + k7 < s.rem[k];\n"
         "      const u32 z = j ? s.r[k].z : s.r[k].x, hw = j ? s.r[k].w : s.r[k].y;\n"
         "      const u32 lo9 = (u32)(hdr >> (9u * k7)) & 511u;\n"
         "      const u32 h = (p << (32 - PB)) | (hw >> PB);\n"
         "      const u32 row = s.rb[k] + (((hw & ((1u << PB) - 1u)) << 9) | lo9);\n"
         "      consume_one(valid, row, h, z, 0u, lane, sKeys, sRows, sVals, sClaimed, sOverflow, queue, qn);\n"
         "    }\n"
         "  }\n"
         "}\n";
  else
    o << "__device__ __forceinline__ void consume(const Stage &s, u32 lane, u32 p, u32 *sKeys, u32 *sRows, u64 *sVals, u32 *sClaimed, u32 *sOverflow, u32 *queue, u32 &qn) {\n"
         "#pragma unroll\n"
         "  for (int k = 0; k < 4; k++) {\n"
         "    const bool valid = lane < s.n[k] && s.r[k].x != 0xFFFFFFFFu;\n"  // not past the segment / padding of the run's last line
         "    consume_one(valid, s.r[k].x, s.r[k].y, s.r[k].z, s.r[k].w, lane, sKeys, sRows, sVals, sClaimed, sOverflow, queue, qn);\n"
         "  }\n"
         "}\n";
  // dimensions of a source row (hr::fused_eval_row), for groups that are new in this batch
  if (!vectorVW) o << "__device__ __forceinline__ void eval_row(const MArgs &a, u32 row, u32 (&bits)[ND], u32 (&ok)[ND]) {\n";
  for (int d = 0; d < nd && !vectorVW; d++) {
    const FusedExpr &e = plan.dims[d];
    if (!plain_store(e.f.rk, e.outKind) || (SL.width[d] != 4 && !int_kind(e.f.rk))) return "";
    const int c = e.col;
    o << "  {\n    const u32 v = " << column_elem(fused_col_step(plan, c), e.f.akind == K_I32, "a.vals[" + std::to_string(c) + "]", "row") << ";\n";
    if (plan.cols[c].nulls) o << "    const u32 bit = row + a.bitOff[" << c << "]; const u32 okb = (a.nulls[" << c << "][bit >> 3] >> (bit & 7u)) & 1u;\n";
    else o << "    const u32 okb = 1u;\n";
    o << "    u32 x;\n";
    if (!gen_value(e.f, o, "v", "okb", "x", const_name(const_slot_dim(d)))) return "";
    o << "    bits[" << d << "] = x; ok[" << d << "] = okb;\n  }\n";
  }
  if (!vectorVW) o << "}\n";
  const bool wide = a.width == 8;
  o << "extern \"C\" __global__ void __launch_bounds__(1024) hr_merge_rtc(MArgs a) {\n"
       "  __shared__ __attribute__((aligned(16))) u32 sKeys[SLOTS];\n"
       "  __shared__ __attribute__((aligned(16))) u32 sRows[SLOTS];\n"
       "  __shared__ __attribute__((aligned(16))) u64 sVals[SLOTS];\n"
       "  __shared__ u32 sRunCount[256];\n"
       "  __shared__ u32 sQueue[16u * QCAP * QW];\n"
       "  __shared__ u32 sClaimed, sOverflow, sCount, sBase, sEmit;\n"
       "  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, p = blockIdx.x;\n"
       "  STAMP(0)\n";
  if (image == 2)  // the table as the query's previous HashReduce left it
    o << "  {\n"
         "    const uint4 *img = a.imgIn + (u64)p * SLOTS;\n"
         "    for (u32 i = tid; i < SLOTS / 4u; i += 1024u) {\n"
         "      reinterpret_cast<uint4 *>(sKeys)[i] = img[i];\n"
         "      reinterpret_cast<uint4 *>(sRows)[i] = img[SLOTS / 4u + i];\n"
         "    }\n"
         "    for (u32 i = tid; i < SLOTS / 2u; i += 1024u) reinterpret_cast<uint4 *>(sVals)[i] = img[SLOTS / 2u + i];\n"
         "    if (tid == 0u) { sClaimed = a.imgInCount[p]; sOverflow = 0u; sCount = 0u; sEmit = 0u; }\n"
         "  }\n";
  else
    o << "  for (u32 s = tid; s < SLOTS; s += 1024u) { sKeys[s] = 0u; sRows[s] = 0xFFFFFFFFu; sVals[s] = IDENT; }\n"
         "  if (tid == 0u) { sClaimed = 0u; sOverflow = 0u; sCount = 0u; sEmit = 0u; }\n";
  o << "  const u32 G = a.streams;\n"
       "  if (tid < G) sRunCount[tid] = a.countsB[(u64)tid * NP + p];\n"
       "  const u32 *ranges = a.prevRanges ? a.prevRanges + (u64)p * RANGEWORDS : nullptr;\n"
       "  u32 nRanges = ranges ? ranges[0] : 0u;\n"
       "  if (nRanges > MAXRANGES) { if (tid == 0u) a.outCount[2] = 1u; nRanges = 0u; }\n"
       "  __syncthreads();\n"
       "  STAMP(1)\n"
       // previous groups of this partition (always the lowest row indices: they stay the representatives)
       "  {\n"
       "    const u8 *nullsIn = a.prevDims + (u64)VB * a.prevCapacity;\n"
       "    for (u32 r = 0u; r < nRanges; r++) {\n"
       "      const u32 start = ranges[1u + 2u * r], cnt = ranges[2u + 2u * r];\n"
       "      for (u32 i0 = 0u; i0 < cnt; i0 += 4096u) {\n"  // four groups per lane: their loads are in flight together
       "        u32 hh[4], rr[4]; u64 vv[4]; bool okk[4];\n"
       "#pragma unroll\n"
       "        for (int k = 0; k < 4; k++) {\n"
       "          const u32 i = i0 + (u32)k * 1024u + tid;\n"
       "          okk[k] = i < cnt;\n"
       "          const u32 row = start + (okk[k] ? i : 0u);\n"
       "          rr[k] = row;\n"
       "          okk[k] = okk[k] && row < a.prevSize;\n"
       "          const u32 safe = row < a.prevSize ? row : 0u;\n"
       "          u32 h = 0u;\n";
  if (SL.all4) {
    o << "          u32 okbytes = 0u;\n"
         "#pragma unroll\n"
         "          for (int d = 0; d < ND; d++) {\n"
         "            h = mix(h, *reinterpret_cast<const u32 *>(a.prevDims + (u64)(4 * d) * a.prevCapacity + 4ull * safe));\n"
         "            okbytes |= (u32)nullsIn[(u64)d * a.prevCapacity + safe] << (8 * d);\n"
         "          }\n";
    if (nd == 4) o << "          h = mix(h, okbytes);\n";
    else o << "          { u32 kk = okbytes * 0xcc9e2d51u; kk = rotl(kk, 15) * 0x1b873593u; h ^= kk; }\n";
    o << "          h ^= " << 5 * nd << "u; h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;\n";
  } else {
    for (int d = 0; d < nd; d++)
      o << "          const u32 pv" << d << " = " << slot_load(SL, d, "a.prevDims", "a.prevCapacity", "safe") << ", po" << d
        << " = (u32)nullsIn[(u64)" << d << " * a.prevCapacity + safe];\n";
    gen_row_hash(o, SL, [](int d) { return "pv" + std::to_string(d); }, [](int d) { return "po" + std::to_string(d); }, "h", "          ");
  }
  o <<
       "          hh[k] = h;\n"
    << (wide ? "          vv[k] = reinterpret_cast<const u64 *>(a.prevValues)[safe];\n"
             : "          vv[k] = reinterpret_cast<const u32 *>(a.prevValues)[safe];\n")
    << "        }\n"
       "#pragma unroll\n"
       "        for (int k = 0; k < 4; k++) {\n"
       "          const u32 i = i0 + (u32)k * 1024u + tid;\n"
       "          if (i >= cnt) continue;\n"
       "          if (!okk[k] || (PB && (hh[k] >> (32 - (PB ? PB : 1))) != p)) { a.outCount[2] = 1u; continue; }\n"
       "          insert(sKeys, sRows, sVals, &sClaimed, &sOverflow, rr[k], hh[k], vv[k], false);\n"
       "        }\n"
       "      }\n"
       "    }\n"
       "  }\n"
       "  __syncthreads();\n"
       "  STAMP(2)\n";
  if (regionA)  // what the TABLE-mode scan left in region A: one record per group and scanning workgroup
    o << "  {\n"
         "    const u32 cur = a.cursorsA[p];\n"
         "    const u32 nA = cur < a.capA ? cur : (u32)a.capA;\n"
         "    const uint4 *recs = a.recA + (u64)p * a.capA;\n"
         "    for (u32 i = tid; i < nA; i += 1024u) {\n"
         "      const uint4 r = recs[i];\n"
         "      insert(sKeys, sRows, sVals, &sClaimed, &sOverflow, r.x, r.y, ((u64)r.w << 32) | r.z, true);\n"
         "    }\n"
         "  }\n"
         "  __syncthreads();\n";
  // Small batches (live batches: 2 Mi rows, 512 tiles, two per scanning workgroup) leave every partition a few hundred runs
  // of a dozen or two records — one to three lines each.  Walking them run by run, sixteen per wavefront, is a chain of
  // dependent loads (20 us of a partition's 33 at 2 Mi rows); when no run is longer than four lines, the first L lines of
  // EVERY run of the partition are fetched at once instead (L = lines of the longest run): a few 16-byte loads per lane, all
  // in flight together.
  o << "  u32 maxRun = 0u;\n"
       "  for (u32 g = lane; g < G; g += 64u) { const u32 c = sRunCount[g]; maxRun = c > maxRun ? c : maxRun; }\n"
       "#pragma unroll\n"
       "  for (int off = 32; off > 0; off >>= 1) { const u32 t = (u32)__shfl_xor((int)maxRun, off); maxRun = t > maxRun ? t : maxRun; }\n"
       "  const u32 LPR = (maxRun + " << (compact ? "13u) / 14u" : "7u) / 8u") << ";\n"   // lines of the longest run
       "  if (G > 0u && LPR <= 4u && LPR <= a.capB" << (compact ? "" : " / 8u") << ") {\n"
       "    u32 qn = 0u;\n"
       "    u32 *queue = sQueue + wave * (QCAP * QW);\n"
       "    const u32 units = G * 8u * LPR;\n"
       "    for (u32 base = 0u; base < units; base += 4096u) {\n"
       "      Stage s;\n"
       "#pragma unroll\n"
       "      for (int k = 0; k < 4; k++) {\n"
       "        const u32 u = base + (u32)k * 1024u + tid;\n"  // unit u = lane (u & 7) of line (u >> 3) % LPR of stream (u >> 3) / LPR
       "        const bool in = u < units;\n"
       "        const u32 ul = in ? u >> 3 : 0u, g = ul / LPR, l = ul - g * LPR, cnt = in ? sRunCount[g] : 0u;\n"
    << (compact ? "        const u32 here = cnt > l * 14u ? cnt - l * 14u : 0u;\n"   // records of the run in this line and behind it
                  "        s.r[k] = reinterpret_cast<const uint4 *>(a.recB)[(((u64)g * NP + p) * a.capB + l) * 8u + (u & 7u)];\n"
                  "        s.n[k] = here ? 64u : 0u; s.rem[k] = (lane >> 3) * 14u + here; s.rb[k] = a.prevSize + g * a.chunkRows;\n"
                : "        const u32 here = cnt > l * 8u ? cnt - l * 8u : 0u;\n"
                  "        s.r[k] = reinterpret_cast<const uint4 *>(a.recB)[((u64)g * NP + p) * a.capB + l * 8u + (u & 7u)];\n"
                  "        s.n[k] = (u & 7u) < here ? 64u : 0u; s.rem[k] = 0u; s.rb[k] = 0u;\n")
    << "      }\n"
       "      consume(s, lane, p, sKeys, sRows, sVals, &sClaimed, &sOverflow, queue, qn);\n"
       "    }\n"
       "    while (qn) { const u32 take = qn < 64u ? qn : 64u; qn -= take; drain(queue, qn, take, lane, sKeys, sRows, sVals, &sClaimed, &sOverflow); }\n"
       "  } else\n"
       // the partition's runs: every wavefront streams whole runs, three register stages
       "  if (G > 0u) {\n"
       "    const uint4 *dummy = reinterpret_cast<const uint4 *>(a.recB);\n"
       "    u32 j = 0u, off = 0u, qn = 0u;\n"
       "    u32 *queue = sQueue + wave * (QCAP * QW);\n"
       // this wavefront's runs are wave, wave + 16, ...: lane i keeps the length of the i-th of them, so that
       // walking the runs costs no LDS round trip per segment
       "    const u32 myRuns = (G + 15u - wave) / 16u;\n"
       "    const u32 myCnt = lane < myRuns ? sRunCount[wave + 16u * lane] : 0u;\n";
  if (compact)
    o << "    auto next = [&]() -> Seg {\n"
         "      Seg c{dummy, 0u, 0u, 0u};\n"
         "      while (j < myRuns) {\n"
         "        const u32 cnt = (u32)__builtin_amdgcn_readlane((int)myCnt, (int)j);\n"
         "        const u32 units = ((cnt + 13u) / 14u) * 8u;\n"         // whole lines, eight 16-byte units each
         "        if (off < units) {\n"
         "          c.ptr = reinterpret_cast<const uint4 *>(a.recB) + ((u64)(wave + 16u * j) * NP + p) * a.capB * 8u + off;\n"
         "          c.n = units - off < 64u ? units - off : 64u;\n"
         "          c.rem = cnt - (off >> 3) * 14u;\n"
         "          c.rb = a.prevSize + (wave + 16u * j) * a.chunkRows;\n"
         "          off += 64u;\n"
         "          break;\n"
         "        }\n"
         "        j++; off = 0u;\n"
         "      }\n"
         "      return c;\n"
         "    };\n";
  else
    o << "    auto next = [&]() -> Seg {\n"
         "      Seg c{dummy, 0u, 0u, 0u};\n"
         "      while (j < myRuns) {\n"
         "        const u32 cnt = (u32)__builtin_amdgcn_readlane((int)myCnt, (int)j);\n"
         "        if (off < cnt) {\n"
         "          c.ptr = reinterpret_cast<const uint4 *>(a.recB) + ((u64)(wave + 16u * j) * NP + p) * a.capB + off;\n"
         "          c.n = cnt - off < 64u ? cnt - off : 64u;\n"
         "          off += 64u;\n"
         "          break;\n"
         "        }\n"
         "        j++; off = 0u;\n"
         "      }\n"
         "      return c;\n"
         "    };\n";
  o << "    auto load = [&](Stage &s) {\n"
       "#pragma unroll\n"
       "      for (int k = 0; k < 4; k++) {\n"
       "        const Seg c = next();\n"
       "        s.n[k] = c.n; s.rem[k] = c.rem; s.rb[k] = c.rb;\n"
       "        s.r[k] = c.ptr[lane < c.n ? lane : (c.n ? c.n - 1u : 0u)];\n"
       "      }\n"
       "    };\n"
       // three register stages: two stages of loads are always in flight behind the one being consumed
       "    Stage s0, s1, s2;\n"
       "    load(s0);\n"
       "    load(s1);\n"
       "    for (;;) {\n"
       "      load(s2);\n"
       "      if (!s0.n[0]) break;\n"
       "      consume(s0, lane, p, sKeys, sRows, sVals, &sClaimed, &sOverflow, queue, qn);\n"
       "      load(s0);\n"
       "      if (!s1.n[0]) break;\n"
       "      consume(s1, lane, p, sKeys, sRows, sVals, &sClaimed, &sOverflow, queue, qn);\n"
       "      load(s1);\n"
       "      if (!s2.n[0]) break;\n"
       "      consume(s2, lane, p, sKeys, sRows, sVals, &sClaimed, &sOverflow, queue, qn);\n"
       "    }\n"
       "    while (qn) { const u32 take = qn < 64u ? qn : 64u; qn -= take; drain(queue, qn, take, lane, sKeys, sRows, sVals, &sClaimed, &sOverflow); }\n"
       "  }\n"
       "  __syncthreads();\n"
       "  STAMP(3)\n"
       "  if (sOverflow) { if (tid == 0u) a.outCount[3] = 1u; FINISH() return; }\n";  // more groups than one table: the generic merge takes over
  // the image's three planes leave (or reach) a partition with 16-byte accesses, consecutive lanes consecutive addresses
  const char *kStoreKeysPos =
      "    for (u32 i = tid; i < SLOTS / 4u; i += 1024u) {\n"
      "      img[i] = reinterpret_cast<const uint4 *>(sKeys)[i];\n"
      "      img[SLOTS / 4u + i] = reinterpret_cast<const uint4 *>(sRows)[i];\n"
      "    }\n";
  const char *kStoreVals = "    for (u32 i = tid; i < SLOTS / 2u; i += 1024u) img[SLOTS / 2u + i] = reinterpret_cast<const uint4 *>(sVals)[i];\n";
  if (image == 2) {
    // ---- groups first seen in this batch: their dimension rows, appended behind the previous result -------------------
    o << "  u32 mineNew = 0u;\n"
         "#pragma unroll\n"
         "  for (int k = 0; k < SLOTS / 1024; k++) mineNew += (sKeys[tid + (u32)k * 1024u] & NEWG) != 0u;\n"
         "  if (mineNew) __hip_atomic_fetch_add(&sCount, mineNew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
         "  __syncthreads();\n"
         "  const u32 totalNew = sCount;\n"
         "  if (tid == 0u) sBase = a.prevSize + (totalNew ? atomicAdd(a.outCount, totalNew) : 0u);\n"  // (outCount counts the NEW groups)
         "  __syncthreads();\n"
         "  STAMP(4)\n"
         "  u8 *nullsOut = a.dimOut + (u64)VB * a.outCapacity;\n"
         "  if (totalNew) {\n"
         "#pragma unroll\n"
         "    for (int half = 0; half < 2; half++) {\n"
         "      u32 dv[4][ND], nv[4][ND], at[4]; bool has[4];\n"
         "#pragma unroll\n"
         "      for (int kk = 0; kk < 4; kk++) {\n"
         "        const u32 s = tid + (u32)(half * 4 + kk) * 1024u;\n"
         "        has[kk] = (sKeys[s] & NEWG) != 0u;\n"
         "        const u64 m = __ballot(has[kk]);\n"
         "        u32 waveBase = 0u;\n"
         "        if (lane == 0u && m) waveBase = __hip_atomic_fetch_add(&sEmit, (u32)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
         "        waveBase = (u32)__builtin_amdgcn_readfirstlane((int)waveBase);\n"
         "        at[kk] = sBase + waveBase + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));\n"
         "        if (has[kk]) eval_row(a, sRows[s] - a.prevSize, dv[kk], nv[kk]);\n"  // (a new group's representative is a row of the batch)
         "      }\n"
         "#pragma unroll\n"
         "      for (int kk = 0; kk < 4; kk++) {\n"
         "        if (!has[kk]) continue;\n"
         "        const u32 s = tid + (u32)(half * 4 + kk) * 1024u;\n";
    for (int d = 0; d < nd; d++)
      o << "        " << slot_store(SL, d, "a.dimOut", "a.outCapacity", "at[kk]", "dv[kk][" + std::to_string(d) + "]") << " nullsOut[(u64)" << d
        << " * a.outCapacity + at[kk]] = (u8)nv[kk][" << d << "];\n";
    o << "        sRows[s] = at[kk];\n"       // from now on the slot holds the group's position
         "        sKeys[s] &= ~NEWG;\n"
         "      }\n"
         "    }\n"
         "  }\n"
         "  __syncthreads();\n"
         // ---- the image again: values always; keys and positions unless the output's image already holds this very set
         // (a partition's count only grows, and both images descend from one table: equal counts = equal key planes)
         "  {\n"
         "    uint4 *img = a.imgOut + (u64)p * SLOTS;\n"
         "    const u32 newCount = a.imgInCount[p] + totalNew;\n"
         "    if (a.imgOutCount[p] != newCount) {\n"
      << kStoreKeysPos
      << "    }\n"
      << kStoreVals
      << "    __syncthreads();\n"
         "    if (tid == 0u) a.imgOutCount[p] = newCount;\n"
         "  }\n"
         // ---- dimension rows [knownOut, prevSize) the output vector has not seen yet: this workgroup's share, copied over
         "  if (a.knownOut < a.prevSize) {\n"
         "    const u8 *nullsIn = a.prevDims + (u64)VB * a.prevCapacity;\n"
         "    const u32 n = a.prevSize - a.knownOut, share = (n + NP - 1u) / NP;\n"
         "    const u32 lo = a.knownOut + p * share, hi = lo + share < a.prevSize ? lo + share : a.prevSize;\n"
         "    for (u32 r = lo + tid; r < hi; r += 1024u) {\n";
    for (int d = 0; d < nd; d++)
      o << "      " << slot_store(SL, d, "a.dimOut", "a.outCapacity", "r", slot_load(SL, d, "a.prevDims", "a.prevCapacity", "r")) << " nullsOut[(u64)" << d
        << " * a.outCapacity + r] = nullsIn[(u64)" << d << " * a.prevCapacity + r];\n";
    o << "    }\n"
         "  }\n"
         "  STAMP(5)\n"
         "  FINISH()\n"
         "}\n";
    return o.str();
  }
  o << // emit: count occupied slots, reserve output rows once, then copy (as hr::merge_body)
       "  u32 mineCount = 0u;\n"
       "#pragma unroll\n"
       "  for (int k = 0; k < SLOTS / 1024; k++) mineCount += sKeys[tid + (u32)k * 1024u] != 0u;\n"
       "  if (mineCount) __hip_atomic_fetch_add(&sCount, mineCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "  __syncthreads();\n"
       "  const u32 total = sCount;\n"
       "  if (tid == 0u) {\n"
       "    u32 base = 0u;\n"
       "    if (total) base = atomicAdd(a.outCount, total);\n"
       "    sBase = base;\n"
       "    if (a.outRanges) { u32 *o = a.outRanges + (u64)p * RANGEWORDS; o[0] = total ? 1u : 0u; o[1] = base; o[2] = total; }\n"
       "  }\n"
       "  __syncthreads();\n"
       "  STAMP(4)\n"
    << (image == 1 ? "" : "  if (!total) { FINISH() return; }\n")  // (an empty partition leaves an empty image)
    << "  const u8 *nullsIn = a.prevDims + (u64)VB * a.prevCapacity;\n"
       "  u8 *nullsOut = a.dimOut + (u64)VB * a.outCapacity;\n"
       "#pragma unroll\n"
       "  for (int half = 0; half < 2; half++) {\n"
       "    u32 dv[4][ND], nv[4][ND], at[4]; bool has[4];\n"
       "#pragma unroll\n"
       "    for (int kk = 0; kk < 4; kk++) {\n"
       "      const u32 s = tid + (u32)(half * 4 + kk) * 1024u;\n"
       "      has[kk] = sKeys[s] != 0u;\n"
       "      const u64 m = __ballot(has[kk]);\n"
       "      u32 waveBase = 0u;\n"
       "      if (lane == 0u && m) waveBase = __hip_atomic_fetch_add(&sEmit, (u32)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "      waveBase = (u32)__builtin_amdgcn_readfirstlane((int)waveBase);\n"
       "      at[kk] = sBase + waveBase + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));\n"
       "      if (!has[kk]) continue;\n"
       "      const u32 row = sRows[s];\n"
    << (vectorVW ? "" : "      if (row >= a.prevSize) { eval_row(a, row - a.prevSize, dv[kk], nv[kk]); continue; }\n");
  if (SL.all4) {
    o << "#pragma unroll\n"
         "      for (int d = 0; d < ND; d++) {\n"
         "        dv[kk][d] = *reinterpret_cast<const u32 *>(a.prevDims + (u64)(4 * d) * a.prevCapacity + 4ull * row);\n"
         "        nv[kk][d] = nullsIn[(u64)d * a.prevCapacity + row];\n"
         "      }\n";
  } else {
    for (int d = 0; d < nd; d++)
      o << "      dv[kk][" << d << "] = " << slot_load(SL, d, "a.prevDims", "a.prevCapacity", "row") << "; nv[kk][" << d << "] = nullsIn[(u64)" << d
        << " * a.prevCapacity + row];\n";
  }
  o << "    }\n"
       "#pragma unroll\n"
       "    for (int kk = 0; kk < 4; kk++) {\n"
       "      if (!has[kk]) continue;\n"
       "      const u32 s = tid + (u32)(half * 4 + kk) * 1024u;\n";
  if (SL.all4) {
    o << "#pragma unroll\n"
         "      for (int d = 0; d < ND; d++) {\n"
         "        *reinterpret_cast<u32 *>(a.dimOut + (u64)(4 * d) * a.outCapacity + 4ull * at[kk]) = dv[kk][d];\n"
         "        nullsOut[(u64)d * a.outCapacity + at[kk]] = (u8)nv[kk][d];\n"
         "      }\n";
  } else {
    for (int d = 0; d < nd; d++)
      o << "      " << slot_store(SL, d, "a.dimOut", "a.outCapacity", "at[kk]", "dv[kk][" + std::to_string(d) + "]") << " nullsOut[(u64)" << d
        << " * a.outCapacity + at[kk]] = (u8)nv[kk][" << d << "];\n";
  }
  o << (wide ? "      reinterpret_cast<u64 *>(a.outValues)[at[kk]] = sVals[s];\n"
             : "      reinterpret_cast<u32 *>(a.outValues)[at[kk]] = (u32)sVals[s];\n")
    << (image == 1 ? "      sRows[s] = at[kk];\n      sKeys[s] &= ~NEWG;\n" : "")  // the image: a group's slot holds its position
    << "    }\n"
       "  }\n";
  if (image == 1)
    o << "  __syncthreads();\n"
         "  {\n"
         "    uint4 *img = a.imgOut + (u64)p * SLOTS;\n"
      << kStoreKeysPos << kStoreVals
      << "    if (tid == 0u) a.imgOutCount[p] = total;\n"
         "  }\n";
  o <<
       "  STAMP(5)\n"
       "  FINISH()\n"
       "}\n";
  return o.str();
}

// ---- kernel cache ------------------------------------------------------------------------------------
}  // namespace

struct RtcEntry {
  int device = 0;
  std::mutex m;
  std::condition_variable cv;
  bool ready = false;      // final: the module is loaded, or there is no kernel (fn == nullptr: generic path)
  bool codeReady = false;  // the code object is here (read from disk / compiled), a query thread has yet to load it
  bool fromDisk = false;
  std::vector<char> code;
  std::string diskPath;
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;  // nullptr once ready = the source does not compile / load: generic kernel
  std::atomic<uint64_t> lastUse{0};
  ~RtcEntry();
};

// Modules of evicted kernels.  Dropped from the cache and by every caller, a launch may still be executing — the module
// cannot be unloaded on the spot without synchronising the device, which would stall every query on it (round-3 review).
// They wait here instead (a code object is tens of kilobytes); only when a hundred have piled up are the older half
// unloaded behind ONE device synchronisation.
namespace {
std::mutex g_graveMutex;
std::vector<std::pair<int, hipModule_t>> g_graveyard;
constexpr size_t kGraveyardLimit = 128;
}  // namespace

RtcEntry::~RtcEntry() {
  if (!module) return;
  std::vector<std::pair<int, hipModule_t>> unload;
  {
    std::lock_guard<std::mutex> lock(g_graveMutex);
    g_graveyard.emplace_back(device, module);
    if (g_graveyard.size() >= kGraveyardLimit) {
      unload.assign(g_graveyard.begin(), g_graveyard.begin() + kGraveyardLimit / 2);
      g_graveyard.erase(g_graveyard.begin(), g_graveyard.begin() + kGraveyardLimit / 2);
    }
  }
  if (unload.empty()) return;
  int current = 0;
  const bool have = hipGetDevice(&current) == hipSuccess;
  int synced = -1;
  for (auto &dm : unload) {
    if (dm.first != synced) {
      (void)hipSetDevice(dm.first);
      (void)hipDeviceSynchronize();
      synced = dm.first;
    }
    (void)hipModuleUnload(dm.second);
  }
  if (have) (void)hipSetDevice(current);
  (void)hipGetLastError();
}

namespace {

struct RtcCache {
  std::mutex mu;
  std::unordered_map<std::string, std::shared_ptr<RtcEntry>> map;  // device + entry point + source -> kernel
  std::atomic<uint64_t> tick{0};
  std::atomic<int> pending{0};
  std::mutex idleMu;
  std::condition_variable idleCv;
  std::atomic<long> compiles{0}, diskHits{0}, evictions{0};
};
RtcCache &cache() {
  static RtcCache *c = new RtcCache;  // never destroyed: its modules must not outlive the HIP runtime's teardown order
  return *c;
}

size_t cache_capacity() {
  static const size_t cap = [] {
    const char *e = getenv("ARES_RTC_CACHE_ENTRIES");
    const long v = e ? atol(e) : 256;
    return static_cast<size_t>(v < 4 ? 4 : v);
  }();
  return cap;
}

bool rtc_async() {
  static const bool on = [] {
    const char *e = getenv("ARES_RTC_ASYNC");
    return !(e && e[0] == '0');
  }();
  return on;
}

// on-disk cache of code objects: ARES_RTC_CACHE_DIR, else $XDG_CACHE_HOME/aresdb_amd/rtc, else ~/.cache/aresdb_amd/rtc;
// "0" / "off" / "" switches it off
std::string disk_dir() {
  static const std::string dir = [] {
    std::string d;
    if (const char *e = getenv("ARES_RTC_CACHE_DIR")) {
      d = e;
      if (d == "0" || d == "off") d.clear();
      if (d.empty()) return d;
    } else if (const char *x = getenv("XDG_CACHE_HOME")) {
      d = std::string(x) + "/aresdb_amd/rtc";
    } else if (const char *h = getenv("HOME")) {
      d = std::string(h) + "/.cache/aresdb_amd/rtc";
    } else {
      return d;
    }
    std::string partial;  // mkdir -p
    for (size_t i = 0; i <= d.size(); i++)
      if (i == d.size() || (d[i] == '/' && i > 0)) {
        partial = d.substr(0, i);
        if (mkdir(partial.c_str(), 0700) != 0 && errno != EEXIST) return std::string();
      }
    return d;
  }();
  return dir;
}

constexpr uint64_t kDiskMagic = 0x3143545253455241ull;  // "ARESRTC1"
uint64_t fnv1a(const std::string &s, uint64_t h) {
  for (unsigned char c : s) {
    h ^= c;
    h *= 1099511628211ull;
  }
  return h;
}

// the options every kernel is compiled with (part of the on-disk cache key: a change here must not load old code)
constexpr const char *kRtcOptions[] = {"-O3", "-std=c++17", "-munsafe-fp-atomics"};
constexpr int kRtcOptionCount = 3;

std::string disk_name(const std::string &arch, const std::string &source) {
  int major = 0, minor = 0, runtime = 0;
  if (rtc_api().version) (void)rtc_api().version(&major, &minor);
  if (hipRuntimeGetVersion(&runtime) != hipSuccess) {  // carries the patch level the hiprtc pair lacks
    (void)hipGetLastError();
    runtime = 0;
  }
  std::string salt = arch + "|hiprtc " + std::to_string(major) + "." + std::to_string(minor) + "|runtime " + std::to_string(runtime) + "|";
  for (int k = 0; k < kRtcOptionCount; k++) salt += std::string(kRtcOptions[k]) + " ";
  salt += "|";
  char b[48];
  snprintf(b, sizeof(b), "%016llx%016llx.co", static_cast<unsigned long long>(fnv1a(source, fnv1a(salt, 14695981039346656037ull))),
           static_cast<unsigned long long>(fnv1a(source, fnv1a(salt, 0x9e3779b97f4a7c15ull))));
  return b;
}

std::string device_arch(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    (void)hipGetLastError();
    return "gfx950";
  }
  return prop.gcnArchName[0] ? std::string(prop.gcnArchName) : std::string("gfx950");
}

bool compile_source(const std::string &source, const std::string &arch, const char *entry, std::vector<char> &code) {
  const RtcApi &api = rtc_api();
  RtcProgram prog = nullptr;
  if (!api.ok || !api.create) return false;  // (no hiprtc in this process: callers check rtc_scan_available() first)
  if (api.create(&prog, source.c_str(), "hr_rtc.hip", 0, nullptr, nullptr) != 0) return false;
  const std::string archOpt = "--offload-arch=" + arch;
  const char *opts[1 + kRtcOptionCount] = {archOpt.c_str()};
  for (int k = 0; k < kRtcOptionCount; k++) opts[1 + k] = kRtcOptions[k];
  bool ok = false;
  if (api.compile(prog, 1 + kRtcOptionCount, opts) == 0) {
    size_t size = 0;
    if (api.codeSize(prog, &size) == 0 && size) {
      code.resize(size);
      ok = api.code(prog, code.data()) == 0;
    }
  } else {
    size_t n = 0;
    std::string log;
    if (api.logSize(prog, &n) == 0 && n) {
      log.resize(n);
      api.log(prog, &log[0]);
    }
    fprintf(stderr, "libalgorithm: hiprtc could not compile %s (generic kernel used): %s\n", entry, log.c_str());
  }
  api.destroy(&prog);
  return ok;
}

// ARES_RTC_TRACE=<file>: one line per kernel build — where its time went (diagnostics of cold starts)
void rtc_trace(const char *what, const std::string &entry, double ms, size_t bytes) {
  static const char *path = getenv("ARES_RTC_TRACE");
  if (!path || !path[0]) return;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (FILE *o = fopen(path, "a")) {
    const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    fprintf(o, "%.3f %s %s %.3f ms %zu bytes\n", now, what, entry.c_str(), ms, bytes);
    fclose(o);
  }
}
struct TraceClock {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// ---- building an entry -----------------------------------------------------------------------------------------
// A code object comes from the on-disk cache (read by the CALLER: tens of microseconds) or from hiprtc (seconds: on a
// background thread unless ARES_RTC_ASYNC=0 / `wait`).  hipModuleLoadData ALWAYS runs on a calling (query) thread, the
// first one that finds the code ready: 0.2-0.7 ms there.  On a background thread the same load was once seen to take
// 530 ms — queued behind a 4.3 GB hipMalloc of the query thread (the runtime serialises the two), which is what the
// "490 ms first query of a process that finds its kernels on disk" of round 3 really was: the DIRECT-mode workspace,
// sized for a region that mode does not write (hash_reduce_lds.hip, profiles/r4_experiments.md "cold start").
std::string disk_path(int device, const std::string &source) {
  const std::string dir = disk_dir();
  return dir.empty() ? std::string() : dir + "/" + disk_name(device_arch(device), source);
}

bool read_disk(const std::string &path, const std::string &entryName, std::vector<char> &code) {
  code.clear();
  if (path.empty()) return false;
  TraceClock tRead;
  std::ifstream in(path, std::ios::binary | std::ios::ate);  // file = {magic, code bytes, FNV-1a of the code} + code
  if (!in) return false;
  const std::streamsize n = in.tellg();
  uint64_t head[3] = {0, 0, 0};
  if (n > static_cast<std::streamsize>(sizeof(head))) {
    in.seekg(0);
    if (in.read(reinterpret_cast<char *>(head), sizeof(head)) && head[0] == kDiskMagic && head[1] == static_cast<uint64_t>(n) - sizeof(head)) {
      code.resize(static_cast<size_t>(head[1]));
      if (!in.read(code.data(), static_cast<std::streamsize>(code.size())) ||
          fnv1a(std::string(code.data(), code.size()), 14695981039346656037ull) != head[2])
        code.clear();
    }
  }
  if (code.empty()) {
    (void)unlink(path.c_str());  // truncated, corrupted or of another format
    return false;
  }
  rtc_trace("disk_read", entryName, tRead.ms(), code.size());
  return true;
}

void write_disk(const std::string &path, const std::vector<char> &code) {  // publish atomically: write aside, then rename
  if (path.empty() || code.empty()) return;
  const std::string tmp = path + ".tmp" + std::to_string(static_cast<long>(getpid())) + "." + std::to_string(cache().tick.load());
  std::ofstream out(tmp, std::ios::binary);
  const uint64_t head[3] = {kDiskMagic, static_cast<uint64_t>(code.size()), fnv1a(std::string(code.data(), code.size()), 14695981039346656037ull)};
  if (out && out.write(reinterpret_cast<const char *>(head), sizeof(head)) && out.write(code.data(), static_cast<std::streamsize>(code.size())) &&
      (out.close(), true)) {
    if (rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
  } else {
    (void)unlink(tmp.c_str());
  }
}

// caller holds e->m and has selected e->device: the entry's code becomes its kernel (or "no kernel": generic path)
void load_entry(RtcEntry &e, const std::string &entryName) {
  TraceClock tLoad;
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
  if (!e.code.empty() && hipModuleLoadData(&module, e.code.data()) == hipSuccess) {
    if (hipModuleGetFunction(&fn, module, entryName.c_str()) != hipSuccess) fn = nullptr;
  } else {
    module = nullptr;
  }
  (void)hipGetLastError();
  rtc_trace("module_load", entryName, tLoad.ms(), e.code.size());
  if (fn) {  // (a kernel with scratch memory pays a queue-wide scratch allocation at its first launch)
    int scratch = 0;
    if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fn) == hipSuccess)
      rtc_trace("scratch_bytes_per_lane", entryName, 0.0, static_cast<size_t>(scratch));
    (void)hipGetLastError();
  }
  if (!fn && e.fromDisk && !e.diskPath.empty()) (void)unlink(e.diskPath.c_str());  // a stale cache file
  if (fn && !e.fromDisk) write_disk(e.diskPath, e.code);
  e.code.clear();
  e.code.shrink_to_fit();
  e.module = module;
  e.fn = fn;
  e.codeReady = false;
  e.ready = true;
}

// hiprtc on this thread (the caller's, or a background thread): leaves the code in the entry for a query thread to load
void compile_entry(const std::shared_ptr<RtcEntry> &e, const std::string &source, const std::string &entryName) {
  RtcCache &c = cache();
  std::vector<char> code;
  TraceClock tCompile;
  const bool ok = compile_source(source, device_arch(e->device), entryName.c_str(), code);
  c.compiles++;
  rtc_trace("hiprtc_compile", entryName, tCompile.ms(), code.size());
  {
    std::lock_guard<std::mutex> lock(e->m);
    if (ok) {
      e->code.swap(code);
      e->codeReady = true;
    } else {
      e->ready = true;  // no kernel: the generic path
    }
  }
  e->cv.notify_all();
  if (c.pending.fetch_sub(1) == 1) {
    std::lock_guard<std::mutex> lock(c.idleMu);
    c.idleCv.notify_all();
  }
}

void wait_idle() {
  RtcCache &c = cache();
  std::unique_lock<std::mutex> lock(c.idleMu);
  c.idleCv.wait(lock, [&] { return c.pending.load() == 0; });
}

// The kernel of this source on `device`: the loaded kernel, or null while it is being built on a background
// thread (the caller uses the generic kernel this time), or — `wait` / ARES_RTC_ASYNC=0 — after building it.
RtcKernel compiled_kernel(int device, const std::string &source, const char *entry, bool wait) {
  if (source.empty()) return nullptr;
  RtcCache &c = cache();
  const std::string key = std::to_string(device) + "|" + entry + "|" + source;
  std::shared_ptr<RtcEntry> e;
  std::vector<std::shared_ptr<RtcEntry>> evicted;  // destroyed (device synchronised, module unloaded) outside the lock
  bool created = false;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.map.find(key);
    if (it != c.map.end()) {
      e = it->second;
    } else {
      while (c.map.size() >= cache_capacity()) {  // drop the least recently used kernel that is not being built
        auto victim = c.map.end();
        for (auto jt = c.map.begin(); jt != c.map.end(); ++jt) {
          bool ready;
          {  // (an entry whose code is here but that nobody asked for again has no module yet: dropping it is free)
            std::lock_guard<std::mutex> el(jt->second->m);
            ready = jt->second->ready || jt->second->codeReady;
          }
          if (ready && (victim == c.map.end() || jt->second->lastUse.load() < victim->second->lastUse.load())) victim = jt;
        }
        if (victim == c.map.end()) break;
        evicted.push_back(victim->second);
        c.map.erase(victim);
        c.evictions++;
      }
      e = std::make_shared<RtcEntry>();
      e->device = device;
      c.map.emplace(key, e);
      created = true;
      c.pending++;
      static const int registered = atexit([] { wait_idle(); });  // no build thread outlives the process's exit handlers
      (void)registered;
    }
    e->lastUse.store(++c.tick);
  }
  evicted.clear();
  if (created) {
    e->diskPath = disk_path(device, source);
    std::vector<char> code;
    if (read_disk(e->diskPath, entry, code)) {  // found on disk: loaded below, on this thread
      c.diskHits++;
      {
        std::lock_guard<std::mutex> lock(e->m);
        e->code.swap(code);
        e->fromDisk = true;
        e->codeReady = true;
      }
      e->cv.notify_all();  // (a second thread may already wait for this entry: `wait` / ARES_RTC_ASYNC=0)
      if (c.pending.fetch_sub(1) == 1) {
        std::lock_guard<std::mutex> idle(c.idleMu);
        c.idleCv.notify_all();
      }
    } else if (rtc_async() && !wait) {
      std::thread([e, source, name = std::string(entry)] { compile_entry(e, source, name); }).detach();
    } else {
      compile_entry(e, source, entry);
    }
  }
  std::unique_lock<std::mutex> lock(e->m);
  if (!e->ready && !e->codeReady) {
    if (rtc_async() && !wait) return nullptr;
    e->cv.wait(lock, [&] { return e->ready || e->codeReady; });
  }
  if (!e->ready) {  // the code is here: this (query) thread loads it
    (void)hipSetDevice(device);
    load_entry(*e, entry);
    e->cv.notify_all();
  }
  return e->fn ? e : nullptr;
}

uint32_t null_mask(const FusedPlanD &plan) {
  uint32_t m = 0;
  for (int c = 0; c < plan.numCols; c++)
    if (plan.cols[c].nulls) m |= 1u << c;
  return m;
}

void fill_scan_args(RtcArgs &args, const FusedPlanD &plan, uint32_t rowBase, int length, const hr::Workspace &ws) {
  memset(&args, 0, sizeof(args));
  for (int c = 0; c < plan.numCols; c++) {
    args.vals[c] = plan.cols[c].vals;
    args.nulls[c] = plan.cols[c].nulls;
    args.bitOff[c] = plan.cols[c].bitOff;
  }
  for (int k = 0; k < plan.numFilters && k < kFusedFilters; k++) args.k[const_slot_filter(k)] = compare_const(plan.filters[k].f);
  for (int d = 0; d < kFusedDims; d++) args.k[const_slot_dim(d)] = value_const(plan.dims[d].f);
  args.k[const_slot_measure()] = value_const(plan.measure.f);
  args.recB = ws.recB;
  args.countsB = ws.countsB;
  args.overflow = ws.outCount + 1;
  args.recA = ws.recA;
  args.cursorsA = ws.cursorsA;
  args.capA = ws.capA;
  args.rowBase = rowBase;
  args.length = length;
  args.capB = ws.capB;
}

void launch_scan(const RtcKernel &kernel, RtcArgs &args, int grid, int length, hipStream_t stream, const char *name) {
  size_t size = sizeof(args);
  void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  static uint64_t *phases = nullptr;
  if (phases_enabled()) {
    if (!phases) hip_check(hipMalloc(reinterpret_cast<void **>(&phases), sizeof(uint64_t) * 8 * hr::kMaxStreams), "hipMalloc");
    args.phases = phases;
  }
  {
    KernelTimer timer(name, stream);
    hip_check(hipModuleLaunchKernel(kernel->fn, static_cast<unsigned>(grid), 1, 1, hr::kThreads, 1, 1, 0, stream, nullptr, config),
              "hipModuleLaunchKernel");
  }
  if (phases_enabled() && args.phases) {  // diagnostics: core-clock cycles lane 0 of each workgroup spent per phase
    static int launches = 0;
    std::vector<uint64_t> h(static_cast<size_t>(8) * grid);
    hip_check(hipMemcpyAsync(h.data(), phases, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (++launches <= 3 || launches % 64 == 0) {
      double sum[7] = {0, 0, 0, 0, 0, 0, 0};
      for (int g = 0; g < grid; g++)
        for (int k = 0; k < 7; k++) sum[k] += static_cast<double>(h[static_cast<size_t>(8) * g + k]);
      const double n = grid * 1e3;
      fprintf(stderr, "%s phases (launch %d, %d workgroups, %d rows): kcycles per workgroup: eval+count %.1f, scan %.1f, scatter %.1f, lines %.1f, leftovers %.1f, drain %.1f, last lines %.1f\n",
              name, launches, grid, length, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, sum[5] / n, sum[6] / n);
    }
  }
}

}  // namespace

bool rtc_scan_available() { return rtc_api().ok; }

// number of workgroups (= private streams per partition) the specialised kernels want for `rows` rows
// ARES_SCAN_MIN_TILES (experiments): at least that many tiles per scanning workgroup — fewer, longer runs for the merge of a
// small batch, a longer scan.  Measured at 2 Mi-row batches (profiles/r5_experiments.md): 2 tiles (what 256 workgroups give)
// scan 0.024 + merge 0.051 ms, 4: 0.031 + 0.049, 8: 0.049 + 0.052 — the default (1: no constraint) stays.
int rtc_scan_grid(int64_t rows) {
  static const int64_t minTiles = [] {
    const char *e = getenv("ARES_SCAN_MIN_TILES");
    const long v = e ? atol(e) : 1;
    return static_cast<int64_t>(v < 1 ? 1 : v);
  }();
  const int64_t tiles = (rows + 4095) / 4096;
  int64_t grid = tiles < hr::kMaxStreams ? (tiles < 1 ? 1 : tiles) : hr::kMaxStreams;
  if (tiles > 1 && tiles / grid < minTiles) grid = (tiles + minTiles - 1) / minTiles;
  return static_cast<int>(grid < 1 ? 1 : grid);
}

// tiles per workgroup of the compact scan, 0 when a chunk would not fit the record's row field
int rtc_compact_chunk_tiles(int64_t rows, int partBits) {
  if (partBits < 3 || rows <= 0) return 0;
  const int64_t tiles = (rows + 4095) / 4096;
  const int64_t grid = rtc_scan_grid(rows);
  const int64_t chunk = (tiles + grid - 1) / grid;
  return chunk <= (1ll << (partBits - 3)) ? static_cast<int>(chunk) : 0;
}

// ---- lookups by plan shape, without writing the source -----------------------------------------------------
// The kernel cache is keyed by source text; writing a kernel's text (tens of kilobytes through a string stream) and
// comparing it costs tens of microseconds — twice per HashReduce call, 30 times per 1 B-row query, a third of a
// 2 Mi-row live batch's host time.  In front of it: a small map from the fields the generators read (expression
// shapes and constants, null mask, dimension count, partition bits, aggregate, generator switches) to the loaded
// kernel.  Only loaded kernels are entered; a miss takes the long way and is always right.
namespace {

struct FrontCache {
  std::mutex mu;
  std::unordered_map<std::string, RtcKernel> map;
};
FrontCache &front() {
  static FrontCache *f = new FrontCache;  // never destroyed: kernels are unloaded by the main cache's exit path
  return *f;
}

template <typename T>
void put(std::string &k, const T &v) {
  k.append(reinterpret_cast<const char *>(&v), sizeof(T));
}
void put_expr(std::string &k, const FusedExpr &e) {
  put(k, e.col); put(k, e.outKind); put(k, e.f.akind); put(k, e.f.arity); put(k, e.f.functor); put(k, e.f.I); put(k, e.f.rk);
  put(k, e.f.bkind); put(k, e.f.bbits); put(k, e.f.bok); put(k, e.f.divLike);
}
std::string shape_key(char tag, int device, const FusedPlanD &plan, int nd, int partBits, int flags, const AggSpec *a, const hr::Widen *w) {
  std::string k;
  k.reserve(512);
  k.push_back(tag);
  put(k, device); put(k, nd); put(k, partBits); put(k, flags);
  const unsigned opt = scan_opt();
  const uint32_t nm = null_mask(plan);
  put(k, opt); put(k, nm); put(k, plan.numCols); put(k, plan.numFilters);
  for (int c = 0; c < plan.numCols && c < kFusedCols; c++) { const int st = fused_col_step(plan, c); put(k, st); }
  for (int d = 0; d < nd && d < kFusedDims; d++) { const int wd = fused_dim_width(plan, d); put(k, wd); }
  for (int i = 0; i < plan.numFilters && i < kFusedFilters; i++) put_expr(k, plan.filters[i]);
  for (int d = 0; d < nd && d < kFusedDims; d++) put_expr(k, plan.dims[d]);
  put_expr(k, plan.measure);
  put(k, plan.measureDtype); put(k, plan.measureWidth); put(k, plan.identity);
  if (a) { put(k, a->vtype); put(k, a->op); put(k, a->width); put(k, a->identity); }
  if (w) { put(k, w->mode); put(k, w->rk); put(k, w->dtype); }
  return k;
}

template <typename Gen>
RtcKernel front_lookup(const std::string &key, int device, Gen &&source, const char *entry, bool wait) {
  FrontCache &f = front();
  {
    std::lock_guard<std::mutex> lock(f.mu);
    auto it = f.map.find(key);
    if (it != f.map.end()) return it->second;
  }
  RtcKernel k = compiled_kernel(device, source(), entry, wait);
  if (k) {
    std::lock_guard<std::mutex> lock(f.mu);
    if (f.map.size() >= 64) f.map.clear();  // holds references: bounded well below the main cache's capacity
    f.map.emplace(key, k);
  }
  return k;
}

}  // namespace

RtcKernel rtc_scan_lookup(int device, const FusedPlanD &plan, int nd, int partBits, bool compact, bool wait) {
  if (!rtc_api().ok) return nullptr;
  return front_lookup(shape_key('s', device, plan, nd, partBits, compact ? 1 : 0, nullptr, nullptr), device,
                      [&] { return generate(plan, nd, partBits, null_mask(plan), compact ? SCAN_COMPACT : SCAN_LINES16); }, "hr_scan_rtc", wait);
}

void rtc_scan_launch(const RtcKernel &kernel, const FusedPlanD &plan, uint32_t rowBase, int length, const hr::Workspace &ws,
                     hipStream_t stream) {
  RtcArgs args;
  fill_scan_args(args, plan, rowBase, length, ws);
  if (ws.lineRecords == static_cast<int>(hr::kCompactLineRecords)) args.chunkTiles = ws.chunkRows / 4096u;
  launch_scan(kernel, args, ws.streams, length, stream, "hr_scan_rtc");
}

RtcKernel rtc_sort_scan_lookup(int device, const FusedPlanD &plan, int nd, int partBits, bool wait) {
  if (!rtc_api().ok) return nullptr;
  return front_lookup(shape_key('o', device, plan, nd, partBits, 0, nullptr, nullptr), device,
                      [&] { return generate(plan, nd, partBits, null_mask(plan), SCAN_SORT64); }, "sr_scan_rtc", wait);
}

void rtc_sort_scan_launch(const RtcKernel &kernel, const FusedPlanD &plan, uint32_t rowBase, int length, const hr::Workspace &ws,
                          hipStream_t stream) {
  RtcArgs args;
  fill_scan_args(args, plan, rowBase, length, ws);
  launch_scan(kernel, args, ws.streams, length, stream, "sr_scan_rtc");
}

std::string rtc_sort_scan_source(const FusedPlanD &plan, int nd, int partBits) {
  return generate(plan, nd, partBits, null_mask(plan), SCAN_SORT64);
}

RtcKernel rtc_table_scan_lookup(int device, const FusedPlanD &plan, int nd, int partBits, const AggSpec &a, const hr::Widen &w, bool wait) {
  if (!rtc_api().ok) return nullptr;
  return front_lookup(shape_key('t', device, plan, nd, partBits, 0, &a, &w), device,
                      [&] { return generate(plan, nd, partBits, null_mask(plan), SCAN_TABLE, &a, &w); }, "hr_scan_rtc", wait);
}

void rtc_table_scan_launch(const RtcKernel &kernel, const FusedPlanD &plan, uint32_t rowBase, int length, const hr::Workspace &ws,
                           hipStream_t stream) {
  RtcArgs args;
  fill_scan_args(args, plan, rowBase, length, ws);
  const int64_t tiles = (static_cast<int64_t>(length) + 4095) / 4096;  // (one workgroup per tile up to a full device: a TABLE scan's
  const int grid = static_cast<int>(tiles < hr::kMaxStreams ? (tiles < 1 ? 1 : tiles) : hr::kMaxStreams);  // records do not grow with it)
  launch_scan(kernel, args, grid, length, stream, "hr_table_scan_rtc");
}

RtcKernel rtc_vector_scan_lookup(int device, int nd, int vw, int partBits, bool wait) {
  if (!rtc_api().ok) return nullptr;
  return compiled_kernel(device, generate_vector(nd, vw, partBits), "hr_scan_rtc", wait);
}

void rtc_vector_scan_launch(const RtcKernel &kernel, const uint8_t *dimValues, size_t capacity, const uint8_t *values, int nd, int vw,
                            uint32_t rowBase, int length, const hr::Workspace &ws, hipStream_t stream) {
  FusedPlanD plan;
  memset(&plan, 0, sizeof(plan));
  plan.numCols = nd + 1;
  for (int d = 0; d < nd; d++) {
    plan.cols[d].vals = reinterpret_cast<const uint32_t *>(dimValues + 4ull * d * capacity) + rowBase;
    plan.cols[d].nulls = dimValues + 4ull * nd * capacity + static_cast<size_t>(d) * capacity + rowBase;
  }
  plan.cols[nd].vals = reinterpret_cast<const uint32_t *>(values + static_cast<size_t>(vw) * rowBase);
  RtcArgs args;
  fill_scan_args(args, plan, rowBase, length, ws);
  launch_scan(kernel, args, ws.streams, length, stream, "hr_scan_rtc");
}

std::string rtc_vector_scan_source(int nd, int vw, int partBits) { return generate_vector(nd, vw, partBits); }

RtcKernel rtc_sort_vector_scan_lookup(int device, int nd, const int *widths, int partBits, bool wait) {
  if (!rtc_api().ok) return nullptr;
  return compiled_kernel(device, generate_vector(nd, 4, partBits, true, widths), "sr_scan_rtc", wait);
}
void rtc_sort_vector_scan_launch(const RtcKernel &kernel, const uint8_t *dimValues, size_t capacity, const uint8_t *values, int nd,
                                 const int *widths, uint32_t rowBase, int length, int totalPartBits, bool spread, const hr::Workspace &ws,
                                 hipStream_t stream) {
  FusedPlanD plan;
  memset(&plan, 0, sizeof(plan));
  plan.numCols = nd + 1;
  size_t valueBytes = 0, off = 0;
  for (int d = 0; d < nd; d++) valueBytes += static_cast<size_t>(widths ? widths[d] : 4);
  for (int d = 0; d < nd; d++) {
    const size_t w = static_cast<size_t>(widths ? widths[d] : 4);
    plan.cols[d].vals = reinterpret_cast<const uint32_t *>(dimValues + off * capacity + w * rowBase);
    plan.cols[d].nulls = dimValues + valueBytes * capacity + static_cast<size_t>(d) * capacity + rowBase;
    off += w;
  }
  plan.cols[nd].vals = reinterpret_cast<const uint32_t *>(values) + rowBase;  // (8-byte values: read at this stride and ignored)
  RtcArgs args;
  fill_scan_args(args, plan, rowBase, length, ws);
  args.chunkTiles = totalPartBits > 0 ? static_cast<uint32_t>(32 - totalPartBits) : 0u;  // (the partition expression's shift)
  args.pad = spread ? 1u : 0u;
  launch_scan(kernel, args, ws.streams, length, stream, "sr_vector_scan_rtc");
}
std::string rtc_sort_vector_scan_source(int nd, const int *widths, int partBits) { return generate_vector(nd, 4, partBits, true, widths); }

RtcKernel rtc_vector_merge_lookup(int device, int nd, int vw, int partBits, const AggSpec &a, bool wait) {
  if (!rtc_api().ok) return nullptr;
  FusedPlanD none;
  memset(&none, 0, sizeof(none));
  return compiled_kernel(device, generate_merge(none, nd, partBits, a, hr::Widen{0, 0, 0}, vw), "hr_merge_rtc", wait);
}

std::string rtc_vector_merge_source(int nd, int vw, int partBits, const AggSpec &a) {
  FusedPlanD none;
  memset(&none, 0, sizeof(none));
  return generate_merge(none, nd, partBits, a, hr::Widen{0, 0, 0}, vw);
}

RtcKernel rtc_merge_lookup(int device, const FusedPlanD &plan, int nd, int partBits, const AggSpec &a, const hr::Widen &w, bool compact,
                           bool wait, bool regionA, int image) {
  if (!rtc_api().ok) return nullptr;
  return front_lookup(shape_key('m', device, plan, nd, partBits, (compact ? 1 : 0) | (regionA ? 2 : 0) | (image << 2), &a, &w), device,
                      [&] { return generate_merge(plan, nd, partBits, a, w, 0, compact, regionA, image); }, "hr_merge_rtc", wait);
}

void rtc_merge_launch(const RtcKernel &kernel, const FusedPlanD &plan, const uint8_t *prevDims, size_t prevCapacity, const uint8_t *prevValues,
                      uint32_t prevSize, uint8_t *dimOut, size_t outCapacity, uint8_t *outValues, const hr::Workspace &ws,
                      hipStream_t stream, const RtcImageArgs *image, uint32_t *hostOut) {
  RtcMergeArgs args;
  memset(&args, 0, sizeof(args));
  for (int c = 0; c < plan.numCols; c++) {
    args.vals[c] = plan.cols[c].vals;
    args.nulls[c] = plan.cols[c].nulls;
    args.bitOff[c] = plan.cols[c].bitOff;
  }
  for (int d = 0; d < kFusedDims; d++) args.k[const_slot_dim(d)] = value_const(plan.dims[d].f);
  args.recB = ws.recB;
  args.countsB = ws.countsB;
  args.prevRanges = ws.prevRanges;
  args.prevDims = prevDims;
  args.prevValues = prevValues;
  args.dimOut = dimOut;
  args.outValues = outValues;
  args.outCount = ws.outCount;
  args.outRanges = ws.outRanges;
  args.prevCapacity = prevCapacity;
  args.outCapacity = outCapacity;
  args.capB = ws.capB;
  args.recA = ws.recA;
  args.cursorsA = ws.cursorsA;
  args.capA = ws.capA;
  args.streams = static_cast<uint32_t>(ws.streams);
  args.prevSize = prevSize;
  args.chunkRows = ws.chunkRows;
  if (image) {
    args.imgIn = reinterpret_cast<const uint4 *>(image->in);
    args.imgOut = reinterpret_cast<uint4 *>(image->out);
    args.imgInCount = image->inCount;
    args.imgOutCount = image->outCount;
    args.knownOut = image->knownOut;
  }
  args.hostOut = hostOut;
  size_t size = sizeof(args);
  void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  static uint64_t *phases = nullptr;
  if (phases_enabled()) {
    if (!phases) hip_check(hipMalloc(reinterpret_cast<void **>(&phases), sizeof(uint64_t) * 8 * hr::kMaxPartitions), "hipMalloc");
    hip_check(hipMemsetAsync(phases, 0, sizeof(uint64_t) * 8 * hr::kMaxPartitions, stream), "hipMemsetAsync");
    args.phases = phases;
  }
  {
    KernelTimer timer("hr_merge_rtc", stream);
    hip_check(hipModuleLaunchKernel(kernel->fn, 1u << ws.partBits, 1, 1, hr::kThreads, 1, 1, 0, stream, nullptr, config),
              "hipModuleLaunchKernel");
  }
  if (phases_enabled()) {  // diagnostics: where a partition's time goes (100 MHz constant clock)
    static int launches = 0;
    const int np = 1 << ws.partBits;
    std::vector<uint64_t> h(static_cast<size_t>(8) * np);
    hip_check(hipMemcpyAsync(h.data(), phases, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (++launches <= 3 || launches % 64 == 0) {
      uint64_t first = ~0ull, last = 0;
      double sum[5] = {0, 0, 0, 0, 0}, whole = 0;
      for (int p = 0; p < np; p++) {
        const uint64_t *t = &h[static_cast<size_t>(8) * p];
        if (t[0] < first) first = t[0];
        if (t[5] > last) last = t[5];
        for (int k = 0; k < 5; k++) sum[k] += static_cast<double>(t[k + 1] - t[k]) * 0.01;
        whole += static_cast<double>(t[5] - t[0]) * 0.01;
      }
      fprintf(stderr, "hr_merge_rtc phases (launch %d, %d partitions, prev %u): span %.1f us; per partition avg %.1f us = init %.1f + prev %.1f + records %.1f + count %.1f + emit %.1f\n",
              launches, np, prevSize, static_cast<double>(last - first) * 0.01, whole / np, sum[0] / np, sum[1] / np, sum[2] / np, sum[3] / np, sum[4] / np);
    }
  }
}

std::string rtc_merge_source(const FusedPlanD &plan, int nd, int partBits, const AggSpec &a, const hr::Widen &w, bool compact, bool regionA,
                             int image) {
  return generate_merge(plan, nd, partBits, a, w, 0, compact, regionA, image);
}

// source text of the kernel a plan would get (tests / tools; empty = unsupported shape)
std::string rtc_scan_source(const FusedPlanD &plan, int nd, int partBits, bool compact) {
  return generate(plan, nd, partBits, null_mask(plan), compact ? SCAN_COMPACT : SCAN_LINES16);
}

std::string rtc_table_scan_source(const FusedPlanD &plan, int nd, int partBits, const AggSpec &a, const hr::Widen &w) {
  return generate(plan, nd, partBits, null_mask(plan), SCAN_TABLE, &a, &w);
}

}  // namespace ares

// Exported (include/ares_extensions.h): blocks until no kernel is being compiled in the background; returns the
// number of kernels the cache holds.  counters (may be null): [0] hiprtc compilations, [1] code objects found in
// the on-disk cache, [2] kernels dropped from the in-memory cache.
extern "C" size_t AresRtcWait(long *counters) {
  ares::wait_idle();
  ares::RtcCache &c = ares::cache();
  if (counters) {
    counters[0] = c.compiles.load();
    counters[1] = c.diskHits.load();
    counters[2] = c.evictions.load();
  }
  std::lock_guard<std::mutex> lock(c.mu);
  return c.map.size();
}
