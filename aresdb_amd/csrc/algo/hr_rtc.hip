// Per-plan specialised scan kernel of the fused HashReduce, compiled at run time with hiprtc.
//
// The generic hr_fused_scan_kernel (hr_kernels.hpp) interprets the plan: operation codes, kinds and
// constants arrive as kernel arguments and every expression is dispatched per quad.  That costs
// ~210 VALU + ~80 SALU instructions per row, 266 KB of code and several hundred spilled SGPRs — the
// kernel is bound by instruction issue and instruction fetch, not by HBM.  Here the host writes the
// plan out as straight-line code (constants as literals, so the compiler strength-reduces the time
// bucket division itself), hiprtc compiles it for gfx950 (~0.5 s, cached per plan signature and
// device for the life of the process) and the result is launched through the module API.
//
// The generated kernel covers DIRECT mode only (rows -> 12-byte records in the workgroup's private
// streams, hr_kernels.hpp): the host uses it when the query is known to have more groups than an LDS
// table holds; otherwise, and for every plan outside the supported shapes, the generic kernel runs.
// Wavefronts work independently (no barrier between prologue and epilogue): each walks its own
// 256-row tiles with two register buffers, so a tile's loads are in flight while the previous tile
// is evaluated, hashed and scattered.
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

#include <map>
#include <mutex>
#include <sstream>
#include <string>
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

// value expression of one element: writes `r` (result bits) given `v` (stored bits) and `okb` (0/1);
// returns false when the shape is outside the fast paths of eval_quad
bool gen_value(const FastOperands &f, std::ostringstream &o, const char *v, const char *okb, const char *r) {
  if (!col_kind(f.akind)) return false;
  const bool intKinds = f.akind != K_F32 && f.I != K_F32 && f.akind != K_BOOL;
  if (f.arity == 1) {
    if (!(f.akind == f.I || intKinds)) return false;
    o << "      " << r << " = " << v << ";\n";  // a null bare column keeps its stored bits (functor.hpp:345-351)
    return true;
  }
  if (f.arity != 2 || !intKinds || !int_kind(f.I) || !int_kind(f.bkind) || !f.bok) return false;
  const uint32_t y = f.bbits;  // cvt32 between the integer kinds keeps the bits
  if (f.divLike) {
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
    o << "      " << r << " = " << okb << " ? (" << v << (f.functor == Plus ? " + " : f.functor == Minus ? " - " : " * ") << hex(y)
      << ") : 0u;\n";
    return true;
  }
  return false;
}

// keep bit of one element for one filter
bool gen_compare(const FastOperands &f, std::ostringstream &o, const char *v, const char *okb, const char *keep) {
  if (!col_kind(f.akind) || f.arity != 2) return false;
  const bool sameBits = f.akind == f.I || (f.akind != K_F32 && f.I != K_F32 && f.akind != K_BOOL);
  if (!sameBits || !f.bok) return false;
  if (f.functor < Equal || f.functor > GreaterThanOrEqual) return false;
  if (!(f.I == K_F32 || f.I == K_I32 || f.I == K_U32)) return false;
  const uint32_t y = host_cvt32(f.bbits, f.bkind, f.I);
  const char *op = f.functor == Equal ? "==" : f.functor == NotEqual ? "!=" : f.functor == LessThan ? "<"
                   : f.functor == LessThanOrEqual ? "<=" : f.functor == GreaterThan ? ">" : ">=";
  if (f.I == K_F32) o << "      " << keep << " &= (" << okb << " && (__uint_as_float(" << v << ") " << op << " __uint_as_float(" << hex(y) << "))) ? 1u : 0u;\n";
  else if (f.I == K_I32) o << "      " << keep << " &= (" << okb << " && ((i32)" << v << " " << op << " (i32)" << hex(y) << ")) ? 1u : 0u;\n";
  else o << "      " << keep << " &= (" << okb << " && (" << v << " " << op << " " << hex(y) << ")) ? 1u : 0u;\n";
  return true;
}

bool plain_store(int rk, int outKind) { return rk == outKind || (rk != K_F32 && outKind != K_F32 && rk != K_BOOL); }

struct RtcArgs {  // mirrors `struct Args` of the generated source (pointers first, then 4-byte fields)
  const uint32_t *vals[kFusedCols];
  const uint8_t *nulls[kFusedCols];
  uint32_t *recB;
  uint32_t *countsB;
  uint32_t *overflow;
  uint32_t bitOff[kFusedCols];
  uint32_t rowBase;
  int length;
  uint32_t capB;
  uint32_t pad;
};

// the whole kernel source for `plan`; empty when the plan is outside the supported shapes
std::string generate(const FusedPlanD &plan, int nd, int partBits, uint32_t nullMask) {
  if (nd < 1 || nd > kFusedDims || plan.numCols > kFusedCols) return "";
  std::ostringstream o;
  const int nc = plan.numCols;
  o << "typedef unsigned int u32; typedef unsigned long long u64; typedef unsigned char u8; typedef unsigned short u16; typedef int i32;\n"
       "struct __attribute__((packed, aligned(1))) PU32x4 { u32 v[4]; };\n"
       "struct __attribute__((packed, aligned(1))) PU16 { u16 v; };\n"
       "struct __attribute__((packed, aligned(4))) Rec3 { u32 row, hash, val; };\n"
       "struct Args { const u32 *vals[" << kFusedCols << "]; const u8 *nulls[" << kFusedCols << "]; u32 *recB; u32 *countsB; u32 *overflow;\n"
       "              u32 bitOff[" << kFusedCols << "]; u32 rowBase; int length; u32 capB; u32 pad; };\n"
       "#define NC " << nc << "\n#define ND " << nd << "\n#define PB " << partBits << "\n#define NP " << (1 << partBits) << "\n"
       "__device__ __forceinline__ u32 rotl(u32 x, int r) { return (x << r) | (x >> (32 - r)); }\n"
       "__device__ __forceinline__ u32 mix(u32 h, u32 k) { k *= 0xcc9e2d51u; k = rotl(k, 15) * 0x1b873593u; h ^= k; return rotl(h, 13) * 5u + 0xe6546b64u; }\n"
       "struct Raw { u32 v[NC][4]; u32 win[NC]; };\n";
  // ---- loads of a full quad (all four rows exist) ----
  o << "__device__ __forceinline__ void load_full(Raw &r, const Args &a, u32 i0) {\n";
  for (int c = 0; c < nc; c++) {
    o << "  { const PU32x4 t = *reinterpret_cast<const PU32x4 *>(a.vals[" << c << "] + i0); r.v[" << c << "][0] = t.v[0]; r.v[" << c
      << "][1] = t.v[1]; r.v[" << c << "][2] = t.v[2]; r.v[" << c << "][3] = t.v[3]; }\n";
    if (nullMask & (1u << c))
      o << "  r.win[" << c << "] = reinterpret_cast<const PU16 *>(a.nulls[" << c << "] + ((i0 + a.bitOff[" << c << "]) >> 3))->v;\n";
    else
      o << "  r.win[" << c << "] = 0xFFFFu;\n";
  }
  o << "}\n";
  // ---- guarded loads of the shard's last, partial tile ----
  o << "__device__ __forceinline__ void load_tail(Raw &r, const Args &a, u32 i0) {\n";
  for (int c = 0; c < nc; c++) {
    o << "  for (int j = 0; j < 4; j++) r.v[" << c << "][j] = (int)(i0 + j) < a.length ? a.vals[" << c << "][i0 + j] : 0u;\n";
    if (nullMask & (1u << c))
      o << "  r.win[" << c << "] = (int)i0 < a.length ? (u32)reinterpret_cast<const PU16 *>(a.nulls[" << c << "] + ((i0 + a.bitOff[" << c
        << "]) >> 3))->v : 0u;\n";
    else
      o << "  r.win[" << c << "] = 0xFFFFu;\n";
  }
  o << "}\n";
  // ---- evaluate + hash + scatter one quad ----
  o << "__device__ __forceinline__ void process(const Raw &r, const Args &a, u32 i0, u32 *sCursor, u32 *myB) {\n"
       "  u32 okc[NC];\n";
  for (int c = 0; c < nc; c++) {
    if (nullMask & (1u << c)) o << "  okc[" << c << "] = (r.win[" << c << "] >> ((i0 + a.bitOff[" << c << "]) & 7u)) & 0xFu;\n";
    else o << "  okc[" << c << "] = 0xFu;\n";
  }
  o << "  u32 hh[4], cv[4], alive[4];\n"
       "#pragma unroll\n"
       "  for (int j = 0; j < 4; j++) {\n"
       "    u32 keep = (int)(i0 + j) < a.length ? 1u : 0u;\n";
  for (int k = 0; k < plan.numFilters; k++) {
    const FusedExpr &e = plan.filters[k];
    o << "    {\n      const u32 v = r.v[" << e.col << "][j]; const u32 okb = (okc[" << e.col << "] >> j) & 1u;\n";
    if (!gen_compare(e.f, o, "v", "okb", "keep")) return "";
    o << "    }\n";
  }
  o << "    alive[j] = keep;\n    u32 h = 0u, okbytes = 0u;\n";
  for (int d = 0; d < nd; d++) {
    const FusedExpr &e = plan.dims[d];
    if (e.col != d || !plain_store(e.f.rk, e.outKind)) return "";
    o << "    {\n      const u32 v = r.v[" << d << "][j]; const u32 okb = (okc[" << d << "] >> j) & 1u; u32 x;\n";
    if (!gen_value(e.f, o, "v", "okb", "x")) return "";
    o << "      h = mix(h, x); okbytes |= okb << " << 8 * d << ";\n    }\n";
  }
  // Murmur32Stream (dim_layout.hpp): the validity bytes are one more block when there are four of them,
  // otherwise the tail
  if (nd == 4) o << "    h = mix(h, okbytes);\n";
  else o << "    { u32 k = okbytes * 0xcc9e2d51u; k = rotl(k, 15) * 0x1b873593u; h ^= k; }\n";
  o << "    h ^= " << 5 * nd << "u; h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;\n"
       "    hh[j] = h;\n";
  {  // measure: fused_carry
    const FusedExpr &e = plan.measure;
    if (e.col != nd) return "";
    o << "    {\n      const u32 v = r.v[" << nd << "][j]; const u32 okb = (okc[" << nd << "] >> j) & 1u; u32 x;\n";
    if (!gen_value(e.f, o, "v", "okb", "x")) return "";
    if (plan.measureWidth == 8) {
      if (plan.identity != 0) return "";
      o << "      cv[j] = okb ? x : 0u;\n";
    } else {
      const int target = plan.measureDtype == Int32 ? K_I32 : plan.measureDtype == Uint32 ? K_U32 : K_F32;
      if (!plain_store(e.f.rk, target)) return "";
      o << "      cv[j] = okb ? x : " << hex(static_cast<uint32_t>(plan.identity)) << ";\n";
    }
    o << "    }\n  }\n";
  }
  o << "  u32 rank[4];\n"
       "#pragma unroll\n"
       "  for (int j = 0; j < 4; j++) {\n"
       "    rank[j] = a.capB;\n"
       "    if (alive[j]) rank[j] = __hip_atomic_fetch_add(&sCursor[PB ? hh[j] >> (32 - (PB ? PB : 1)) : 0u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
       "  }\n"
       "#pragma unroll\n"
       "  for (int j = 0; j < 4; j++) {\n"
       "    if (rank[j] < a.capB) {\n"
       "      const u32 p = PB ? hh[j] >> (32 - (PB ? PB : 1)) : 0u;\n"
       "      Rec3 rec; rec.row = a.rowBase + i0 + j; rec.hash = hh[j]; rec.val = cv[j];\n"
       "      *reinterpret_cast<Rec3 *>(myB + (p * a.capB + rank[j]) * 3u) = rec;\n"
       "    }\n"
       "  }\n"
       "}\n";
  // ---- the kernel: every wavefront walks its own 256-row tiles, two register buffers ----
  o << "extern \"C\" __global__ void __launch_bounds__(1024) hr_scan_rtc(Args a) {\n"
       "  __shared__ u32 sCursor[NP];\n"
       "  for (int p = threadIdx.x; p < NP; p += 1024) sCursor[p] = 0u;\n"
       "  __syncthreads();\n"
       "  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;\n"
       "  u32 *myB = a.recB + (u64)blockIdx.x * NP * a.capB * 3u;\n"
       "  const u32 fullTiles = (u32)a.length >> 8;\n"
       "  const u32 stride = gridDim.x * 16u;\n"
       "  u32 tile = blockIdx.x * 16u + wave;\n"
       "  if (tile < fullTiles) {\n"
       "    const u32 last = fullTiles - 1u;\n"
       "    Raw A, B;\n"
       "    load_full(A, a, tile * 256u + lane * 4u);\n"
       "    for (;;) {\n"
       "      const u32 t2 = tile + stride;\n"
       "      load_full(B, a, (t2 < last ? t2 : last) * 256u + lane * 4u);\n"  // unconditional: the compiler counts the loads
       "      process(A, a, tile * 256u + lane * 4u, sCursor, myB);\n"
       "      if (t2 >= fullTiles) break;\n"
       "      const u32 t3 = t2 + stride;\n"
       "      load_full(A, a, (t3 < last ? t3 : last) * 256u + lane * 4u);\n"
       "      process(B, a, t2 * 256u + lane * 4u, sCursor, myB);\n"
       "      if (t3 >= fullTiles) break;\n"
       "      tile = t3;\n"
       "    }\n"
       "  }\n"
       "  if (((u32)a.length & 255u) && (fullTiles % stride) == blockIdx.x * 16u + wave) {\n"  // the partial tile
       "    Raw T;\n"
       "    load_tail(T, a, fullTiles * 256u + lane * 4u);\n"
       "    process(T, a, fullTiles * 256u + lane * 4u, sCursor, myB);\n"
       "  }\n"
       "  __syncthreads();\n"
       "  for (int p = threadIdx.x; p < NP; p += 1024) {\n"
       "    u32 cnt = sCursor[p];\n"
       "    if (cnt > a.capB) { *a.overflow = 1u; cnt = a.capB; }\n"
       "    a.countsB[(u64)blockIdx.x * NP + p] = cnt;\n"
       "  }\n"
       "}\n";
  return o.str();
}

struct Compiled {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;  // nullptr = this signature does not compile / is unsupported: generic kernel
};
std::mutex g_rtcMutex;
std::map<std::pair<int, std::string>, Compiled> g_rtcCache;

// compiles (or finds) the kernel of this source on the current device
hipFunction_t compiled_kernel(int device, const std::string &source) {
  std::lock_guard<std::mutex> lock(g_rtcMutex);
  auto it = g_rtcCache.find({device, source});
  if (it != g_rtcCache.end()) return it->second.fn;
  Compiled c;
  const RtcApi &api = rtc_api();
  RtcProgram prog = nullptr;
  if (api.create(&prog, source.c_str(), "hr_scan_rtc.hip", 0, nullptr, nullptr) == 0) {
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics"};
    const int rc = api.compile(prog, 4, opts);
    if (rc == 0) {
      size_t size = 0;
      if (api.codeSize(prog, &size) == 0 && size) {
        std::vector<char> code(size);
        if (api.code(prog, code.data()) == 0 && hipModuleLoadData(&c.module, code.data()) == hipSuccess) {
          if (hipModuleGetFunction(&c.fn, c.module, "hr_scan_rtc") != hipSuccess) c.fn = nullptr;
        }
        (void)hipGetLastError();
      }
    } else {
      size_t n = 0;
      std::string log;
      if (api.logSize(prog, &n) == 0 && n) {
        log.resize(n);
        api.log(prog, &log[0]);
      }
      fprintf(stderr, "libalgorithm: hiprtc could not compile a specialised scan kernel (generic kernel used): %s\n", log.c_str());
    }
    api.destroy(&prog);
  }
  g_rtcCache[{device, source}] = c;
  return c.fn;
}

}  // namespace

bool rtc_scan_available() { return rtc_api().ok; }

bool rtc_scan_launch(int device, const FusedPlanD &plan, int nd, uint32_t rowBase, int length, const hr::Workspace &ws,
                     hipStream_t stream) {
  if (!rtc_api().ok || length <= 0 || ws.streams <= 0) return false;
  uint32_t nullMask = 0;
  for (int c = 0; c < plan.numCols; c++)
    if (plan.cols[c].nulls) nullMask |= 1u << c;
  const std::string source = generate(plan, nd, ws.partBits, nullMask);
  if (source.empty()) return false;
  hipFunction_t fn = compiled_kernel(device, source);
  if (!fn) return false;
  RtcArgs args;
  memset(&args, 0, sizeof(args));
  for (int c = 0; c < plan.numCols; c++) {
    args.vals[c] = plan.cols[c].vals;
    args.nulls[c] = plan.cols[c].nulls;
    args.bitOff[c] = plan.cols[c].bitOff;
  }
  args.recB = ws.recB;
  args.countsB = ws.countsB;
  args.overflow = ws.outCount + 1;
  args.rowBase = rowBase;
  args.length = length;
  args.capB = ws.capB;
  size_t size = sizeof(args);
  void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  KernelTimer timer("hr_scan_rtc", stream);
  hip_check(hipModuleLaunchKernel(fn, static_cast<unsigned>(ws.streams), 1, 1, hr::kThreads, 1, 1, 0, stream, nullptr, config),
            "hipModuleLaunchKernel");
  return true;
}

// number of workgroups (= private streams per partition) the specialised kernel wants for `rows` rows
int rtc_scan_grid(int64_t rows) {
  const int64_t waveTiles = (rows + 255) / 256;
  const int64_t groups = (waveTiles + 15) / 16;
  return static_cast<int>(groups < hr::kMaxStreams ? (groups < 1 ? 1 : groups) : hr::kMaxStreams);
}

// source text of the kernel a plan would get (tests / tools; empty = unsupported shape)
std::string rtc_scan_source(const FusedPlanD &plan, int nd, int partBits) {
  uint32_t nullMask = 0;
  for (int c = 0; c < plan.numCols; c++)
    if (plan.cols[c].nulls) nullMask |= 1u << c;
  return generate(plan, nd, partBits, nullMask);
}

}  // namespace ares
