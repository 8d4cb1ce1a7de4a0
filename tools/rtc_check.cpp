// Builds the C3 plan by hand, prints the hiprtc source the library generates for it, compiles it for
// gfx950 (no GPU needed) and writes the code object next to it:  tools/bin/rtc_check [out-prefix]
// Links against aresdb_amd/lib/libalgorithm.so (ares::rtc_scan_source is an ordinary exported symbol).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "hash_reduce_lds.hpp"
#include "hr_kernels.hpp"
#include "hr_rtc.hpp"

using namespace ares;

static FastOperands col(int akind) {
  FastOperands f;
  memset(&f, 0, sizeof(f));
  f.akind = akind; f.arity = 1; f.functor = Noop; f.I = akind; f.rk = akind; f.bkind = akind;
  return f;
}

int main(int argc, char **argv) {
  FusedPlanD p;
  memset(&p, 0, sizeof(p));
  p.numCols = 5;
  for (int c = 0; c < 5; c++) { p.cols[c].vals = reinterpret_cast<const uint32_t *>(0x1000); p.cols[c].nulls = reinterpret_cast<const uint8_t *>(0x2000); }
  p.numFilters = 1;
  p.filters[0].f = col(K_U32); p.filters[0].f.arity = 2; p.filters[0].f.functor = LessThan; p.filters[0].f.bkind = K_I32;
  p.filters[0].f.bbits = 90; p.filters[0].f.bok = 1; p.filters[0].col = 1; p.filters[0].outKind = K_BOOL;
  p.dims[0].f = col(K_U32); p.dims[0].f.arity = 2; p.dims[0].f.functor = Floor; p.dims[0].f.bkind = K_I32; p.dims[0].f.bbits = 3600;
  p.dims[0].f.bok = 1; p.dims[0].f.divLike = 1; p.dims[0].col = 0; p.dims[0].outKind = K_U32;
  for (int d = 1; d < 4; d++) { p.dims[d].f = col(K_U32); p.dims[d].col = d; p.dims[d].outKind = K_U32; }
  p.measure.f = col(K_F32); p.measure.col = 4; p.measure.outKind = K_F32;
  p.measureDtype = Float64; p.measureWidth = 8; p.identity = 0;
  const std::string src = rtc_scan_source(p, 4, 9);
  if (src.empty()) { puts("unsupported plan"); return 2; }
  const std::string prefix = argc > 1 ? argv[1] : "/tmp/hr_scan_rtc";
  std::ofstream(prefix + ".hip") << src;
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "hr_scan_rtc.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return 3;
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics"};
  const hiprtcResult rc = hiprtcCompileProgram(prog, 4, opts);
  size_t n = 0; hiprtcGetProgramLogSize(prog, &n);
  std::string log(n, 0); if (n) hiprtcGetProgramLog(prog, &log[0]);
  printf("compile rc %d\n%s\n", static_cast<int>(rc), log.c_str());
  if (rc != HIPRTC_SUCCESS) return 4;
  size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
  std::vector<char> code(cs); hiprtcGetCode(prog, code.data());
  std::ofstream(prefix + ".co", std::ios::binary).write(code.data(), static_cast<std::streamsize>(cs));
  printf("code object %zu bytes -> %s.co\n", cs, prefix.c_str());
  // the specialised merge of the same plan
  AggSpec agg = make_agg_spec(AGGR_SUM_FLOAT, 8);
  hr::Widen w{1, K_F32, Float64};
  const std::string msrc = rtc_merge_source(p, 4, 9, agg, w);
  if (msrc.empty()) { puts("merge: unsupported plan"); return 5; }
  std::ofstream(prefix + "_merge.hip") << msrc;
  hiprtcProgram mp;
  if (hiprtcCreateProgram(&mp, msrc.c_str(), "hr_merge_rtc.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return 6;
  const hiprtcResult mrc = hiprtcCompileProgram(mp, 4, opts);
  n = 0; hiprtcGetProgramLogSize(mp, &n);
  std::string mlog(n, 0); if (n) hiprtcGetProgramLog(mp, &mlog[0]);
  printf("merge compile rc %d\n%s\n", static_cast<int>(mrc), mlog.c_str());
  if (mrc != HIPRTC_SUCCESS) return 7;
  hiprtcGetCodeSize(mp, &cs);
  std::vector<char> mcode(cs); hiprtcGetCode(mp, mcode.data());
  std::ofstream(prefix + "_merge.co", std::ios::binary).write(mcode.data(), static_cast<std::streamsize>(cs));
  printf("merge code object %zu bytes\n", cs);
  // the vector-sourced scan (HashReduce on materialised dimension / measure vectors)
  for (int vw = 4; vw <= 8; vw += 4)
    for (int nd = 1; nd <= 4; nd += 3) {
      const std::string vsrc = rtc_vector_scan_source(nd, vw, 9);
      if (vsrc.empty()) { puts("vector scan: unsupported"); return 8; }
      if (nd == 4 && vw == 8) std::ofstream(prefix + "_vector.hip") << vsrc;
      hiprtcProgram vp;
      if (hiprtcCreateProgram(&vp, vsrc.c_str(), "hr_vscan_rtc.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return 9;
      const hiprtcResult vrc = hiprtcCompileProgram(vp, 4, opts);
      n = 0; hiprtcGetProgramLogSize(vp, &n);
      std::string vlog(n, 0); if (n) hiprtcGetProgramLog(vp, &vlog[0]);
      printf("vector scan nd %d vw %d compile rc %d\n%s\n", nd, vw, static_cast<int>(vrc), vlog.c_str());
      if (vrc != HIPRTC_SUCCESS) return 10;
      {
        const AggSpec va = make_agg_spec(vw == 8 ? AGGR_SUM_FLOAT : AGGR_SUM_UNSIGNED, vw);
        const std::string vm = rtc_vector_merge_source(nd, vw, 9, va);
        if (vm.empty()) { puts("vector merge: unsupported"); return 11; }
        if (nd == 4 && vw == 8) std::ofstream(prefix + "_vmerge.hip") << vm;
        hiprtcProgram vq;
        if (hiprtcCreateProgram(&vq, vm.c_str(), "hr_vmerge_rtc.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return 12;
        const hiprtcResult qrc = hiprtcCompileProgram(vq, 4, opts);
        n = 0; hiprtcGetProgramLogSize(vq, &n);
        std::string qlog(n, 0); if (n) hiprtcGetProgramLog(vq, &qlog[0]);
        printf("vector merge nd %d vw %d compile rc %d\n%s\n", nd, vw, static_cast<int>(qrc), qlog.c_str());
        if (qrc != HIPRTC_SUCCESS) return 13;
        if (nd == 4 && vw == 8) {
          hiprtcGetCodeSize(vq, &cs);
          std::vector<char> qcode(cs); hiprtcGetCode(vq, qcode.data());
          std::ofstream(prefix + "_vmerge.co", std::ios::binary).write(qcode.data(), static_cast<std::streamsize>(cs));
        }
      }
      if (nd == 4 && vw == 8) {
        hiprtcGetCodeSize(vp, &cs);
        std::vector<char> vcode(cs); hiprtcGetCode(vp, vcode.data());
        std::ofstream(prefix + "_vector.co", std::ios::binary).write(vcode.data(), static_cast<std::streamsize>(cs));
      }
    }
  return 0;
}
