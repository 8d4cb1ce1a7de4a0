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
  const std::string prefix = argc > 1 ? argv[1] : "/tmp/hr_scan_rtc";
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics"};
  size_t n = 0, cs = 0;
  auto build = [&](const std::string &src, const std::string &tag, const char *what) -> int {
    if (src.empty()) { printf("%s: unsupported plan\n", what); return 2; }
    std::ofstream(prefix + tag + ".hip") << src;
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "hr_rtc.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return 3;
    const hiprtcResult rc = hiprtcCompileProgram(prog, 4, opts);
    size_t ln = 0; hiprtcGetProgramLogSize(prog, &ln);
    std::string log(ln, 0); if (ln) hiprtcGetProgramLog(prog, &log[0]);
    printf("%s compile rc %d\n%s\n", what, static_cast<int>(rc), log.c_str());
    if (rc != HIPRTC_SUCCESS) return 4;
    size_t sz = 0; hiprtcGetCodeSize(prog, &sz);
    std::vector<char> code(sz); hiprtcGetCode(prog, code.data());
    std::ofstream(prefix + tag + ".co", std::ios::binary).write(code.data(), static_cast<std::streamsize>(sz));
    printf("%s code object %zu bytes -> %s%s.co\n", what, sz, prefix.c_str(), tag.c_str());
    return 0;
  };
  AggSpec agg = make_agg_spec(AGGR_SUM_FLOAT, 8);
  hr::Widen w{1, K_F32, Float64};
  // the same shape with another comparison constant is the same source text (constants are kernel arguments)
  {
    FusedPlanD p2 = p;
    p2.filters[0].f.bbits = 17;
    if (rtc_scan_source(p2, 4, 9, true) != rtc_scan_source(p, 4, 9, true)) { puts("comparison constants leak into the source"); return 20; }
    p2.dims[0].f.bbits = 60;
    if (rtc_scan_source(p2, 4, 9, true) == rtc_scan_source(p, 4, 9, true)) { puts("divisors must be literals"); return 21; }
  }
  if (int rc = build(rtc_scan_source(p, 4, 9, false), "", "scan")) return rc;
  if (int rc = build(rtc_merge_source(p, 4, 9, agg, w, false), "_merge", "merge")) return rc;
  if (int rc = build(rtc_scan_source(p, 4, 9, true), "_compact", "compact scan")) return rc;
  if (int rc = build(rtc_merge_source(p, 4, 9, agg, w, true), "_cmerge", "compact merge")) return rc;
  if (int rc = build(rtc_table_scan_source(p, 4, 9, agg, w), "_table", "table scan")) return rc;
  // table images (hash_reduce_lds.hip): the merge that leaves its table in HBM, and the one that starts from it
  if (int rc = build(rtc_merge_source(p, 4, 9, agg, w, true, false, 1), "_cmerge_img1", "compact merge + image out")) return rc;
  if (int rc = build(rtc_merge_source(p, 4, 9, agg, w, true, false, 2), "_cmerge_img2", "compact merge from image")) return rc;
  if (int rc = build(rtc_merge_source(p, 4, 9, agg, w, false, false, 2), "_merge_img2", "merge from image")) return rc;
  {  // another shape: two dimensions, int32 measure summed into 4 bytes, no nulls, 8 partitions
    FusedPlanD q = p;
    q.numCols = 3;
    for (int c = 0; c < 3; c++) q.cols[c].nulls = nullptr;
    q.numFilters = 0;
    q.dims[0] = p.dims[1]; q.dims[0].col = 0;
    q.dims[1] = p.dims[2]; q.dims[1].col = 1; q.dims[1].f.arity = 2; q.dims[1].f.functor = Plus; q.dims[1].f.bkind = K_I32; q.dims[1].f.bbits = 5; q.dims[1].f.bok = 1;
    q.measure.f = col(K_I32); q.measure.col = 2; q.measure.outKind = K_I32;
    q.measureDtype = Int32; q.measureWidth = 4; q.identity = 0;
    AggSpec a4 = make_agg_spec(AGGR_SUM_SIGNED, 4);
    hr::Widen w4{0, K_I32, Int32};
    if (int rc = build(rtc_scan_source(q, 2, 3, true), "_compact2", "compact scan (2 dims)")) return rc;
    if (int rc = build(rtc_merge_source(q, 2, 3, a4, w4, true), "_cmerge2", "compact merge (2 dims)")) return rc;
    if (int rc = build(rtc_table_scan_source(q, 2, 0, a4, w4), "_table2", "table scan (2 dims, 1 partition)")) return rc;
  }
  {  // a narrow plan, the reference's example schema (examples/1k_trips/schema/trips.json): dimensions [Floor(request_at, 3600)
     // Uint32, city_id Uint16 -> a 2-byte slot], SUM(fare), filters request_at >= / < (time range), status == k on a Uint8 column
    FusedPlanD t = p;
    t.numCols = 4;
    t.cols[0].step = 4; t.cols[1].step = 2; t.cols[2].step = 4; t.cols[3].step = 1;
    t.dims[1] = p.dims[1]; t.dims[1].col = 1;
    t.dimWidth[0] = 4; t.dimWidth[1] = 2;
    t.measure.f = col(K_F32); t.measure.col = 2; t.measure.outKind = K_F32;
    t.numFilters = 3;
    t.filters[0] = p.filters[0]; t.filters[0].f.functor = GreaterThanOrEqual; t.filters[0].col = 0;
    t.filters[1] = p.filters[0]; t.filters[1].f.functor = LessThan; t.filters[1].col = 0;
    t.filters[2] = p.filters[0]; t.filters[2].f.functor = Equal; t.filters[2].col = 3;
    if (int rc = build(rtc_scan_source(t, 2, 9, true), "_ncompact", "narrow compact scan")) return rc;
    if (int rc = build(rtc_merge_source(t, 2, 9, agg, w, true), "_ncmerge", "narrow compact merge")) return rc;
    if (int rc = build(rtc_scan_source(t, 2, 9, false), "_nlines", "narrow scan")) return rc;
    if (int rc = build(rtc_merge_source(t, 2, 9, agg, w, false), "_nmerge", "narrow merge")) return rc;
    if (int rc = build(rtc_table_scan_source(t, 2, 9, agg, w), "_ntable", "narrow table scan")) return rc;
    if (int rc = build(rtc_merge_source(t, 2, 9, agg, w, false, true), "_namerge", "narrow region-A merge")) return rc;
    if (int rc = build(rtc_merge_source(t, 2, 9, agg, w, true, false, 2), "_ncmerge_img2", "narrow compact merge from image")) return rc;
    // signed narrow columns and a 1-byte slot: dimensions [Int16 column -> 2-byte slot, Int8 column -> 1-byte slot], no nulls
    FusedPlanD u = t;
    u.numFilters = 0; u.numCols = 3;
    for (int c = 0; c < 3; c++) u.cols[c].nulls = nullptr;
    u.cols[0].step = 2; u.cols[1].step = 1; u.cols[2].step = 4;
    u.dims[0].f = col(K_I32); u.dims[0].col = 0; u.dims[0].outKind = K_I32;
    u.dims[1].f = col(K_I32); u.dims[1].col = 1; u.dims[1].outKind = K_I32;
    u.dimWidth[0] = 2; u.dimWidth[1] = 1;
    if (int rc = build(rtc_scan_source(u, 2, 9, true), "_scompact", "signed narrow compact scan")) return rc;
    if (int rc = build(rtc_merge_source(u, 2, 9, agg, w, true), "_scmerge", "signed narrow compact merge")) return rc;
  }
  {  // the Sort + Reduce path (sort_reduce_fused.hip): records keyed by lo64(murmur3_x64_128), constant measure (COUNT(*)) and a
     // column measure summed into 8 bytes, the C3 dimensions and the narrow trips shape
    FusedPlanD c = p;
    c.numCols = 4;  // no measure column: the filter's d1 is dimension 1's column
    c.measure.col = -1; c.measure.f = col(K_U32); c.measure.f.bbits = 1; c.measureDtype = Uint32; c.measureWidth = 4; c.identity = 0;
    if (int rc = build(rtc_sort_scan_source(c, 4, 9), "_sort_count", "sort scan (COUNT)")) return rc;
    {
      FusedPlanD c2 = c;
      c2.measure.f.bbits = 7;
      if (rtc_sort_scan_source(c2, 4, 9) != rtc_sort_scan_source(c, 4, 9)) { puts("the constant measure leaks into the source"); return 22; }
    }
    FusedPlanD m8 = p;
    m8.measure.f = col(K_U32); m8.measure.outKind = K_I32; m8.measureDtype = Int64; m8.measureWidth = 8;
    if (int rc = build(rtc_sort_scan_source(m8, 4, 9), "_sort_sum8", "sort scan (SUM into 8 bytes)")) return rc;
    FusedPlanD t = p;  // trips: dims [Floor(request_at, 3600) Uint32, city_id Uint16 -> 2-byte slot], COUNT(*), three filters
    t.numCols = 3;
    t.cols[0].step = 4; t.cols[1].step = 2; t.cols[2].step = 1;
    t.dims[1] = p.dims[1]; t.dims[1].col = 1;
    t.dimWidth[0] = 4; t.dimWidth[1] = 2;
    t.measure = c.measure; t.measureDtype = Uint32; t.measureWidth = 4;
    t.numFilters = 3;
    t.filters[0] = p.filters[0]; t.filters[0].f.functor = GreaterThanOrEqual; t.filters[0].col = 0;
    t.filters[1] = p.filters[0]; t.filters[1].f.functor = LessThan; t.filters[1].col = 0;
    t.filters[2] = p.filters[0]; t.filters[2].f.functor = Equal; t.filters[2].col = 2;
    if (int rc = build(rtc_sort_scan_source(t, 2, 9), "_sort_trips", "narrow sort scan (COUNT)")) return rc;
  }
  {  // eight dimensions (MAX_DIMENSIONS, query/time_series_aggregate.h:36-37): six 4-byte slots, a 2-byte and a 1-byte one, the
     // measure and a filter on a column of its own — ten column slots
    FusedPlanD e = p;
    e.numCols = 10;
    for (int c = 0; c < 10; c++) { e.cols[c].vals = reinterpret_cast<const uint32_t *>(0x1000); e.cols[c].nulls = c % 3 ? reinterpret_cast<const uint8_t *>(0x2000) : nullptr; e.cols[c].step = 4; }
    e.cols[6].step = 2; e.cols[7].step = 1;
    for (int d = 0; d < 8; d++) { e.dims[d].f = col(K_U32); e.dims[d].col = d; e.dims[d].outKind = K_U32; e.dimWidth[d] = 4; }
    e.dims[0] = p.dims[0];
    e.dimWidth[6] = 2; e.dimWidth[7] = 1;
    e.measure.f = col(K_F32); e.measure.col = 8; e.measure.outKind = K_F32;
    e.numFilters = 1; e.filters[0] = p.filters[0]; e.filters[0].col = 9;
    if (int rc = build(rtc_scan_source(e, 8, 9, true), "_8compact", "8-dimension compact scan")) return rc;
    if (int rc = build(rtc_merge_source(e, 8, 9, agg, w, true), "_8cmerge", "8-dimension compact merge")) return rc;
    if (int rc = build(rtc_merge_source(e, 8, 9, agg, w, true, false, 2), "_8cmerge_img2", "8-dimension compact merge from image")) return rc;
    if (int rc = build(rtc_table_scan_source(e, 8, 9, agg, w), "_8table", "8-dimension table scan")) return rc;
    if (int rc = build(rtc_merge_source(e, 8, 9, agg, w, false, true), "_8amerge", "8-dimension region-A merge")) return rc;
    FusedPlanD ec = e;
    ec.numCols = 9; ec.cols[8] = e.cols[9]; ec.filters[0].col = 8;
    ec.measure.col = -1; ec.measure.f = col(K_U32); ec.measure.f.bbits = 1; ec.measureDtype = Uint32; ec.measureWidth = 4;
    if (int rc = build(rtc_sort_scan_source(ec, 8, 9), "_8sort", "8-dimension sort scan (COUNT)")) return rc;
  }
  // the vector-sourced sort scans (Sort + Reduce over materialised vectors: 64-bit row hash, up to eight 4-byte dimensions)
  if (int rc = build(rtc_sort_vector_scan_source(2, nullptr, 9), "_vsort2", "vector sort scan nd 2")) return rc;
  if (int rc = build(rtc_sort_vector_scan_source(8, nullptr, 9), "_vsort8", "vector sort scan nd 8")) return rc;
  if (int rc = build(rtc_sort_vector_scan_source(1, nullptr, 0), "_vsort1", "vector sort scan nd 1, one partition")) return rc;
  {
    const int narrow[4] = {4, 4, 2, 1};
    if (int rc = build(rtc_sort_vector_scan_source(4, narrow, 9), "_vsort_narrow", "vector sort scan, slots 4 4 2 1")) return rc;
  }
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
