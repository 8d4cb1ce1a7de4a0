// PROTOTYPE (next round): cost of the fix-up that follows a radix sort of the TOP 32 bits only (sort_topbits_fixup.hpp).
// Host: n random 64-bit keys (`distinct` different values, so keys repeat like group hashes do), stably sorted by their
// top half — what four one-sweep passes over bits 32..63 leave.  Device: detect + sort kernels, timed; result checked
// against a stable sort by the whole key.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -o ../bin/ubench_sort_topbits ubench_sort_topbits.hip && ../bin/ubench_sort_topbits [n] [distinct]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#include "sort_topbits_fixup.hpp"

using namespace ares_proto;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ __launch_bounds__(256) void fixup_detect_kernel(FixupParams p) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < p.n; i += static_cast<int64_t>(gridDim.x) * 256)
    fixup_detect(p, static_cast<int>(i), [](uint32_t *ctr) { return atomicAdd(ctr, 1u); });
}
__global__ __launch_bounds__(64) void fixup_sort_kernel(FixupParams p) {
  const uint32_t count = *p.workCount < p.workCap ? *p.workCount : p.workCap;
  for (uint32_t w = blockIdx.x * 64 + threadIdx.x; w < count; w += gridDim.x * 64) fixup_sort(p, w);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1 << 24;
  const int distinct = argc > 2 ? atoi(argv[2]) : n / 2;
  std::mt19937_64 rng(7);
  std::vector<uint64_t> pool(distinct);
  for (auto &k : pool) k = rng();
  std::vector<uint64_t> keys(n);
  for (auto &k : keys) k = pool[rng() % distinct];
  std::vector<int> perm(n);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return (keys[a] >> 32) < (keys[b] >> 32); });
  std::vector<uint64_t> k(n);
  std::vector<uint32_t> v(n);
  for (int i = 0; i < n; i++) { k[i] = keys[perm[i]]; v[i] = static_cast<uint32_t>(perm[i]); }
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return keys[a] < keys[b]; });

  uint64_t *dk; uint32_t *dv, *dctr; Segment *dwork;
  const uint32_t cap = static_cast<uint32_t>(n / 8 + 1024);
  CHECK(hipMalloc(&dk, 8ull * n)); CHECK(hipMalloc(&dv, 4ull * n)); CHECK(hipMalloc(&dctr, 8)); CHECK(hipMalloc(&dwork, sizeof(Segment) * cap));
  FixupParams p{dk, dv, n, dwork, dctr, cap, dctr + 1, 64};
  hipEvent_t e0, e1, e2;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
  float detect = 0, sort = 0;
  uint32_t ctr[2] = {0, 0};
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipMemcpy(dk, k.data(), 8ull * n, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dv, v.data(), 4ull * n, hipMemcpyHostToDevice));
    CHECK(hipMemset(dctr, 0, 8));
    CHECK(hipEventRecord(e0));
    fixup_detect_kernel<<<256 * 16, 256>>>(p);
    CHECK(hipEventRecord(e1));
    fixup_sort_kernel<<<256 * 4, 64>>>(p);
    CHECK(hipEventRecord(e2));
    CHECK(hipEventSynchronize(e2));
    CHECK(hipEventElapsedTime(&detect, e0, e1)); CHECK(hipEventElapsedTime(&sort, e1, e2));
    CHECK(hipMemcpy(ctr, dctr, 8, hipMemcpyDeviceToHost));
  }
  std::vector<uint64_t> outK(n);
  std::vector<uint32_t> outV(n);
  CHECK(hipMemcpy(outK.data(), dk, 8ull * n, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(outV.data(), dv, 4ull * n, hipMemcpyDeviceToHost));
  size_t bad = 0;
  if (!ctr[1])
    for (int i = 0; i < n; i++) bad += outK[i] != keys[perm[i]] || outV[i] != static_cast<uint32_t>(perm[i]);
  printf("n %d distinct %d: detect %.3f ms (%.0f GB/s of keys), sort %.3f ms, %u segments listed, fallback %u, mismatches %zu\n", n, distinct,
         detect, 8.0 * n / (detect * 1e-3) / 1e9, sort, ctr[0], ctr[1], bad);
  return bad ? 1 : 0;
}
