// Design micro-benchmark (not part of the product): what do random read-modify-writes cost on
// MI355X as a function of table size and scope?  Drives the HashReduce design.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_atomics.hip -o /tmp/ubench_atomics
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);        \
      return 1;                                                               \
    }                                                                         \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}

enum { M_ADD_F64_AGENT, M_ADD_F64_WG, M_LOAD, M_ADD_U32_AGENT, M_KEYLOAD_ADD_SAMELINE, M_CAS_AGENT, M_LDS_ADD_F64, M_LDS_ADD_U32 };

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *table, uint64_t mask, int64_t n, uint64_t *sink) {
  __shared__ uint64_t lds[8192];  // 64 KB
  if (MODE == M_LDS_ADD_F64 || MODE == M_LDS_ADD_U32) {
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0;
    __syncthreads();
  }
  uint64_t acc = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256) {
    const uint64_t h = (static_cast<uint64_t>(mix(static_cast<uint32_t>(i))) | (static_cast<uint64_t>(mix(static_cast<uint32_t>(i) ^ 0x9e3779b9u)) << 32));
    const uint64_t s = h & mask;
    if (MODE == M_ADD_F64_AGENT) {
      __hip_atomic_fetch_add(reinterpret_cast<double *>(table) + s, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == M_ADD_F64_WG) {
      __hip_atomic_fetch_add(reinterpret_cast<double *>(table) + s, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == M_LOAD) {
      acc += table[s];
    } else if (MODE == M_ADD_U32_AGENT) {
      __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(table) + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == M_KEYLOAD_ADD_SAMELINE) {
      const uint64_t s2 = s & ~1ull;  // 16-byte slot: key at [0], value at [1]
      acc += __hip_atomic_load(table + s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(reinterpret_cast<double *>(table) + s2 + 1, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == M_CAS_AGENT) {
      unsigned long long exp = 0;
      __hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned long long *>(table) + s, &exp, h | 1ull, __ATOMIC_RELAXED,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += exp;
    } else if (MODE == M_LDS_ADD_F64) {
      __hip_atomic_fetch_add(reinterpret_cast<double *>(lds) + (s & 8191), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == M_LDS_ADD_U32) {
      __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(lds) + (s & 16383), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  if (MODE == M_LDS_ADD_F64 || MODE == M_LDS_ADD_U32) {
    __syncthreads();
    acc += lds[threadIdx.x];
  }
  if (acc == 0x1234567887654321ull) *sink = acc;
}

template <int MODE>
static int run(const char *name, uint64_t *table, uint64_t slots, int64_t n, uint64_t *sink, int grid) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipMemset(table, 0, slots * 8));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, table, slots - 1, n, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("{\"op\": \"%s\", \"table_MB\": %.2f, \"ops\": %lld, \"ms\": %.3f, \"Gops_per_s\": %.2f}\n", name, slots * 8 / 1048576.0,
         static_cast<long long>(n), best, n / best / 1e6);
  fflush(stdout);
  return 0;
}

int main() {
  const int64_t n = 1ll << 26;
  const uint64_t maxSlots = 1ull << 29;  // 4 GB
  uint64_t *table, *sink;
  CK(hipMalloc(&table, maxSlots * 8));
  CK(hipMalloc(&sink, 8));
  const int grid = 256 * 8;
  const uint64_t sizes[] = {1ull << 15, 1ull << 19, 1ull << 22, 1ull << 24, 1ull << 25, 1ull << 27, 1ull << 29};
  for (uint64_t slots : sizes) {
    if (run<M_ADD_F64_AGENT>("atomicAdd f64 agent", table, slots, n, sink, grid)) return 1;
    if (run<M_ADD_F64_WG>("atomicAdd f64 workgroup-scope", table, slots, n, sink, grid)) return 1;
    if (run<M_ADD_U32_AGENT>("atomicAdd u32 agent", table, slots, n, sink, grid)) return 1;
    if (run<M_LOAD>("load u64", table, slots, n, sink, grid)) return 1;
    if (run<M_KEYLOAD_ADD_SAMELINE>("key load + f64 add, same 16B slot", table, slots, n, sink, grid)) return 1;
    if (run<M_CAS_AGENT>("CAS u64 agent", table, slots, n, sink, grid)) return 1;
  }
  if (run<M_LDS_ADD_F64>("LDS atomicAdd f64 (64KB/block)", table, 1 << 15, n, sink, grid)) return 1;
  if (run<M_LDS_ADD_U32>("LDS atomicAdd u32 (64KB/block)", table, 1 << 15, n, sink, grid)) return 1;
  return 0;
}
