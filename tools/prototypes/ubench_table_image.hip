// PROTOTYPE (next round): what a partition's table image costs to carry between HashReduce calls.
// One 1024-lane workgroup per partition, 512 partitions (two per CU, as hr_merge_rtc runs them), each with the merge's
// LDS table: 8192 keys (u32), 8192 representative rows (u32), 8192 values (u64) = 128 KB.
//   image     load the three arrays from HBM into LDS with 16-byte loads, touch them (one LDS atomic per lane and round,
//             stands for the batch's ~3.7 k records), store them back with 16-byte stores
//   reinsert  what the merge does today for its ~4.5 k previous groups: read 4 dimension values + 4 validity bytes + the
//             value per group from the previous output vectors, hash (murmur3 over the row), probe the LDS table linearly,
//             claim, add; then count and write every occupied slot's group out again (20 B dims + 8 B value)
// Both print microseconds per launch; the difference is what step (a) of the plan can win per 2 Mi-row batch.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -o ../bin/ubench_table_image ubench_table_image.hip && ../bin/ubench_table_image
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int kSlots = 8192, kParts = 512, kGroups = 4500;

__global__ __launch_bounds__(1024) void image_kernel(uint4 *images /* [parts][kSlots * 16 B / 16] */, int touches) {
  __shared__ uint4 sTable[kSlots];  // keys | rows | values, 16 bytes per slot in three planes: addressed as one block
  uint4 *mine = images + static_cast<size_t>(blockIdx.x) * kSlots;
  for (int i = threadIdx.x; i < kSlots; i += 1024) sTable[i] = mine[i];
  __syncthreads();
  unsigned *words = reinterpret_cast<unsigned *>(sTable);
  for (int t = 0; t < touches; t++) atomicAdd(words + ((threadIdx.x * 2654435761u + t * 40503u) & (kSlots * 4 - 1)), 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < kSlots; i += 1024) mine[i] = sTable[i];
}

__device__ __forceinline__ unsigned rotl(unsigned x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ unsigned mix(unsigned h, unsigned k) {
  k *= 0xcc9e2d51u; k = rotl(k, 15) * 0x1b873593u; h ^= k; h = rotl(h, 13);
  return h * 5u + 0xe6546b64u;
}

__global__ __launch_bounds__(1024) void reinsert_kernel(const unsigned *dims /* [4][cap] */, const unsigned char *nulls /* [4][cap] */,
                                                        const double *values, size_t cap, unsigned *outDims, unsigned char *outNulls,
                                                        double *outValues, unsigned *outCount) {
  __shared__ unsigned sKeys[kSlots];
  __shared__ unsigned sRows[kSlots];
  __shared__ double sVals[kSlots];
  __shared__ unsigned sBase, sEmit;
  for (int i = threadIdx.x; i < kSlots; i += 1024) { sKeys[i] = 0u; sRows[i] = ~0u; sVals[i] = 0.0; }
  if (threadIdx.x == 0) sEmit = 0u;
  __syncthreads();
  const size_t first = static_cast<size_t>(blockIdx.x) * kGroups;
  for (int g = threadIdx.x; g < kGroups; g += 1024) {
    const size_t row = first + g;
    unsigned h = 0u, ok = 0u;
    for (int d = 0; d < 4; d++) { h = mix(h, dims[d * cap + row]); ok |= static_cast<unsigned>(nulls[d * cap + row]) << (8 * d); }
    h = mix(h, ok); h ^= 20u; h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    const unsigned key = h | 1u;
    unsigned s = h & (kSlots - 1);
    for (;;) {
      const unsigned old = atomicCAS(&sKeys[s], 0u, key);
      if (old == 0u || old == key) break;
      s = (s + 1) & (kSlots - 1);
    }
    atomicMin(&sRows[s], static_cast<unsigned>(row));
    atomicAdd(&sVals[s], values[row]);
  }
  __syncthreads();
  unsigned mine = 0;
  for (int i = threadIdx.x; i < kSlots; i += 1024) mine += sKeys[i] != 0u;
  if (mine) atomicAdd(&sEmit, mine);
  __syncthreads();
  if (threadIdx.x == 0) { sBase = atomicAdd(outCount, sEmit); sEmit = 0u; }
  __syncthreads();
  for (int i = threadIdx.x; i < kSlots; i += 1024) {
    if (sKeys[i] == 0u) continue;
    const unsigned at = sBase + atomicAdd(&sEmit, 1u), row = sRows[i];
    for (int d = 0; d < 4; d++) { outDims[d * cap + at] = dims[d * cap + row]; outNulls[d * cap + at] = nulls[d * cap + row]; }
    outValues[at] = sVals[i];
  }
}

template <typename F>
static float time_us(F &&launch) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int i = 0; i < 20; i++) launch();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.0f / 20;
}

int main() {
  uint4 *images;
  CHECK(hipMalloc(&images, sizeof(uint4) * kSlots * kParts));
  CHECK(hipMemset(images, 0, sizeof(uint4) * kSlots * kParts));
  const size_t cap = static_cast<size_t>(kParts) * kGroups;
  std::vector<unsigned> hd(4 * cap);
  for (size_t i = 0; i < hd.size(); i++) hd[i] = static_cast<unsigned>(i * 2654435761u >> 7);
  unsigned *dims, *outDims, *count; unsigned char *nulls, *outNulls; double *values, *outValues;
  CHECK(hipMalloc(&dims, 16 * cap)); CHECK(hipMalloc(&outDims, 16 * cap)); CHECK(hipMalloc(&nulls, 4 * cap)); CHECK(hipMalloc(&outNulls, 4 * cap));
  CHECK(hipMalloc(&values, 8 * cap)); CHECK(hipMalloc(&outValues, 8 * cap)); CHECK(hipMalloc(&count, 4));
  CHECK(hipMemcpy(dims, hd.data(), 16 * cap, hipMemcpyHostToDevice)); CHECK(hipMemset(nulls, 1, 4 * cap)); CHECK(hipMemset(values, 0, 8 * cap));
  for (int touches : {0, 4}) {
    const float us = time_us([&] { image_kernel<<<kParts, 1024>>>(images, touches); });
    printf("image load + %d LDS atomics per lane + store, %d partitions: %.1f us per launch (%.0f GB/s)\n", touches, kParts, us,
           2.0 * sizeof(uint4) * kSlots * kParts / (us * 1e-6) / 1e9);
  }
  const float us = time_us([&] { CHECK(hipMemsetAsync(count, 0, 4)); reinsert_kernel<<<kParts, 1024>>>(dims, nulls, values, cap, outDims, outNulls, outValues, count); });
  printf("re-insert %d groups per partition from the previous vectors + emit all of them: %.1f us per launch\n", kGroups, us);
  return 0;
}
