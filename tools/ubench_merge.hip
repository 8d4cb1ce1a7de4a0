// Microbenchmark: where does the time of the per-partition merge go?  512 workgroups (one per
// partition) stream 256 record runs each (16-byte records, ~117 k per partition) and aggregate them in
// an 8192-slot LDS table, in variants that add one cost at a time and with both record layouts.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o tools/bin/ubench_merge tools/ubench_merge.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32;
typedef uint64_t u64;

constexpr u32 NP = 512, G = 256, SLOTS = 8192, GROUPS = 4500;
__device__ __forceinline__ u32 fmix(u32 h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

struct Args {
  uint4 *rec;
  u32 *counts;  // [g][p]
  u32 cap;
  u32 *sink;
  int skew;     // 1: 29 % of the records fall on 2 % of the groups (the Zipf dimension of config C3)
  int wgMajor;  // 1: stream (g, p) at (g * NP + p) * cap; 0: (p * G + g) * cap
};
__device__ __forceinline__ u64 stream_base(const Args &a, u32 g, u32 p) {
  return a.wgMajor ? (static_cast<u64>(g) * NP + p) * a.cap : (static_cast<u64>(p) * G + g) * a.cap;
}

__global__ void fill(Args a, u32 mean) {
  const u32 g = blockIdx.x / NP, p = blockIdx.x % NP;
  const u32 cnt = (mean - 24 + fmix(blockIdx.x * 77u + 5u) % 48u) & ~7u;
  uint4 *run = a.rec + stream_base(a, g, p);
  for (u32 i = threadIdx.x; i < cnt; i += blockDim.x) {
    const u32 r = fmix(blockIdx.x * 1000003u + i);
    const u32 r2 = fmix(r + 99u);
    const u32 grp = a.skew ? ((r % 100u < 29u) ? r2 % 90u : (r % 100u < 43u) ? 90u + r2 % 90u : 180u + r2 % (GROUPS - 180u)) : r % GROUPS;
    const u32 h = (p << 23) | (fmix(grp * 2654435761u + p) & 0x7FFFFFu);
    run[i] = make_uint4(1000000u + g * 4096u + i, h, __float_as_uint(static_cast<float>(r % 400u) * 0.25f), 0u);
  }
  if (threadIdx.x == 0) a.counts[g * NP + p] = cnt;
}

struct Chunk { const uint4 *ptr; u32 rem; };
struct Stage { uint4 r[4]; };
__device__ __forceinline__ void load_chunk(Stage &s, const Chunk &c, u32 lane) {
  const u32 last = c.rem ? c.rem - 1u : 0u;
#pragma unroll
  for (int k = 0; k < 4; k++) { const u32 i = static_cast<u32>(k) * 64u + lane; s.r[k] = c.ptr[i < last ? i : last]; }
}

// MODE: 0 loads only; 1 + home-slot key read and compare; 2 + ds_add_f64 at the home slot (no probing: the
// table is direct-mapped for timing); 3 = 2 with keys claimed by CAS on first touch (linear probing)
template <int MODE>
__global__ void __launch_bounds__(1024) merge_waves(Args a) {
  __shared__ u64 sKeys[SLOTS];
  __shared__ u64 sVals[SLOTS];
  __shared__ u32 sRun[G];
  const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, p = blockIdx.x;
  for (u32 s = tid; s < SLOTS; s += 1024u) { sKeys[s] = ~0ull; sVals[s] = 0ull; }
  if (tid < G) sRun[tid] = a.counts[tid * NP + p];
  __syncthreads();
  u32 dummy = 0;
  u32 g = wave, off = 0;
  auto next = [&]() -> Chunk {
    Chunk c{a.rec, 0u};
    while (g < G) {
      const u32 cnt = static_cast<u32>(__builtin_amdgcn_readfirstlane(static_cast<int>(sRun[g])));
      if (off < cnt) { c.ptr = a.rec + stream_base(a, g, p) + off; c.rem = cnt - off; off += 256u; break; }
      g += 16u; off = 0u;
    }
    return c;
  };
  auto consume = [&](const Stage &s, const Chunk &c) {
    const u32 take = c.rem < 256u ? c.rem : 256u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 i = static_cast<u32>(k) * 64u + lane;
      if (i >= take) continue;
      const u32 h = s.r[k].y;
      if (MODE == 0) { dummy += h ^ s.r[k].z; continue; }
      u32 slot = h & (SLOTS - 1);
      if (MODE == 3) {
        const u64 mine = (static_cast<u64>(h) << 32) | s.r[k].x;
        for (;;) {
          u64 cur = sKeys[slot];
          if (cur == ~0ull) {
            unsigned long long expected = ~0ull;
            if (__hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned long long *>(sKeys + slot), &expected, static_cast<unsigned long long>(mine), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
            cur = expected;
          }
          if (static_cast<u32>(cur >> 32) == h) break;
          slot = (slot + 1) & (SLOTS - 1);
        }
      } else {
        const u64 key = sKeys[slot];
        if (static_cast<u32>(key >> 32) == h) dummy++;
      }
      if (MODE >= 2)
        __hip_atomic_fetch_add(reinterpret_cast<double *>(sVals + slot), static_cast<double>(__uint_as_float(s.r[k].z)), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  Stage sa, sb;
  Chunk ca = next();
  load_chunk(sa, ca, lane);
  while (ca.rem) {
    Chunk cb = next();
    load_chunk(sb, cb, lane);
    consume(sa, ca);
    if (!cb.rem) break;
    ca = next();
    load_chunk(sa, ca, lane);
    consume(sb, cb);
  }
  __syncthreads();
  if (dummy == 0x12345678u || sVals[tid] == 0x1234ull) a.sink[0] = dummy;
}

// the whole workgroup walks the partition's runs as ONE flattened list: lane i takes record i (binary
// search in the prefix sums of the run lengths), 4 records per lane per stage
template <int MODE>
__global__ void __launch_bounds__(1024) merge_flat(Args a) {
  __shared__ u64 sKeys[SLOTS];
  __shared__ u64 sVals[SLOTS];
  __shared__ u32 sStart[G + 1];
  const u32 tid = threadIdx.x, p = blockIdx.x;
  for (u32 s = tid; s < SLOTS; s += 1024u) { sKeys[s] = ~0ull; sVals[s] = 0ull; }
  if (tid == 0) {
    u32 acc = 0;
    for (u32 g = 0; g < G; g++) { sStart[g] = acc; acc += a.counts[g * NP + p]; }
    sStart[G] = acc;
  }
  __syncthreads();
  const u32 total = sStart[G];
  u32 dummy = 0;
  auto fetch = [&](u32 i) -> uint4 {
    if (i >= total) i = total - 1u;
    u32 lo = 0, hi = G;  // largest g with sStart[g] <= i
    while (hi - lo > 1u) { const u32 mid = (lo + hi) >> 1; if (sStart[mid] <= i) lo = mid; else hi = mid; }
    return a.rec[stream_base(a, lo, p) + (i - sStart[lo])];
  };
  uint4 ra[4], rb[4];
  auto load = [&](uint4 (&r)[4], u32 base) {
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = fetch(base + static_cast<u32>(k) * 1024u + tid);
  };
  auto consume = [&](const uint4 (&r)[4], u32 base) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (base + static_cast<u32>(k) * 1024u + tid >= total) continue;
      const u32 h = r[k].y;
      if (MODE == 0) { dummy += h ^ r[k].z; continue; }
      const u32 slot = h & (SLOTS - 1);
      const u64 key = sKeys[slot];
      if (static_cast<u32>(key >> 32) == h) dummy++;
      if (MODE >= 2)
        __hip_atomic_fetch_add(reinterpret_cast<double *>(sVals + slot), static_cast<double>(__uint_as_float(r[k].z)), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  load(ra, 0);
  for (u32 base = 0; base < total; base += 8192u) {
    load(rb, base + 4096u);
    consume(ra, base);
    load(ra, base + 8192u);
    consume(rb, base + 4096u);
  }
  __syncthreads();
  if (dummy == 0x12345678u || sVals[tid] == 0x1234ull) a.sink[0] = dummy;
}

template <typename F>
float time_it(F &&launch, int reps = 6) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int r = 0; r < reps; r++) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const u32 mean = 460, cap = 1096;
  Args a;
  CHECK(hipMalloc(&a.rec, static_cast<size_t>(NP) * G * cap * 16));
  CHECK(hipMalloc(&a.counts, NP * G * 4));
  CHECK(hipMalloc(&a.sink, 64));
  a.cap = cap;
  const double gb = static_cast<double>(NP) * G * mean * 16 / 1e9;
  for (int variant = 0; variant < 2; variant++) {
    const int wgMajor = 1;
    a.wgMajor = wgMajor;
    a.skew = variant;
    hipLaunchKernelGGL(fill, dim3(NP * G), dim3(256), 0, 0, a, mean);
    CHECK(hipDeviceSynchronize());
    printf("---- groups %s ----\n", a.skew ? "skewed like C3 (29 %% of records on 2 %% of groups)" : "uniform");
    auto report = [&](const char *name, float ms) { printf("%-58s %.3f ms  (%.0f GB/s of records)\n", name, ms, gb / (ms * 1e-3)); fflush(stdout); };
    report("per-wave runs: loads only", time_it([&] { hipLaunchKernelGGL(merge_waves<0>, dim3(NP), dim3(1024), 0, 0, a); }));
    report("per-wave runs: + home-slot key read", time_it([&] { hipLaunchKernelGGL(merge_waves<1>, dim3(NP), dim3(1024), 0, 0, a); }));
    report("per-wave runs: + ds_add_f64 (direct-mapped)", time_it([&] { hipLaunchKernelGGL(merge_waves<2>, dim3(NP), dim3(1024), 0, 0, a); }));
    report("per-wave runs: claim + linear probing + ds_add_f64", time_it([&] { hipLaunchKernelGGL(merge_waves<3>, dim3(NP), dim3(1024), 0, 0, a); }));
    report("flattened list: loads only", time_it([&] { hipLaunchKernelGGL(merge_flat<0>, dim3(NP), dim3(1024), 0, 0, a); }));
    report("flattened list: + key read + ds_add_f64", time_it([&] { hipLaunchKernelGGL(merge_flat<2>, dim3(NP), dim3(1024), 0, 0, a); }));
  }
  return 0;
}
