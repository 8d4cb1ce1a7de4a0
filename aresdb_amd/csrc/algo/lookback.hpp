// Single-pass (chained, "decoupled look-back") prefix sums across the tiles of one launch.
//
// Used by the filter (stream compaction), Reduce (segment numbering), Expand (output offsets)
// and the radix sort (per-digit scatter bases).  Protocol per tile, on gfx950's 8 XCDs with
// non-coherent L2s (cdna_hip_programming.md, Guideline 16 "R2": the data is the flag):
//   * tiles are handed out by an atomic ticket, so every predecessor of a running tile is
//     resident or finished and the spin below cannot deadlock;
//   * a tile publishes ONE naturally aligned word {flag, value} with a relaxed agent-scope
//     atomic store (write-through, sc1): first its own aggregate, later its inclusive prefix;
//   * successors poll those words with relaxed agent-scope atomic loads only.
// No fences are needed because no other data is handed between workgroups inside the launch.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ares {

constexpr uint64_t kFlagAggregate = 1ull << 62;
constexpr uint64_t kFlagInclusive = 2ull << 62;
constexpr uint64_t kFlagMask = 3ull << 62;
constexpr uint64_t kValueMask = ~kFlagMask;
// Every spin is bounded (a predecessor that never publishes would otherwise hang the GPU): after
// ~2^22 polls (seconds) the waiting lane raises the launch's error word and proceeds with garbage;
// the host turns the error word into an exception.
constexpr uint32_t kMaxSpins = 1u << 22;

__device__ __forceinline__ uint64_t ld_status(const uint64_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(uint64_t *p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exclusive prefix of `tile` from its predecessors' status words; must be called by all 64
// lanes of ONE wavefront.  Lane l inspects tile (look - l); the nearest predecessor that already
// holds an inclusive prefix terminates the walk.
__device__ __forceinline__ uint64_t lookback_wave(uint64_t *status, int tile, int lane, uint32_t *error) {
  uint64_t exclusive = 0;
  int look = tile - 1;
  while (look >= 0) {
    const int t = look - lane;
    uint64_t w;
    if (t >= 0) {
      w = ld_status(status + t);
      uint32_t spins = 0;
      while ((w & kFlagMask) == 0) {
        __builtin_amdgcn_s_sleep(2);
        w = ld_status(status + t);
        if (++spins > kMaxSpins) {  // never hang the device: flag the launch as failed instead
          atomicOr(error, 1u);
          w = kFlagInclusive;
        }
      }
    } else {
      w = kFlagInclusive;  // before the first tile: inclusive prefix 0
    }
    const uint64_t inclusiveLanes = __ballot((w & kFlagMask) == kFlagInclusive);
    const int stop = inclusiveLanes ? __builtin_ctzll(inclusiveLanes) : 64;
    uint64_t v = lane <= stop ? (w & kValueMask) : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    exclusive += v;
    if (inclusiveLanes) break;
    look -= 64;
  }
  return exclusive;
}

}  // namespace ares
