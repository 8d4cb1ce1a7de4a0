// Columnar dimension vectors (group-by keys) on the device.
//
// Layout (reference query/common/dimval.go:122-144, query/iterator.hpp:934-1025): for each
// dimension, in width order 16,8,4,2,1, `capacity * width` value bytes; then one validity byte
// vector of `capacity` bytes per dimension.  The "packed row" that is hashed is
// [value bytes of every dim][one validity byte per dim].  Widths descend, so every field is
// naturally aligned inside the packed row and can be streamed into murmur3 without ever
// materialising the row in memory (no per-thread byte array, no scratch traffic).
#pragma once

#include <hip/hip_runtime.h>

#include <stdexcept>

#include "ares_algorithm.h"
#include "device_model.hpp"

namespace ares {

constexpr int kMaxDims = 16;

struct DimLayoutD {
  int numDims;
  int valueBytes;  // sum of widths
  int rowBytes;    // valueBytes + numDims
  uint8_t width[kMaxDims];
  uint16_t valueOff[kMaxDims];  // byte offset of the dim's value vector, in units of `capacity`
};

inline DimLayoutD make_dim_layout(const uint8_t numDimsPerDimWidth[NUM_DIM_WIDTH]) {
  DimLayoutD L;
  memset(&L, 0, sizeof(L));
  int d = 0, off = 0;
  for (int w = 0; w < NUM_DIM_WIDTH; w++) {
    const int bytes = 1 << (NUM_DIM_WIDTH - 1 - w);
    for (int j = 0; j < numDimsPerDimWidth[w]; j++) {
      if (d >= kMaxDims) throw std::invalid_argument("too many dimensions");
      L.width[d] = static_cast<uint8_t>(bytes);
      L.valueOff[d] = static_cast<uint16_t>(off);
      off += bytes;
      d++;
    }
  }
  L.numDims = d;
  L.valueBytes = off;
  L.rowBytes = off + d;
  if (L.rowBytes > 64) throw std::invalid_argument("dimension row wider than 64 bytes");
  return L;
}

// ---- streaming murmur3_x86_32 (query/utils.cu:113-155) -------------------------------------------
struct Murmur32Stream {
  uint32_t h, stage, fill, total;
  __device__ __forceinline__ explicit Murmur32Stream(uint32_t seed) : h(seed), stage(0), fill(0), total(0) {}
  __device__ __forceinline__ void block(uint32_t k) {
    k *= 0xcc9e2d51u;
    k = rotl32(k, 15) * 0x1b873593u;
    h ^= k;
    h = rotl32(h, 13) * 5u + 0xe6546b64u;
  }
  // push `bytes` (1, 2, 4 or 8) little-endian bytes; fields are naturally aligned in the row
  __device__ __forceinline__ void push(uint64_t v64, uint32_t bytes) {
    total += bytes;
    const uint32_t v = static_cast<uint32_t>(v64);
    if (bytes == 8) { block(v); block(static_cast<uint32_t>(v64 >> 32)); return; }
    if (bytes == 4) { block(v); return; }
    stage |= v << (8 * fill);
    fill += bytes;
    if (fill == 4) { block(stage); stage = 0; fill = 0; }
  }
  __device__ __forceinline__ uint32_t finish() {
    uint32_t k = stage * 0xcc9e2d51u;  // zero tail mixes to zero
    k = rotl32(k, 15) * 0x1b873593u;
    h ^= k;
    h ^= total;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
  }
};

// ---- streaming murmur3_x64_128, low 64 bits (query/utils.cu:157-241) ----------------------------
struct Murmur128Stream {
  uint64_t h1, h2, k1, k2;
  uint32_t fill, total;
  __device__ __forceinline__ explicit Murmur128Stream(uint32_t seed)
      : h1(seed), h2(seed), k1(0), k2(0), fill(0), total(0) {}
  __device__ __forceinline__ void block() {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    k1 = k2 = 0;
    fill = 0;
  }
  // push `bytes` (1, 2, 4 or 8) little-endian bytes
  __device__ __forceinline__ void push(uint64_t v, uint32_t bytes) {
    total += bytes;
    if (fill < 8) k1 |= v << (8 * fill);
    else k2 |= v << (8 * (fill - 8));
    fill += bytes;
    if (fill == 16) block();
  }
  __device__ __forceinline__ uint64_t finish() {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    h1 ^= total; h2 ^= total;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    return h1;
  }
};

// Feeds the packed row of `row` into a murmur stream: values in width order, then validity bytes.
template <typename Stream>
__device__ __forceinline__ void hash_dim_row(Stream &s, const uint8_t *dimValues, const DimLayoutD &L,
                                             size_t capacity, uint32_t row) {
  for (int d = 0; d < L.numDims; d++) {
    const uint32_t w = L.width[d];
    const uint8_t *p = dimValues + static_cast<size_t>(L.valueOff[d]) * capacity + static_cast<size_t>(w) * row;
    switch (w) {
      case 16: {
        const uint4 v = *reinterpret_cast<const uint4 *>(p);
        s.push((static_cast<uint64_t>(v.y) << 32) | v.x, 8);
        s.push((static_cast<uint64_t>(v.w) << 32) | v.z, 8);
        break;
      }
      case 8: {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        s.push((static_cast<uint64_t>(v.y) << 32) | v.x, 8);
        break;
      }
      case 4: s.push(*reinterpret_cast<const uint32_t *>(p), 4); break;
      case 2: s.push(*reinterpret_cast<const uint16_t *>(p), 2); break;
      default: s.push(*p, 1); break;
    }
  }
  const uint8_t *nulls = dimValues + static_cast<size_t>(L.valueBytes) * capacity;
  for (int d = 0; d < L.numDims; d++) s.push(nulls[static_cast<size_t>(d) * capacity + row], 1);
}

// Copies the dims (+validity) of input row `src` to output row `dst`; both vectors use `capacity`
// for their strides (the reference uses inputKeys.VectorCapacity for both, sort_reduce.cu:234-239,
// hash_reduction.cu:101-141).
__device__ __forceinline__ void copy_dim_row(const uint8_t *in, size_t inCap, uint8_t *out, size_t outCap,
                                             const DimLayoutD &L, uint32_t src, uint32_t dst) {
  for (int d = 0; d < L.numDims; d++) {
    const uint32_t w = L.width[d];
    const uint8_t *p = in + static_cast<size_t>(L.valueOff[d]) * inCap + static_cast<size_t>(w) * src;
    uint8_t *q = out + static_cast<size_t>(L.valueOff[d]) * outCap + static_cast<size_t>(w) * dst;
    switch (w) {
      case 16: *reinterpret_cast<uint4 *>(q) = *reinterpret_cast<const uint4 *>(p); break;
      case 8: *reinterpret_cast<uint2 *>(q) = *reinterpret_cast<const uint2 *>(p); break;
      case 4: *reinterpret_cast<uint32_t *>(q) = *reinterpret_cast<const uint32_t *>(p); break;
      case 2: *reinterpret_cast<uint16_t *>(q) = *reinterpret_cast<const uint16_t *>(p); break;
      default: *q = *p; break;
    }
  }
  const uint8_t *inNulls = in + static_cast<size_t>(L.valueBytes) * inCap;
  uint8_t *outNulls = out + static_cast<size_t>(L.valueBytes) * outCap;
  for (int d = 0; d < L.numDims; d++)
    outNulls[static_cast<size_t>(d) * outCap + dst] = inNulls[static_cast<size_t>(d) * inCap + src];
}

}  // namespace ares
