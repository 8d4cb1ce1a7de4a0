// HashReduce for MI355X (gfx950): group-by aggregation through an open-addressed hash table.
//
// Reference: query/hash_reduction.cu:183-391 (+ cudf concurrent_unordered_map on the device,
// std::unordered_map on the host, query/concurrent_unordered_map.hpp).  Observable contract:
//   * group identity = murmur3_x86_32(seed 0) of the packed dim row — equality on the 32-bit hash
//     only (hash_reduction.cu:216-243);
//   * the representative row of a group is the FIRST row carrying the hash; the reference's
//     device build leaves "first" to the CAS race, its host build means input order.  We pin it
//     to input order deterministically: the slot key (hash << 32 | row) is lowered with a 64-bit
//     atomicMin, so the smallest row index always wins, whatever the scheduling;
//   * output rows [0, groups) in unspecified order; res = groups.
// The table has 2 * length slots rounded up to a power of two (reference load factor 2); keys and
// values live in two separate arrays so the probe sequence touches 8-byte keys only.
#include <hip/hip_runtime.h>

#include "aggregate.hpp"
#include "common.hpp"
#include "device_model.hpp"
#include "dim_layout.hpp"
#include "hash_reduce_lds.hpp"

namespace ares {

constexpr int kBlock = 256;
constexpr uint64_t kEmptyKey = ~0ull;

__global__ __launch_bounds__(kBlock) void hash_table_init_kernel(uint64_t *keys, uint8_t *values, uint64_t capacity,
                                                                 AggSpec a) {
  for (uint64_t s = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; s < capacity;
       s += static_cast<uint64_t>(gridDim.x) * kBlock) {
    keys[s] = kEmptyKey;
    if (a.width == 8) reinterpret_cast<uint64_t *>(values)[s] = a.identity;
    else reinterpret_cast<uint32_t *>(values)[s] = static_cast<uint32_t>(a.identity);
  }
}

// Finds (or claims) the slot of hash h and lowers its key to min(key, h<<32|row).
__device__ __forceinline__ uint64_t find_or_claim(uint64_t *keys, uint64_t mask, uint32_t h, uint32_t row) {
  const uint64_t mine = (static_cast<uint64_t>(h) << 32) | row;
  uint64_t slot = h & mask;
  for (;;) {
    uint64_t cur = __hip_atomic_load(keys + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == kEmptyKey) {
      const uint64_t prev = atomicCAS(reinterpret_cast<unsigned long long *>(keys + slot), kEmptyKey, mine);
      if (prev == kEmptyKey) return slot;
      cur = prev;
    }
    if (static_cast<uint32_t>(cur >> 32) == h) {
      if (mine < cur) atomicMin(reinterpret_cast<unsigned long long *>(keys + slot), mine);
      return slot;
    }
    slot = (slot + 1) & mask;
  }
}

__global__ __launch_bounds__(kBlock) void hash_insert_kernel(const uint8_t *dimValues, DimLayoutD L, size_t capacity,
                                                             const uint8_t *inputValues, AggSpec a, uint64_t *keys,
                                                             uint8_t *values, uint64_t mask, int length) {
  for (int64_t i64 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i64 < length;
       i64 += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint32_t row = static_cast<uint32_t>(i64);
    Murmur32Stream ms(0);
    hash_dim_row(ms, dimValues, L, capacity, row);
    const uint32_t h = ms.finish();
    const uint64_t slot = find_or_claim(keys, mask, h, row);
    aggregate_slot(values + slot * a.width, inputValues + static_cast<size_t>(a.width) * row, a);
  }
}

// Compacts the occupied slots into output rows.  One returning atomic on a single word costs
// ~11 ns and a word sustains < 100 of them per microsecond, so survivors are counted per
// 4096-slot tile (wave scan + LDS) and each tile reserves its output range with ONE atomicAdd.
constexpr int kExItems = 16;
__global__ __launch_bounds__(kBlock) void hash_extract_kernel(const uint64_t *keys, const uint8_t *values,
                                                              uint64_t tableSize, const uint8_t *dimIn, uint8_t *dimOut,
                                                              DimLayoutD L, size_t capacity, uint8_t *outputValues,
                                                              AggSpec a, uint32_t *counter) {
  __shared__ uint32_t sWave[kBlock / 64];
  __shared__ uint32_t sBase;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t tileSlots = static_cast<uint64_t>(kBlock) * kExItems;
  for (uint64_t base = blockIdx.x * tileSlots; base < tableSize; base += gridDim.x * tileSlots) {
    uint64_t k[kExItems];
    uint32_t cnt = 0;
#pragma unroll
    for (int it = 0; it < kExItems; it++) {
      const uint64_t s = base + static_cast<uint64_t>(it) * kBlock + threadIdx.x;
      k[it] = s < tableSize ? keys[s] : kEmptyKey;
      cnt += k[it] != kEmptyKey;
    }
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t total = 0;
      for (int w = 0; w < kBlock / 64; w++) {
        const uint32_t c = sWave[w];
        sWave[w] = total;
        total += c;
      }
      sBase = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t dst = sBase + sWave[wave] + incl - cnt;
#pragma unroll
    for (int it = 0; it < kExItems; it++) {
      if (k[it] == kEmptyKey) continue;
      const uint64_t s = base + static_cast<uint64_t>(it) * kBlock + threadIdx.x;
      copy_dim_row(dimIn, capacity, dimOut, capacity, L, static_cast<uint32_t>(k[it]), dst);
      if (a.width == 8) reinterpret_cast<uint64_t *>(outputValues)[dst] = reinterpret_cast<const uint64_t *>(values)[s];
      else reinterpret_cast<uint32_t *>(outputValues)[dst] = reinterpret_cast<const uint32_t *>(values)[s];
      dst++;
    }
    __syncthreads();  // sWave / sBase are reused by the next tile
  }
}

}  // namespace ares

using namespace ares;

// ARES_HASH_REDUCE=global pins the global-table path (tests exercise both implementations).
namespace ares {
bool global_table_forced() {
  static EnvSwitch<bool> forced("ARES_HASH_REDUCE", [](const char *e) { return e && strcmp(e, "global") == 0; });
  return forced.get();
}
}  // namespace ares

extern "C" CGoCallResHandle HashReduce(DimensionVector inputKeys, uint8_t *inputValues, DimensionVector outputKeys,
                                       uint8_t *outputValues, int valueBytes, int length,
                                       enum AggregateFunction aggFunc, void *cudaStream, int device) {
  ARES_ABI_BEGIN_NOFLUSH(device)
  hipStream_t stream = reinterpret_cast<hipStream_t>(cudaStream);
  if (length > 0) {  // lazily filled inputs are written first, lazily filled outputs retired
    settle_dimension_vector(device, inputKeys);
    const DimLayoutD inLayout = make_dim_layout(inputKeys.NumDimsPerDimWidth);
    const size_t cap = inputKeys.VectorCapacity > 0 ? static_cast<size_t>(inputKeys.VectorCapacity) : 0;
    materialize_fills_for_read(device, inputKeys.DimValues, static_cast<size_t>(inLayout.rowBytes) * cap);
    materialize_fills_for_read(device, inputValues, static_cast<size_t>(valueBytes) * length);
    retire_fills_for_write(device, outputKeys.DimValues, static_cast<size_t>(inLayout.rowBytes) * cap);
    retire_fills_for_write(device, outputValues, static_cast<size_t>(valueBytes) * length);
  }
  {  // whatever an earlier HashReduce skipped that would write into this call's outputs is dead
    const DimLayoutD outLayout = make_dim_layout(outputKeys.NumDimsPerDimWidth);
    drop_skipped_outputs(device, outputKeys.DimValues,
                         static_cast<size_t>(outLayout.rowBytes) * (inputKeys.VectorCapacity > 0 ? inputKeys.VectorCapacity : 0),
                         outputValues, static_cast<size_t>(valueBytes) * (length > 0 ? length : 0));
  }
  // the dimension / measure transforms of this batch may still be pending: evaluate them on the fly
  int fusedGroups = 0;
  if (length > 0 && fuse_pending_into_hash_reduce(device, stream, inputKeys, inputValues, outputKeys, outputValues,
                                                  valueBytes, length, aggFunc, &fusedGroups)) {
    resHandle.res = int_result(fusedGroups);
    return resHandle;
  }
  {  // launch what is pending; of the work an earlier HashReduce skipped only what this call reads
    const DimLayoutD layout = make_dim_layout(inputKeys.NumDimsPerDimWidth);
    flush_deferred_for_inputs(device, inputKeys.DimValues, static_cast<size_t>(layout.rowBytes) * inputKeys.VectorCapacity,
                              inputValues, static_cast<size_t>(valueBytes) * (length > 0 ? length : 0));
  }
  const AggSpec a = make_agg_spec(aggFunc, valueBytes);
  int groups = -1;
  if (length > 0 && hash_reduce_lds_supported(a) && !global_table_forced())
    groups = hash_reduce_lds(device, inputKeys, inputValues, outputKeys, outputValues, a, length, stream);
  if (groups >= 0) {
    resHandle.res = int_result(groups);
  } else if (length > 0) {
    const DimLayoutD L = make_dim_layout(inputKeys.NumDimsPerDimWidth);
    grouped_note_write(device, outputValues, static_cast<size_t>(a.width) * length);
    grouped_note_write(device, outputKeys.DimValues, static_cast<size_t>(L.rowBytes) * inputKeys.VectorCapacity);
    uint64_t tableSize = 1024;
    while (tableSize < 2ull * static_cast<uint64_t>(length)) tableSize <<= 1;
    StreamBuffer keyBuf(tableSize * 8, stream), valBuf(tableSize * a.width, stream), counter(16, stream);
    hip_check(hipMemsetAsync(counter.get(), 0, 16, stream), "hipMemsetAsync");
    const int initGrid = capped_grid(static_cast<int64_t>((tableSize + kBlock - 1) / kBlock), 256 * 16);
    ARES_LAUNCH("hash_table_init_kernel", hash_table_init_kernel, initGrid, kBlock, stream, keyBuf.as<uint64_t>(),
                       valBuf.as<uint8_t>(), tableSize, a);
    const int grid = capped_grid((static_cast<int64_t>(length) + kBlock - 1) / kBlock, 256 * 16);
    ARES_LAUNCH("hash_insert_kernel", hash_insert_kernel, grid, kBlock, stream, inputKeys.DimValues, L,
                       static_cast<size_t>(inputKeys.VectorCapacity), inputValues, a, keyBuf.as<uint64_t>(),
                       valBuf.as<uint8_t>(), tableSize - 1, length);
    const int exGrid = capped_grid(static_cast<int64_t>((tableSize + kBlock * kExItems - 1) / (kBlock * kExItems)), 256 * 8);
    ARES_LAUNCH("hash_extract_kernel", hash_extract_kernel, exGrid, kBlock, stream, keyBuf.as<uint64_t>(),
                       valBuf.as<uint8_t>(), tableSize, inputKeys.DimValues, outputKeys.DimValues, L,
                       static_cast<size_t>(inputKeys.VectorCapacity), outputValues, a, counter.as<uint32_t>());
    uint32_t groups = 0;
    read_back_u32(counter.as<uint32_t>(), &groups, 1, stream);
    mem_note_dim_rows(device, outputKeys, 0, groups);
    mem_note_write(device, outputValues, static_cast<size_t>(a.width) * groups);
    resHandle.res = int_result(groups);
  }
  ARES_ABI_END("HashReduce")
}
