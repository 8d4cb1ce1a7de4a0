// Pieces of the sort-based group-by path that HyperLogLog reuses (sort_reduce.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "dim_layout.hpp"

namespace ares {

// keyVector[i] = 64-bit row hash of dimension row rowIndex[i] (with hllValues: the HyperLogLog sort
// key of entry i), then a stable sort of (keyVector, payload) by key.  iotaPayload: payload is
// initialised to the entry positions first.
void sort_rows(const uint8_t *dimValues, const DimLayoutD &L, size_t capacity, const uint32_t *rowIndex,
               const uint32_t *hllValues, uint64_t *keyVector, uint32_t *payload, bool iotaPayload, int length,
               hipStream_t stream);

// runs of equal keys -> (key, indexSrc[first position], max valuesSrc[position]); returns the runs
int hll_reduce_sorted(const uint64_t *keys, const uint32_t *positions, const uint32_t *indexSrc,
                      const uint32_t *valuesSrc, uint64_t *hashOut, uint32_t *indexOut, uint32_t *valuesOut,
                      int length, hipStream_t stream);

}  // namespace ares
