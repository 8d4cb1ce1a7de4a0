// Partitioned, LDS-resident HashReduce (hash_reduce_lds.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "aggregate.hpp"
#include "ares_algorithm.h"

namespace ares {

// true when the aggregate has a native LDS atomic (sums of 4/8-byte integers and floats, integer
// min/max); AVG and float min/max stay on the global-table path.
bool hash_reduce_lds_supported(const AggSpec &a);

// Returns the number of groups, or -1 when a partition region overflowed (the caller then runs the
// global-table path; outputs written so far are simply overwritten).
int hash_reduce_lds(const DimensionVector &inputKeys, const uint8_t *inputValues, const DimensionVector &outputKeys,
                    uint8_t *outputValues, const AggSpec &a, int length, hipStream_t stream);

}  // namespace ares
