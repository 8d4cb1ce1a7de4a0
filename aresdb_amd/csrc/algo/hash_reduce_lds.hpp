// Partitioned, LDS-resident HashReduce (hash_reduce_lds.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "aggregate.hpp"
#include "ares_algorithm.h"
#include "fast_eval.hpp"

namespace ares {

// true when the aggregate has a native LDS atomic (sums of 4/8-byte integers and floats, integer
// min/max); AVG and float min/max stay on the global-table path.
bool hash_reduce_lds_supported(const AggSpec &a);

// Returns the number of groups, or -1 when a partition region overflowed (the caller then runs the
// global-table path; outputs written so far are simply overwritten).
int hash_reduce_lds(int device, const DimensionVector &inputKeys, const uint8_t *inputValues,
                    const DimensionVector &outputKeys, uint8_t *outputValues, const AggSpec &a, int length,
                    hipStream_t stream);

// ---- fused scan: filter + projection evaluated from the source columns ------------------------------
// (kFusedFilters: the Go host's two time filters, a cutoff filter on live batches and three of the query's own)
// kFusedDims / kFusedCols (round 6): what the ABI allows — MAX_DIMENSIONS = 8 (query/time_series_aggregate.h:36-37) — in slots
// of 4, 2 or 1 bytes; eight dimensions + measure + a filter column = ten column slots.  More than four dimensions run on the
// kernels generated for the plan's shape only (the precompiled generic kernels are instantiated for one to four).
constexpr int kFusedCols = 10, kFusedFilters = 6, kFusedDims = 8;
constexpr int kGenericFusedDims = 4;
constexpr int kExtensionFilters = 4;  // AresFusedQuery::filters (include/ares_extensions.h)
struct FusedColumn {
  const uint32_t *vals;
  const uint8_t *nulls;
  uint32_t bitOff;
  uint32_t step;  // bytes per stored value (FastOperands::step): 4, 2 or 1; 0 reads as 4
};
struct FusedExpr {
  FastOperands f;  // akind / arity / functor / I / rk / constant / divLike (pointers unused)
  int col;
  int outKind;     // kind of the stored dimension value
};
struct FusedPlanD {
  int numCols;
  FusedColumn cols[kFusedCols];
  int numFilters;
  FusedExpr filters[kFusedFilters];
  FusedExpr dims[kFusedDims];
  FusedExpr measure;
  int measureDtype, measureWidth;
  uint64_t identity;  // measure-transform identity of the aggregate (query/utils.hpp:169-184)
  // bytes of dimension d's slot in the dimension vector, in vector (= descending width) order: 4, 2 or 1; 0 reads as 4.
  // Plans with a narrow slot or a narrow column ("narrow plans") run on the kernels generated for their shape only
  // (hr_rtc.hip); the precompiled generic kernels read 4-byte columns and write 4-byte slots.
  uint8_t dimWidth[kFusedDims];
};
inline int fused_dim_width(const FusedPlanD &p, int d) { return p.dimWidth[d] ? p.dimWidth[d] : 4; }
inline int fused_col_step(const FusedPlanD &p, int c) { return p.cols[c].step ? static_cast<int>(p.cols[c].step) : 4; }
inline bool fused_plan_narrow(const FusedPlanD &p, int nd) {
  if (nd > kGenericFusedDims) return true;  // (generated kernels only, like a plan with narrow slots)
  for (int d = 0; d < nd; d++)
    if (fused_dim_width(p, d) != 4) return true;
  for (int c = 0; c < p.numCols; c++)
    if (fused_col_step(p, c) != 4) return true;
  return false;
}

// Column slots of a plan of ND dimensions: dimension d -> slot d, measure -> slot ND; a filter reuses
// a slot that already holds its column or takes the one spare slot ND + 1.
// Returns the number of groups; -1: a partition region overflowed (or a partition needs the generic multi-round merge and
// the plan is narrow) — outputs may be partly written; kFusedUnavailable: declined before anything was launched (narrow
// plan whose generated kernels are not loaded yet, previous results not grouped ...).  Either way the caller runs the
// unfused sequence.
constexpr int kFusedUnavailable = -2;
// Room of a scanning workgroup's private record stream of one partition: (2 << slack) x the mean + 64.  Batches whose rows
// arrive SORTED (archive batches: a workgroup's contiguous chunk holds two or three time buckets, hence few distinct groups
// and unevenly filled partitions) overflow the default (slack 0: twice the mean); the call that sees the overflow flag grows
// the slack — it stays grown for the process — and runs again instead of leaving the fast path.
int record_stream_slack();
bool grow_record_stream_slack();  // false: already at its limit (8 x the mean)
int fused_hash_reduce_run(int device, const FusedPlanD &plan, int batchRows, const DimensionVector &prevKeys,
                          const uint8_t *prevValues, int prevSize, const DimensionVector &outKeys, uint8_t *outValues,
                          const AggSpec &a, hipStream_t stream);

}  // namespace ares
