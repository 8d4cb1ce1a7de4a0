// Sort + Reduce without sorting rows: the hash-keyed group-by behind the reference's default aggregation path
// (sort_reduce_fused.hip; host-side deferral: transform.hip "lazy sorts").
#pragma once

#include <hip/hip_runtime.h>

#include "aggregate.hpp"
#include "ares_algorithm.h"
#include "hash_reduce_lds.hpp"

namespace ares {

// ARES_SORT_FUSE=0 switches the path off (every Sort sorts rows); needs hiprtc for the scan
bool fused_sort_reduce_enabled();
// integer SUM / MIN / MAX on 4 bytes, integer SUM on 8: a float aggregate depends on the order of its rows
bool fused_sort_reduce_supported(const AggSpec &a);

// Sort + Reduce over the previous result (rows [0, prevSize) of `in` / `inValues`) and the batch described by `plan`
// (filters, dimension expressions and measure over the batch's source columns; plan.measure.col < 0 + constMeasure: every
// surviving row carries the value pattern `constBits`).  Writes the groups — ascending 64-bit row hash, dimension row of
// the lowest-indexed row, aggregated value — to rows [0, groups) of `out` / `outValues` (strided by in.VectorCapacity, as
// the reference: query/sort_reduce.cu:234-239) and returns their number.  out.IndexVector, in.HashValues and
// in.IndexVector are NOT written (the caller keeps them defined: transform.hip).
// kFusedUnavailable: declined before anything was launched; -1: launched, outputs possibly part-written, take the real
// Sort + Reduce (a partition's table or record stream overflowed, a row hash equals the table's empty word).
int fused_sort_reduce_run(int device, const FusedPlanD &plan, int nd, bool constMeasure, uint64_t constBits, int batchRows,
                          const DimensionVector &in, const uint8_t *inValues, int prevSize, const DimensionVector &out,
                          uint8_t *outValues, const AggSpec &a, hipStream_t stream);

// The same over MATERIALISED vectors: rows [0, length) of `in` / `inValues` exist (4-byte dimensions, a 4-byte integer
// aggregate); rows a previous fused Reduce left there are recognised by the row hashes kept beside them.  Returns alike.
int fused_sort_reduce_vectors(int device, int length, const DimensionVector &in, const uint8_t *inValues, const DimensionVector &out,
                              uint8_t *outValues, const AggSpec &a, hipStream_t stream);

// something writes [ptr, ptr + bytes): row hashes a fused Reduce kept beside a result in there are forgotten (called by
// grouped_note_write, which every writer of a result vector reports to)
void sorted_state_note_write(int device, const void *ptr, size_t bytes);

// sort_reduce.hip: the real thing, for a lazily defined Sort (+ Reduce) that somebody reads after all
void sort_keys_now(const DimensionVector &keys, int length, hipStream_t stream);
int reduce_now(const DimensionVector &in, uint8_t *inputValues, const DimensionVector &out, uint8_t *outputValues, int valueBytes,
               int length, int aggFunc, hipStream_t stream);
// transform.hip: Sort over rows whose transforms are still pending is DEFINED, not run (true = nothing left to do) ...
bool define_lazy_sort(int device, hipStream_t stream, const DimensionVector &keys, int length);
// ... or, when the rows exist (written by kernels): asked BEFORE Sort's flush (the index vector's lazy iota is kept from being
// written), defined AFTER it (false: the index vector has been written, the caller sorts)
bool lazy_vector_sort_candidate(int device, const DimensionVector &keys, int length);
bool define_lazy_sort_vectors(int device, hipStream_t stream, const DimensionVector &keys, int length);
// ... and Reduce consumes it together with the pending transforms (true = *groups is the result); false: whatever was lazy
// about the inputs has been written, the caller runs the ordinary Reduce
bool fuse_pending_into_sort_reduce(int device, hipStream_t stream, const DimensionVector &in, uint8_t *inValues,
                                   const DimensionVector &out, uint8_t *outValues, int valueBytes, int length, int aggFunc, int *groups);

}  // namespace ares
