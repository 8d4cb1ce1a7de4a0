// Run-time compiled, per-plan specialised scan kernel of the fused HashReduce (hr_rtc.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "hash_reduce_lds.hpp"

#include "aggregate.hpp"

namespace ares {
namespace hr {
struct Workspace;
struct Widen;
}

// hiprtc could be loaded (and ARES_RTC is not 0)
bool rtc_scan_available();
// workgroups (= private record streams per partition) for a batch of `rows` rows
int rtc_scan_grid(int64_t rows);
// The specialised DIRECT-mode scan kernel of `plan` on `device` (generated, compiled and cached on first
// use); nullptr = the plan is outside the supported shapes or the kernel could not be built: the caller
// uses the generic kernel.
void *rtc_scan_lookup(int device, const FusedPlanD &plan, int nd, int partBits);
// Launches it over rows [0, length) of the plan's columns into the private streams of `ws`
// (ws.streams workgroups; 16-byte records in whole 128-byte lines, ws.capB a multiple of 8).
void rtc_scan_launch(void *kernel, const FusedPlanD &plan, uint32_t rowBase, int length, const hr::Workspace &ws,
                     hipStream_t stream);
// The same scan over rows [rowBase, rowBase + length) of a dimension vector of `nd` 4-byte dimensions and a
// measure vector of `vw`-byte values (HashReduce on materialised vectors); records carry the whole value.
void *rtc_vector_scan_lookup(int device, int nd, int vw, int partBits);
void rtc_vector_scan_launch(void *kernel, const uint8_t *dimValues, size_t capacity, const uint8_t *values, int nd, int vw,
                            uint32_t rowBase, int length, const hr::Workspace &ws, hipStream_t stream);
std::string rtc_vector_scan_source(int nd, int vw, int partBits);
// The specialised merge for what that scan produces (line records in region B, previous groups in their
// partition-grouped ranges or none, one round over the whole hash range).  It raises outCount[3] when a
// partition holds more groups than one LDS table: the caller then runs the generic merge.
void *rtc_merge_lookup(int device, const FusedPlanD &plan, int nd, int partBits, const AggSpec &a, const hr::Widen &w);
void rtc_merge_launch(void *kernel, const FusedPlanD &plan, const uint8_t *prevDims, size_t prevCapacity, const uint8_t *prevValues,
                      uint32_t prevSize, uint8_t *dimOut, size_t outCapacity, uint8_t *outValues, const hr::Workspace &ws,
                      hipStream_t stream);
std::string rtc_merge_source(const FusedPlanD &plan, int nd, int partBits, const AggSpec &a, const hr::Widen &w);
// ... and for what the vector-sourced scan produces (launched with rtc_merge_launch and an empty plan: every
// row, old or new, is a row of the input vectors passed as prevDims / prevValues)
void *rtc_vector_merge_lookup(int device, int nd, int vw, int partBits, const AggSpec &a);
std::string rtc_vector_merge_source(int nd, int vw, int partBits, const AggSpec &a);
// the generated source (empty = unsupported shape); for tools and tests
std::string rtc_scan_source(const FusedPlanD &plan, int nd, int partBits);

}  // namespace ares
