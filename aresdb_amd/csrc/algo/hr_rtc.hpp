// Run-time compiled, per-plan specialised scan / merge kernels of the fused HashReduce (hr_rtc.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <memory>
#include <string>

#include "hash_reduce_lds.hpp"

#include "aggregate.hpp"

namespace ares {
namespace hr {
struct Workspace;
struct Widen;
}

// A loaded kernel.  The handle keeps it alive: the cache may drop the entry (least recently used shapes go when it
// is full), the module is unloaded when the last handle is gone.  Null = not available (unsupported shape, no
// hiprtc, does not compile, or still being compiled on a background thread): the caller uses the generic kernel.
struct RtcEntry;
using RtcKernel = std::shared_ptr<RtcEntry>;

// hiprtc could be loaded (and ARES_RTC is not 0)
bool rtc_scan_available();
// workgroups (= private record streams per partition) for a batch of `rows` rows
int rtc_scan_grid(int64_t rows);
// tiles (4096 rows) each workgroup of the compact scan walks — its chunk is contiguous —, or 0 when a chunk would
// not fit the record's row field (chunk rows <= 1 << (partBits + 9)): the caller takes the 16-byte records
int rtc_compact_chunk_tiles(int64_t rows, int partBits);

// Every lookup generates the kernel's source for the plan's SHAPE (comparison and + - x constants are kernel
// arguments, divisors literals), and returns the loaded kernel or null.  `wait`: build it on this thread if it is
// not there yet (otherwise it is built in the background and null is returned this time; ARES_RTC_ASYNC=0 makes
// every lookup wait).

// DIRECT-mode scan of `plan` (every surviving row becomes a record in the workgroup's private stream of its
// partition): `compact` = 8-byte records in compact lines (hr::Workspace::lineRecords == 14, ws.chunkRows set),
// otherwise 16-byte records in lines of 8.
RtcKernel rtc_scan_lookup(int device, const FusedPlanD &plan, int nd, int partBits, bool compact, bool wait = false);
void rtc_scan_launch(const RtcKernel &kernel, const FusedPlanD &plan, uint32_t rowBase, int length, const hr::Workspace &ws,
                     hipStream_t stream);
// The scan of the fused Sort + Reduce path (sort_reduce_fused.hip): like the DIRECT scan with 16-byte line records, but keyed
// by lo64(murmur3_x64_128) of the packed row — records {row, hash >> 32, carried measure, (u32)hash}, partition = top bits of
// the 64-bit hash.  plan.measure.col < 0: constant measure (the records carry plan.measure.f.bbits).
RtcKernel rtc_sort_scan_lookup(int device, const FusedPlanD &plan, int nd, int partBits, bool wait = false);
void rtc_sort_scan_launch(const RtcKernel &kernel, const FusedPlanD &plan, uint32_t rowBase, int length, const hr::Workspace &ws,
                          hipStream_t stream);
std::string rtc_sort_scan_source(const FusedPlanD &plan, int nd, int partBits);
// ... and over rows [rowBase, rowBase + length) of a materialised dimension vector of `nd` dimensions (widths: 4 / 2 / 1 bytes in vector
// order, null = all four bytes) and a measure vector of 4-byte values
// (Sort + Reduce after joins or generic expressions): the same records, the value carried whole; the level-1 partition is the
// hash's top partBits bits or — spread — the scrambled low partBits bits of its top totalPartBits bits (sort_reduce_fused.hip)
RtcKernel rtc_sort_vector_scan_lookup(int device, int nd, const int *widths, int partBits, bool wait = false);
void rtc_sort_vector_scan_launch(const RtcKernel &kernel, const uint8_t *dimValues, size_t capacity, const uint8_t *values, int nd,
                                 const int *widths, uint32_t rowBase, int length, int totalPartBits, bool spread, const hr::Workspace &ws,
                                 hipStream_t stream);
std::string rtc_sort_vector_scan_source(int nd, const int *widths, int partBits);
// TABLE-mode scan of `plan` (low cardinality): LDS aggregation per workgroup, one record per group into region A
// (what hr::flush_table writes), rtc_scan_grid(length) workgroups; the generic merge reads it.
RtcKernel rtc_table_scan_lookup(int device, const FusedPlanD &plan, int nd, int partBits, const AggSpec &a, const hr::Widen &w,
                                bool wait = false);
void rtc_table_scan_launch(const RtcKernel &kernel, const FusedPlanD &plan, uint32_t rowBase, int length, const hr::Workspace &ws,
                           hipStream_t stream);
// The DIRECT scan over rows [rowBase, rowBase + length) of a dimension vector of `nd` 4-byte dimensions and a
// measure vector of `vw`-byte values (HashReduce on materialised vectors); 16-byte records carry the whole value.
RtcKernel rtc_vector_scan_lookup(int device, int nd, int vw, int partBits, bool wait = false);
void rtc_vector_scan_launch(const RtcKernel &kernel, const uint8_t *dimValues, size_t capacity, const uint8_t *values, int nd, int vw,
                            uint32_t rowBase, int length, const hr::Workspace &ws, hipStream_t stream);
// The specialised merge for what the DIRECT scans produce (line records in region B, previous groups in their
// partition-grouped ranges or none, one round over the whole hash range).  It raises outCount[3] when a
// partition holds more groups than one LDS table: the caller then runs the generic merge.
// regionA: the records come from region A as well (TABLE-mode scans) — the merge of narrow plans' low-cardinality batches.
// image: 0 = none; 1 = the merge also leaves the partition's LDS table in HBM ("table image": keys, each group's output
// position, values — 128 KB per partition); 2 = the merge STARTS from the previous call's image, emits the dimension rows of
// new groups only (appended: a group keeps its position) and writes the image again — the measure vector stays unwritten.
RtcKernel rtc_merge_lookup(int device, const FusedPlanD &plan, int nd, int partBits, const AggSpec &a, const hr::Widen &w, bool compact,
                           bool wait = false, bool regionA = false, int image = 0);
struct RtcImageArgs {  // device pointers: images of hr::kSlots x 16 bytes per partition, one group count per partition
  const void *in;
  void *out;
  const uint32_t *inCount;
  uint32_t *outCount;
  uint32_t knownOut;  // leading rows of the output dimension vector that already hold the query's groups (image == 2)
};
void rtc_merge_launch(const RtcKernel &kernel, const FusedPlanD &plan, const uint8_t *prevDims, size_t prevCapacity,
                      const uint8_t *prevValues, uint32_t prevSize, uint8_t *dimOut, size_t outCapacity, uint8_t *outValues,
                      const hr::Workspace &ws, hipStream_t stream, const RtcImageArgs *image = nullptr, uint32_t *hostOut = nullptr);
// hostOut: four words of host memory the device can write (the calling thread's pinned slot): the kernel's last workgroup
// leaves ws.outCount[0 .. 3] there, so the caller waits for the stream instead of enqueueing a copy (ws.outCount[4] must be 0)
// ... and for what the vector-sourced scan produces (launched with rtc_merge_launch and an empty plan: every
// row, old or new, is a row of the input vectors passed as prevDims / prevValues)
RtcKernel rtc_vector_merge_lookup(int device, int nd, int vw, int partBits, const AggSpec &a, bool wait = false);

// the generated sources (empty = unsupported shape); for tools and tests
std::string rtc_scan_source(const FusedPlanD &plan, int nd, int partBits, bool compact = false);
std::string rtc_table_scan_source(const FusedPlanD &plan, int nd, int partBits, const AggSpec &a, const hr::Widen &w);
std::string rtc_merge_source(const FusedPlanD &plan, int nd, int partBits, const AggSpec &a, const hr::Widen &w, bool compact = false,
                             bool regionA = false, int image = 0);
std::string rtc_vector_scan_source(int nd, int vw, int partBits);
std::string rtc_vector_merge_source(int nd, int vw, int partBits, const AggSpec &a);

}  // namespace ares
