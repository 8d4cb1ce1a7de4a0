// Run-time compiled, per-plan specialised scan kernel of the fused HashReduce (hr_rtc.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "hash_reduce_lds.hpp"

namespace ares {
namespace hr {
struct Workspace;
}

// hiprtc could be loaded (and ARES_RTC is not 0)
bool rtc_scan_available();
// workgroups (= private record streams per partition) for a batch of `rows` rows
int rtc_scan_grid(int64_t rows);
// Launches the specialised DIRECT-mode scan of `plan` over rows [0, length) of its columns into the
// private streams of `ws` (ws.streams workgroups).  false = the plan is outside the supported shapes or
// the kernel could not be built: the caller launches the generic kernel instead.
bool rtc_scan_launch(int device, const FusedPlanD &plan, int nd, uint32_t rowBase, int length, const hr::Workspace &ws,
                     hipStream_t stream);
// the generated source (empty = unsupported shape); for tools and tests
std::string rtc_scan_source(const FusedPlanD &plan, int nd, int partBits);

}  // namespace ares
