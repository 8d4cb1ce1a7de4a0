/* ares_extensions.h — entry points of the MI355X libalgorithm.so that are NOT part of the
 * reference ABI.  A host that only knows query/time_series_aggregate.h never needs them.
 */
#ifndef ARES_EXTENSIONS_H_
#define ARES_EXTENSIONS_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-kernel timing with HIP events recorded on each launch's own stream.  Enable(1) clears the
 * log and starts recording, Enable(0) stops.  Report() resolves the events (the caller has already
 * synchronised its streams) and writes one "kernel launches total_ms" line per kernel name into
 * buf; it returns the buffer size needed.  The Go host's counterpart is the per-stage wall timing
 * of query/stats.go:160-169, which must synchronise the stream after every stage instead. */
void AresProfilerEnable(int on);
size_t AresProfilerReport(char *buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* ARES_EXTENSIONS_H_ */
