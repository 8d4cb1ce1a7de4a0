/* ares_extensions.h — entry points of the MI355X libalgorithm.so that are NOT part of the
 * reference ABI.  A host that only knows query/time_series_aggregate.h never needs them.
 */
#ifndef ARES_EXTENSIONS_H_
#define ARES_EXTENSIONS_H_

#include <stddef.h>

#include "ares_algorithm.h"

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

/* Cross-call fusion inside the unchanged ABI.  Root transforms of the hot shape (a 4-byte column,
 * optionally combined with a constant, written to a dimension vector or a measure vector) are not
 * launched one by one: libalgorithm.so keeps up to 8 of them per (device, stream) and runs them as
 * ONE kernel that reads the index vector once.  Nothing observable changes: every other
 * libalgorithm.so entry point, and every libmem.so entry point through which the host could read,
 * overwrite or free device memory (WaitForCudaStream, AsyncCopy*, DeviceFree, DestroyCudaStream,
 * CudaProfilerStop), first launches what is pending.  libalgorithm.so finds the sibling libmem.so
 * (same directory) at run time and registers AresFlushDeferred through AresMemSetFlushHook; when
 * that is not possible (another allocator library in use) or ARES_DEFER=0 is set, every transform
 * is launched immediately as before. */
void AresFlushDeferred(int device);                      /* exported by libalgorithm.so */
void AresMemSetFlushHook(void (*hook)(int device));      /* exported by libmem.so */

/* Fused batch execution: filter -> dimension / measure projection -> hash reduction of ONE batch in
 * a single pass over the source columns, without the index / predicate / dimension vectors the
 * one-call-per-AST-node ABI materialises in between (~145 B/row of HBM traffic on
 * BASELINE config C3; this path moves ~53 B/row).  It replaces, for one batch, the call sequence
 *   InitIndexVector, BinaryFilter x numFilters, Unary/BinaryTransform x (numDims + 1), HashReduce
 * of query/aql_batchexecutor.go:103-273 and produces the same groups and aggregates.
 *
 * Shapes accepted (anything else returns an error string starting with "not fusable", and the host
 * simply runs the ordinary sequence): every expression is a main-table VectorPartyInput of a
 * 4-byte type (modes 1/2), bare (arity 1, Noop) or combined with a ConstantInput by a binary
 * functor (arity 2); filters are comparison functors (ANDed); dimensions are stored as
 * Int32 / Uint32 / Float32; the aggregate is one HashReduce supports with native LDS atomics.
 * prevKeys / prevValues hold the prevSize groups accumulated so far (the reference keeps them as the
 * first rows of the input vectors, query/aql_processor.go:752-774); outKeys.VectorCapacity must be
 * at least prevSize + batchRows.  res = number of groups written to outKeys / outValues. */
typedef struct {
  InputVector lhs, rhs; /* rhs is read only when arity == 2 */
  int arity;
  int functor; /* Unary/BinaryFunctorType */
  enum DataType outType;
} AresFusedExpr;

typedef struct {
  int numFilters;
  AresFusedExpr filters[4];
  int numDims;
  AresFusedExpr dims[4];
  AresFusedExpr measure; /* outType = data type of the measure vector */
  enum AggregateFunction aggFunc;
} AresFusedQuery;

CGoCallResHandle AresFusedFilterHashReduce(const AresFusedQuery *query, int batchRows, DimensionVector prevKeys,
                                           uint8_t *prevValues, int prevSize, DimensionVector outKeys,
                                           uint8_t *outValues, void *cudaStream, int device);

#ifdef __cplusplus
}
#endif
#endif /* ARES_EXTENSIONS_H_ */
