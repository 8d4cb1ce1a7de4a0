/* ares_driver.h — C API of libaresdriver.so: the host side of the AQL batch pipeline in C++.
 *
 * The reference's host is Go (query/aql_processor.go, aql_batchexecutor.go, aql_context.go,
 * time_series_aggregate.go) and talks to libalgorithm.so / libmem.so through cgo.  No Go toolchain
 * exists in this environment, so the same host logic — same stage order, same ABI calls, same buffer
 * ownership — is written in C++ above the same C ABI, bound with dlopen() instead of cgo.  It is
 * backend-agnostic: it drives whatever (libalgorithm, libmem) pair it is given.
 *
 * The plan is handed over as a flat node array (what the AQL compiler's OOPK context holds as
 * expr trees, query/aql_context.go:148-150).
 */
#ifndef ARES_DRIVER_H_
#define ARES_DRIVER_H_

#include <stdint.h>

#include "ares_algorithm.h"

#ifdef __cplusplus
extern "C" {
#endif

enum AresNodeKind { ARES_NODE_COLUMN = 0, ARES_NODE_CONST_INT = 1, ARES_NODE_CONST_FLOAT = 2, ARES_NODE_UNARY = 3, ARES_NODE_BINARY = 4 };

typedef struct {
  int kind;     /* enum AresNodeKind */
  int op;       /* Unary/BinaryFunctorType */
  int lhs, rhs; /* child node indices */
  int table;    /* 0 = main table, k > 0 = foreign table k-1 */
  int column;   /* column index inside that table */
  int32_t ival;
  float fval;
  int outType;  /* enum DataType of the scratch vector when this is an inner node */
} AresPlanNode;

typedef struct {
  int joinColumn;            /* main-table column the join key comes from */
  CuckooHashIndex index;     /* device image of the primary-key index */
  int numColumns;
  int numBatches;
  const VectorPartySlice *slices; /* numColumns x numBatches, column-major */
  const int *dataTypes;      /* per column */
  int32_t baseBatchID;
  int32_t numRecordsInLastBatch;
} AresForeignTable;

/* geoIntersection (query/aql_context.go:327-353) once the shapes are on the device */
typedef struct {
  const uint8_t *shapeLatLongs; /* device: [lats f32][longs f32][shape index u8] of all polygon points */
  int numShapes;
  int totalNumPoints;
  int pointTable;  /* 0 = main table, k > 0 = foreign table k-1 */
  int pointColumn; /* column index inside that table */
  int inOrOut;
  int dimIndex;    /* query dimension that is the shape number (uint8), < 0: filter only */
} AresGeoIntersection;

typedef struct {
  const AresPlanNode *nodes;
  int numNodes;
  const int *filters; int numFilters;               /* root node of every main-table filter */
  const int *foreignFilters; int numForeignFilters; /* filters that read joined columns */
  const int *dimNodes; const int *dimTypes; int numDims; /* dimension roots + their output DataType */
  int measureNode;
  int aggFunc;      /* enum AggregateFunction */
  int measureType;  /* enum DataType of the measure vector */
  int useHashReduction;
  const AresForeignTable *foreignTables; int numForeignTables;
  /* when set (and the library exports AresFusedFilterHashReduce, include/ares_extensions.h) a batch
   * whose plan has the fusable shape is executed by ONE fused call instead of the per-node sequence;
   * any other plan, and any batch the library declines, runs the ordinary sequence */
  int useFusedExtension;
  const AresGeoIntersection *geo; /* NULL: no geo intersection in this query */
} AresQueryPlan;

typedef struct AresQuery AresQuery; /* oopkBatchContext + executor of one query on one device */

/* Loads the two libraries (dlopen, RTLD_LOCAL) and resolves every ABI symbol; NULL + message on failure. */
void *AresDriverOpen(const char *libalgorithmPath, const char *libmemPath, char *err, int errLen);
void AresDriverClose(void *driver);

AresQuery *AresQueryCreate(void *driver, const AresQueryPlan *plan, int device, void *stream, char *err, int errLen);
/* One batch through preExec / filter / join / project / reduce / postExec
 * (query/aql_batchexecutor.go:103-273).  columns: one slice per main-table column.  Returns 0, or
 * -1 with the library's error message in err. */
int AresQueryRunBatch(AresQuery *q, const VectorPartySlice *columns, int numColumns, int size,
                      uint32_t *baseCounts, uint32_t startRow, char *err, int errLen);
/* numBatches batches of numColumns slices each (batch-major), run like numBatches calls of AresQueryRunBatch (no base
 * counts, start row 0, isLastBatch = 0) without returning to the caller in between. */
int AresQueryRunResidentBatches(AresQuery *q, const VectorPartySlice *columns, int numColumns, const int *sizes, int numBatches,
                                char *err, int errLen);
int AresQueryResultSize(const AresQuery *q);
int AresQueryResultCapacity(const AresQuery *q);
uint8_t *AresQueryDimensionVector(const AresQuery *q); /* device pointer, capacity stride */
uint8_t *AresQueryMeasureVector(const AresQuery *q);
long AresQueryNumCalls(const AresQuery *q);            /* ABI calls issued so far */
long AresQueryNumFusedBatches(const AresQuery *q);     /* batches that took the fused extension */
/* D2H of the result (query/aql_processor.go:641-671): dims = for each dim in vector order
 * resultSize*width value bytes, then numDims x resultSize validity bytes; measures. */
int AresQueryFetch(AresQuery *q, uint8_t *dims, uint8_t *measures, char *err, int errLen);
/* HyperLogLog queries (aggFunc == AGGR_HLL, measure = GetHLLValue(column) into a Uint32 vector):
 * the executor must know which batch is the last one (query/aql_batchexecutor.go:62-100); after it,
 * the result is the dimension columns (AresQueryFetch, measures may be NULL), the registers per
 * dimension (uint16 x resultSize) and the encoded HLL vector (query/hll.go:52-63). */
void AresQuerySetLastBatch(AresQuery *q, int isLast);
/* The Go host owns two streams per query and swaps them after every batch
 * (query/aql_processor.go:66-67, :218, :247).  With a second stream set the driver does the same:
 * batch k's calls go to one stream, batch k+1's to the other; fetches use the current one. */
void AresQuerySetSecondStream(AresQuery *q, void *stream);
/* Device allocations (DeviceAllocate) holding the NEXT batch's columns: the driver releases them with
 * DeviceFree in cleanupBeforeAggregation — between project() and reduce() — exactly where the Go host
 * frees a batch's input columns (query/aql_processor.go:695-699).  Without this call the columns stay
 * the caller's (a device-resident column cache). */
void AresQueryAdoptColumns(AresQuery *q, void *const *allocations, int count);
int64_t AresQueryHLLVectorSize(const AresQuery *q);
int AresQueryFetchHLL(AresQuery *q, uint16_t *regCounts, uint8_t *hllVector, char *err, int errLen);
void AresQueryDestroy(AresQuery *q);

/* ---- shards across devices ------------------------------------------------------------
 * One process per GPU, one shard per process.  The only exchange is the merge of the per-shard group
 * tables: all-gather of their sizes, all-gather of the padded columnar partials, then every rank appends
 * them and re-reduces with the library's own HashReduce / Sort+Reduce — the reference's "previous
 * results + re-reduce" contract (query/aql_batchexecutor.go:236-251) with the aggregate's own combine
 * rule (broker/result_merge.go:77-94).  Key sets differ per shard: there is no element-wise all-reduce.
 *
 * The collective is pluggable: AresCommCreateRccl binds RCCL (librccl.so is dlopen()ed; the all-gathers
 * run on the query's stream, over xGMI between the GPUs of a node); AresCommCreate takes any all-gather
 * function — the multi-process CPU tests pass one built on torch.distributed / gloo, so the same C++
 * merge runs there on host memory. */
typedef int (*AresAllGatherFn)(void *user, const void *send, void *recv, size_t bytesPerRank, void *stream);
typedef struct AresComm AresComm;
AresComm *AresCommCreate(int rank, int nranks, AresAllGatherFn allGather, void *user);
/* RCCL: rank 0 obtains an id with AresCommRcclUniqueId and hands its 128 bytes to every rank (any
 * side channel: torch.distributed broadcast, a file, MPI).  NULL + message on failure. */
int AresCommRcclUniqueId(uint8_t id[128], char *err, int errLen);
AresComm *AresCommCreateRccl(const uint8_t id[128], int rank, int nranks, int device, char *err, int errLen);
/* Ranks as threads of ONE process — the reference's process model (one device per shard inside the server,
 * query/device_manager.go:185-218): out[0 .. nranks) receive one communicator per rank, to be used by nranks threads
 * at once.  A rank copies its peers' blocks itself: hipMemcpyAsync through unified addressing (peer-to-peer over
 * xGMI) on its own stream when deviceMemory != 0, memcpy otherwise.  0, or -1 + message. */
int AresCommCreateLocal(int nranks, int deviceMemory, AresComm **out, char *err, int errLen);
void AresCommDestroy(AresComm *c);
/* Replaces q's result by the merged result of all ranks (every rank ends with the whole table).
 * Hash-reduction and sort-reduction queries; HyperLogLog queries whose batches all ran with isLastBatch = 0 (the
 * merge exchanges the (group, register, rho) entries, keeps the maximum per register — broker/result_merge.go:95-104 —
 * and finalises).  0, or -1 + message. */
int AresQueryMergeShards(AresQuery *q, AresComm *c, char *err, int errLen);

/* Option (B), for group tables too large to replicate: every rank ends with the groups whose 64-bit row
 * hash falls into its 1/nranks share of the hash range (q's result becomes that share; *totalGroups = the
 * size of the whole result).  Blocks travel by one all-to-all — ncclSend / ncclRecv pairs on the query's
 * stream for an RCCL communicator, AresCommSetAllToAll for any other — or, without one, by nranks padded
 * all-gathers.  0, or -1 + message. */
typedef int (*AresAllToAllFn)(void *user, const void *send, const size_t *sendBytes, const size_t *sendOffsets, void *recv,
                              const size_t *recvBytes, const size_t *recvOffsets, void *stream);
void AresCommSetAllToAll(AresComm *c, AresAllToAllFn allToAll);
int AresQueryMergeShardsPartitioned(AresQuery *q, AresComm *c, int64_t *totalGroups, char *err, int errLen);

/* ---- host batches: transfer pipeline + device-resident column cache --------------------
 * The Go host uploads every batch's columns for every query (query/aql_processor.go:513-540,
 * :1345-1431) and overlaps the upload of batch k+1 with the execution of batch k on its second stream
 * (:850-881).  AresQueryRunHostBatches does the same from pinned host memory (HostAlloc): one
 * DeviceAllocate + AsyncCopyHostToDevice per column on the transfer stream, the previous batch executed
 * on a worker thread meanwhile, columns freed by the executor before its aggregation stage.
 * With a cache (AresColumnCacheCreate) columns stay in HBM under (table, batch, column) keys up to a
 * byte budget, least recently used first out: a query whose batches are cached uploads nothing —
 * MI355X's 288 GB hold hot batches resident.  stats: {bytes uploaded, uploads, cache hits, cache bytes}. */
typedef struct {
  const void *host;  /* pinned host image of the column allocation: [counts][validity][values] */
  size_t bytes;
  VectorPartySlice slice; /* offsets are relative to BasePtr, a byte offset into the host buffer (0 for a whole allocation) */
  uint64_t cacheKey;      /* identifies (table, batch, column) in the cache; 0 = never cache */
} AresHostColumn;
typedef struct AresColumnCache AresColumnCache;
AresColumnCache *AresColumnCacheCreate(void *driver, int device, size_t budgetBytes);
void AresColumnCacheDestroy(AresColumnCache *c);
int AresQueryRunHostBatches(AresQuery *q, const AresHostColumn *columns, int numColumns, const int *batchSizes,
                            int numBatches, AresColumnCache *cache, uint64_t stats[4], char *err, int errLen);

#ifdef __cplusplus
}
#endif
#endif /* ARES_DRIVER_H_ */
