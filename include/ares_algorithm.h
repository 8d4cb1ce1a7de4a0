/* ares_algorithm.h — C ABI of libalgorithm.so, the AQL batch-execution library.
 *
 * Drop-in for the reference's query/time_series_aggregate.h (bound from Go by
 * query/time_series_aggregate.go:17-19 with `-lalgorithm`).  The Go host keeps compiling
 * against the reference's own header; this header declares the very same types and the very
 * same 14 entry points so that the MI355X implementation (aresdb_amd/csrc/algo, hand-written
 * HIP for gfx950) is link-compatible.  Struct layouts are pinned by the static assertions at
 * the bottom (sizes measured on the reference header).
 *
 * Conventions shared by all entry points (reference query/filter.cu:141-165):
 *   - the last two arguments are always `void *cudaStream, int device`; every call selects the
 *     device itself and enqueues work only on the given stream;
 *   - the CGoCallResHandle is returned by value; a C++ exception never crosses the boundary:
 *     it is turned into a strdup()'ed message in pStrErr;
 *   - entry points that return a count (filters, Reduce, HashReduce, Expand, HyperLogLog)
 *     have resolved that count when they return.
 */
#ifndef ARES_ALGORITHM_H_
#define ARES_ALGORITHM_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include "ares_cgo.h"

/* ---- limits (time_series_aggregate.h:33-47) ------------------------------------------- */
enum {
  MAX_FOREIGN_TABLES = 7,
  MAX_COLUMNS_OF_A_TABLE = 32,
  MAX_DIMENSIONS = 8,
  MAX_DIMENSION_BYTES = 32,
  MAX_MEASURES = 32,
  MAX_INSTRUCTIONS = 1024,
  HASH_BUCKET_SIZE = 8, /* slots per cuckoo bucket */
  HASH_STASH_SIZE = 4,  /* slots in the overflow stash */
  HLL_BITS = 14,
  HLL_DENSE_SIZE = 1 << HLL_BITS,
  HLL_DENSE_THRESHOLD = HLL_DENSE_SIZE / 4,
  NUM_DIM_WIDTH = 5, /* dimension widths 16, 8, 4, 2, 1 bytes, in that order */
};

/* ---- enums: values are positional and part of the ABI (time_series_aggregate.h:50-130) -- */
enum AggregateFunction {
  AGGR_SUM_UNSIGNED = 1,
  AGGR_SUM_SIGNED = 2,
  AGGR_SUM_FLOAT = 3,
  AGGR_MIN_UNSIGNED = 4,
  AGGR_MIN_SIGNED = 5,
  AGGR_MIN_FLOAT = 6,
  AGGR_MAX_UNSIGNED = 7,
  AGGR_MAX_SIGNED = 8,
  AGGR_MAX_FLOAT = 9,
  AGGR_HLL = 10,
  AGGR_AVG_FLOAT = 11,
};

enum DataType {
  Bool, Int8, Uint8, Int16, Uint16, Int32, Uint32, Float32,
  Int64, Uint64, Float64, GeoPoint, UUID,
};

enum ConstDataType { ConstInt, ConstFloat, ConstGeoPoint, ConstUUID };

enum UnaryFunctorType {
  Negate, Not, BitwiseNot, IsNull, IsNotNull, Noop,
  GetWeekStart, GetMonthStart, GetQuarterStart, GetYearStart,
  GetDayOfMonth, GetDayOfYear, GetMonthOfYear, GetQuarterOfYear,
  GetHLLValue, ArrayLength,
};

enum BinaryFunctorType {
  And, Or, Equal, NotEqual, LessThan, LessThanOrEqual, GreaterThan, GreaterThanOrEqual,
  Plus, Minus, Multiply, Divide, Mod, BitwiseAnd, BitwiseOr, BitwiseXor, Floor,
  ArrayContains, ArrayElementAt,
};

/* ---- plain-old-data carried across the boundary ----------------------------------------- */

/* time_series_aggregate.h:133-136 — {0,0} means "no match" after HashLookup. */
typedef struct {
  int32_t batchID;
  uint32_t index;
} RecordID;

/* time_series_aggregate.h:141-148 — device image of memstore/cuckoo_index.go.  Bucket layout:
 * [RecordID x8][signature u8 x8][key keyBytes x8]; numBuckets buckets followed by one stash. */
typedef struct {
  uint8_t *buckets;
  uint32_t seeds[4];
  int keyBytes;
  int numHashes;
  int numBuckets;
} CuckooHashIndex;

typedef struct { float Lat; float Long; } GeoPointT;   /* :151-154 */
typedef struct { uint64_t p1; uint64_t p2; } UUIDT;    /* :157-160 */

/* :162-173 */
typedef struct DefaultValue {
  bool HasDefault;
  union {
    bool BoolVal;
    int32_t Int32Val;
    uint32_t Uint32Val;
    float FloatVal;
    int64_t Int64Val;
    GeoPointT GeoPointVal;
    UUIDT UUIDVal;
  } Value;
} DefaultValue;

/* :177-198 — one column of one batch: a single allocation [counts][nulls][values].
 * BasePtr==NULL: mode 0 (constant DefaultValue).  ValuesOffset==0: mode 1 (values only,
 * all valid).  NullsOffset==0: mode 2 (validity bitmap at BasePtr).  Otherwise mode 3
 * (cumulative run counts at BasePtr, bitmap at NullsOffset, values at ValuesOffset). */
typedef struct {
  uint8_t *BasePtr;
  uint32_t NullsOffset;
  uint32_t ValuesOffset;
  uint8_t StartingIndex; /* bit offset (0..7) into bit-packed bool values / validity */
  enum DataType DataType;
  struct DefaultValue DefaultValue;
  uint32_t Length;
} VectorPartySlice;

/* :202-206 — intermediate AST result: values[n] then one validity byte per row at
 * Values + NullsOffset. */
typedef struct {
  uint8_t *Values;
  uint32_t NullsOffset;
  enum DataType DataType;
} ScratchSpaceVector;

/* :209-221 */
typedef struct {
  union {
    int32_t IntVal;
    float FloatVal;
    GeoPointT GeoPointVal;
    UUIDT UUIDVal;
  } Value;
  bool IsValid;
  enum ConstDataType DataType;
} ConstantVector;

/* :227-237 — a dimension-table column read through the RecordIDs produced by HashLookup.
 * `Batches` points to HOST memory that is only valid during the call. */
typedef struct {
  RecordID *RecordIDs;
  VectorPartySlice *Batches;
  int32_t BaseBatchID;
  int32_t NumBatches;
  int32_t NumRecordsInLastBatch;
  int16_t *const TimezoneLookup;
  int16_t TimezoneLookupSize;
  enum DataType DataType;
  struct DefaultValue DefaultValue;
} ForeignColumnVector;

/* :240-247 */
typedef struct {
  uint8_t *OffsetLengthVector;
  uint32_t ValueOffsetAdj;
  enum DataType DataType;
  uint32_t Length;
} ArrayVectorPartySlice;

/* :250-256 */
enum InputVectorType {
  VectorPartyInput, ScratchSpaceInput, ConstantInput, ForeignColumnInput, ArrayVectorPartyInput
};

/* :260-269 */
typedef struct {
  union {
    ConstantVector Constant;
    VectorPartySlice VP;
    ScratchSpaceVector ScratchSpace;
    ForeignColumnVector ForeignVP;
    ArrayVectorPartySlice ArrayVP;
  } Vector;
  enum InputVectorType Type;
} InputVector;

/* :277-283 — columnar group-by key store: for each dim (widths 16,8,4,2,1 in that order)
 * VectorCapacity * width value bytes, then numDims * VectorCapacity validity bytes. */
typedef struct {
  uint8_t *DimValues;
  uint64_t *HashValues;
  uint32_t *IndexVector;
  int VectorCapacity;
  uint8_t NumDimsPerDimWidth[NUM_DIM_WIDTH];
} DimensionVector;

/* :287-291 */
typedef struct {
  uint8_t *DimValues;
  uint8_t *DimNulls;
  enum DataType DataType;
} DimensionOutputVector;

/* :296-301 */
typedef struct {
  uint32_t *Values;
  enum DataType DataType;
  enum AggregateFunction AggFunc;
} MeasureOutputVector;

/* :304-308 */
enum OutputVectorType { ScratchSpaceOutput, MeasureOutput, DimensionOutput };

/* :312-319 */
typedef struct {
  union {
    ScratchSpaceVector ScratchSpace;
    DimensionOutputVector Dimension;
    MeasureOutputVector Measure;
  } Vector;
  enum OutputVectorType Type;
} OutputVector;

/* :390-402 / :405-414 — geofence inputs (geo entry points are exported for link
 * compatibility; see DESIGN.md "out of scope"). */
typedef struct {
  float *Lats;
  float *Longs;
  uint16_t NumPoints;
} GeoShape;

typedef struct {
  uint8_t *LatLongs;
  int32_t TotalNumPoints;
  uint8_t TotalWords;
} GeoShapeBatch;

#ifdef __cplusplus
extern "C" {
#endif

/* time_series_aggregate.h:432-436 — idx[i] = start + i. */
CGoCallResHandle InitIndexVector(uint32_t *indexVector, uint32_t start, int indexVectorLength,
                                 void *cudaStream, int device);

/* :440-448 — foreign-key column -> RecordID per surviving row (cuckoo probe). */
CGoCallResHandle HashLookup(InputVector input, RecordID *output, uint32_t *indexVector,
                            int indexVectorLength, uint32_t *baseCounts, uint32_t startCount,
                            CuckooHashIndex hashIndex, void *cudaStream, int device);

/* :455-465 */
CGoCallResHandle UnaryTransform(InputVector input, OutputVector output, uint32_t *indexVector,
                                int indexVectorLength, uint32_t *baseCounts, uint32_t startCount,
                                enum UnaryFunctorType functorType, void *cudaStream, int device);

/* :472-484 — evaluates the predicate into predicateVector and compacts indexVector (and the
 * numForeignTables RecordID vectors) in place, stably; res = surviving length. */
CGoCallResHandle UnaryFilter(InputVector input, uint32_t *indexVector, uint8_t *predicateVector,
                             int indexVectorLength, RecordID **recordIDVectors,
                             int numForeignTables, uint32_t *baseCounts, uint32_t startCount,
                             enum UnaryFunctorType functorType, void *cudaStream, int device);

/* :489-500 */
CGoCallResHandle BinaryTransform(InputVector lhs, InputVector rhs, OutputVector output,
                                 uint32_t *indexVector, int indexVectorLength,
                                 uint32_t *baseCounts, uint32_t startCount,
                                 enum BinaryFunctorType functorType, void *cudaStream, int device);

/* :503-516 */
CGoCallResHandle BinaryFilter(InputVector lhs, InputVector rhs, uint32_t *indexVector,
                              uint8_t *predicateVector, int indexVectorLength,
                              RecordID **recordIDVectors, int numForeignTables,
                              uint32_t *baseCounts, uint32_t startCount,
                              enum BinaryFunctorType functorType, void *cudaStream, int device);

/* :522-525 — keys.HashValues[i] = lo64(murmur3_x64_128(row keys.IndexVector[i])); then a
 * stable sort of keys.IndexVector by keys.HashValues. */
CGoCallResHandle Sort(DimensionVector keys, int length, void *cudaStream, int device);

/* :531-539 — segmented reduction over runs of equal hash; res = number of groups. */
CGoCallResHandle Reduce(DimensionVector inputKeys, uint8_t *inputValues,
                        DimensionVector outputKeys, uint8_t *outputValues, int valueBytes,
                        int length, enum AggregateFunction aggFunc, void *cudaStream, int device);

/* :546-554 — hash-table group-by (32-bit murmur3 of the dim row is the group identity);
 * res = number of groups; output order unspecified. */
CGoCallResHandle HashReduce(DimensionVector inputKeys, uint8_t *inputValues,
                            DimensionVector outputKeys, uint8_t *outputValues, int valueBytes,
                            int length, enum AggregateFunction aggFunc, void *cudaStream,
                            int device);

/* :567-574 — run-length expansion of dimension rows for non-aggregate queries. */
CGoCallResHandle Expand(DimensionVector inputKeys, DimensionVector outputKeys,
                        uint32_t *baseCounts, uint32_t *indexVector, int indexVectorLen,
                        int outputOccupiedLen, void *cudaStream, int device);

/* :585-596 */
CGoCallResHandle HyperLogLog(DimensionVector prevDimOut, DimensionVector curDimOut,
                             uint32_t *prevValuesOut, uint32_t *curValuesOut, int prevResultSize,
                             int curBatchSize, bool isLastBatch, uint8_t **hllVectorPtr,
                             size_t *hllVectorSizePtr, uint16_t **hllDimRegIDCountPtr,
                             void *cudaStream, int device);

/* :603-607 */
CGoCallResHandle GeoBatchIntersects(GeoShapeBatch geoShapeBatch, InputVector points,
                                    uint32_t *indexVector, int indexVectorLength,
                                    uint32_t startCount, RecordID **recordIDVectors,
                                    int numForeignTables, uint32_t *outputPredicate, bool inOrOut,
                                    void *cudaStream, int device);

/* :616-619 */
CGoCallResHandle WriteGeoShapeDim(int shapeTotalWords, DimensionOutputVector dimOut,
                                  int indexVectorLengthBeforeGeo, uint32_t *outputPredicate,
                                  void *cudaStream, int device);

/* :623 — one-time per-process device initialisation. */
CGoCallResHandle BootstrapDevice(void);

#ifdef __cplusplus
}
#endif

/* ---- ABI layout pins (x86-64 SysV; sizes/offsets measured on the reference header) ------ */
#ifdef __cplusplus
#define ARES_ABI_ASSERT(c, m) static_assert(c, m)
#else
#define ARES_ABI_ASSERT(c, m) _Static_assert(c, m)
#endif
ARES_ABI_ASSERT(sizeof(CGoCallResHandle) == 16, "CGoCallResHandle");
ARES_ABI_ASSERT(sizeof(RecordID) == 8, "RecordID");
ARES_ABI_ASSERT(sizeof(CuckooHashIndex) == 40, "CuckooHashIndex");
ARES_ABI_ASSERT(sizeof(DefaultValue) == 24, "DefaultValue");
ARES_ABI_ASSERT(sizeof(VectorPartySlice) == 56, "VectorPartySlice");
ARES_ABI_ASSERT(offsetof(VectorPartySlice, DataType) == 20, "VectorPartySlice.DataType");
ARES_ABI_ASSERT(offsetof(VectorPartySlice, DefaultValue) == 24, "VectorPartySlice.DefaultValue");
ARES_ABI_ASSERT(offsetof(VectorPartySlice, Length) == 48, "VectorPartySlice.Length");
ARES_ABI_ASSERT(sizeof(ScratchSpaceVector) == 16, "ScratchSpaceVector");
ARES_ABI_ASSERT(sizeof(ConstantVector) == 24, "ConstantVector");
ARES_ABI_ASSERT(offsetof(ConstantVector, IsValid) == 16, "ConstantVector.IsValid");
ARES_ABI_ASSERT(sizeof(ForeignColumnVector) == 72, "ForeignColumnVector");
ARES_ABI_ASSERT(offsetof(ForeignColumnVector, DataType) == 44, "ForeignColumnVector.DataType");
ARES_ABI_ASSERT(sizeof(ArrayVectorPartySlice) == 24, "ArrayVectorPartySlice");
ARES_ABI_ASSERT(sizeof(InputVector) == 80, "InputVector");
ARES_ABI_ASSERT(offsetof(InputVector, Type) == 72, "InputVector.Type");
ARES_ABI_ASSERT(sizeof(DimensionVector) == 40, "DimensionVector");
ARES_ABI_ASSERT(offsetof(DimensionVector, NumDimsPerDimWidth) == 28, "DimensionVector.NumDims");
ARES_ABI_ASSERT(sizeof(DimensionOutputVector) == 24, "DimensionOutputVector");
ARES_ABI_ASSERT(sizeof(MeasureOutputVector) == 16, "MeasureOutputVector");
ARES_ABI_ASSERT(sizeof(OutputVector) == 32, "OutputVector");
ARES_ABI_ASSERT(offsetof(OutputVector, Type) == 24, "OutputVector.Type");
ARES_ABI_ASSERT(sizeof(GeoShapeBatch) == 16, "GeoShapeBatch");

#endif /* ARES_ALGORITHM_H_ */
