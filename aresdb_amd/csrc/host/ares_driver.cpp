// libaresdriver.so — C++ mirror of the reference's Go batch executor above the cgo C ABI.
//
//   AbiLibrary         <- cgo bindings of query/time_series_aggregate.go:17-19 + cgoutils/memory.go:17-19
//                         and the DoCGoCall error convention (cgoutils/utils.go:26-34)
//   BatchContext       <- oopkBatchContext (query/aql_context.go, query/aql_processor.go:690-804)
//   processExpression  <- query/time_series_aggregate.go:491-593 (AST walk, one ABI call per node)
//   BatchExecutor      <- BatchExecutorImpl (query/aql_batchexecutor.go:103-273)
//
// Stage order, ABI calls, buffer ownership (the host allocates and frees every buffer through
// libmem, the result buffers grow by 12.5 % and are double-buffered) follow the Go code line by
// line; only the language differs.
#include "ares_driver.h"

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <map>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "ares_extensions.h"
#include "ares_memory.h"

namespace {

struct AbiError : std::runtime_error {
  explicit AbiError(const std::string &m) : std::runtime_error(m) {}
};

// DoCGoCall: a non-null pStrErr is the callee's malloc'ed message; the Go host panics with it
intptr_t check(CGoCallResHandle h) {
  if (h.pStrErr) {
    std::string msg(h.pStrErr);
    free(const_cast<char *>(h.pStrErr));
    throw AbiError(msg);
  }
  return reinterpret_cast<intptr_t>(h.res);
}

struct AbiLibrary {
  void *algoHandle = nullptr, *memHandle = nullptr;
  // libalgorithm
  decltype(&::InitIndexVector) InitIndexVector;
  decltype(&::HashLookup) HashLookup;
  decltype(&::UnaryTransform) UnaryTransform;
  decltype(&::UnaryFilter) UnaryFilter;
  decltype(&::BinaryTransform) BinaryTransform;
  decltype(&::BinaryFilter) BinaryFilter;
  decltype(&::Sort) Sort;
  decltype(&::Reduce) Reduce;
  decltype(&::HashReduce) HashReduce;
  decltype(&::AresFusedFilterHashReduce) FusedFilterHashReduce = nullptr;  // optional extension
  // libmem
  decltype(&::DeviceAllocate) DeviceAllocate;
  decltype(&::DeviceFree) DeviceFree;
  decltype(&::WaitForCudaStream) WaitForCudaStream;
  decltype(&::HyperLogLog) HyperLogLog;
  decltype(&::GeoBatchIntersects) GeoBatchIntersects;
  decltype(&::WriteGeoShapeDim) WriteGeoShapeDim;
  decltype(&::AsyncCopyDeviceToDevice) AsyncCopyDeviceToDevice;
  decltype(&::AsyncCopyDeviceToHost) AsyncCopyDeviceToHost;
  decltype(&::AsyncCopyHostToDevice) AsyncCopyHostToDevice;
  // optional (include/ares_extensions.h): libmem.so clears a freed block only where something wrote, so
  // whatever writes DeviceAllocate memory behind its back — a collective receiving into it — says so
  void (*NoteWrite)(int, const void *, size_t) = nullptr;
  // ... and whoever submits work to the query's streams behind its back says that too: frees that follow one another with
  // nothing submitted in between share their fence events (a collective is such a submission)
  void (*NoteActivity)() = nullptr;
  void noteWrite(int device, const void *p, size_t bytes) const {
    if (NoteWrite) NoteWrite(device, p, bytes);
    if (NoteActivity) NoteActivity();
  }

  template <typename F>
  void bind(void *handle, const char *name, F &fn) {
    fn = reinterpret_cast<F>(dlsym(handle, name));
    if (!fn) throw AbiError(std::string("missing symbol ") + name);
  }

  AbiLibrary(const char *algoPath, const char *memPath) {
    memHandle = dlopen(memPath, RTLD_NOW | RTLD_LOCAL);
    if (!memHandle) throw AbiError(std::string("dlopen ") + memPath + ": " + dlerror());
    algoHandle = dlopen(algoPath, RTLD_NOW | RTLD_LOCAL);
    if (!algoHandle) throw AbiError(std::string("dlopen ") + algoPath + ": " + dlerror());
    bind(algoHandle, "InitIndexVector", InitIndexVector);
    bind(algoHandle, "HashLookup", HashLookup);
    bind(algoHandle, "UnaryTransform", UnaryTransform);
    bind(algoHandle, "UnaryFilter", UnaryFilter);
    bind(algoHandle, "BinaryTransform", BinaryTransform);
    bind(algoHandle, "BinaryFilter", BinaryFilter);
    bind(algoHandle, "Sort", Sort);
    bind(algoHandle, "Reduce", Reduce);
    bind(algoHandle, "HashReduce", HashReduce);
    bind(algoHandle, "HyperLogLog", HyperLogLog);
    bind(algoHandle, "GeoBatchIntersects", GeoBatchIntersects);
    bind(algoHandle, "WriteGeoShapeDim", WriteGeoShapeDim);
    FusedFilterHashReduce = reinterpret_cast<decltype(FusedFilterHashReduce)>(dlsym(algoHandle, "AresFusedFilterHashReduce"));
    bind(memHandle, "DeviceAllocate", DeviceAllocate);
    bind(memHandle, "DeviceFree", DeviceFree);
    bind(memHandle, "WaitForCudaStream", WaitForCudaStream);
    bind(memHandle, "AsyncCopyDeviceToDevice", AsyncCopyDeviceToDevice);
    bind(memHandle, "AsyncCopyDeviceToHost", AsyncCopyDeviceToHost);
    bind(memHandle, "AsyncCopyHostToDevice", AsyncCopyHostToDevice);
    NoteWrite = reinterpret_cast<decltype(NoteWrite)>(dlsym(memHandle, "AresMemNoteWrite"));
    NoteActivity = reinterpret_cast<decltype(NoteActivity)>(dlsym(memHandle, "AresMemNoteActivity"));
  }
  ~AbiLibrary() {
    if (algoHandle) dlclose(algoHandle);
    if (memHandle) dlclose(memHandle);
  }
};

// The reference header gives ForeignColumnVector a `*const` member, which makes InputVector neither
// default-constructible nor assignable in C++; a zeroed byte box with the same layout is.
template <typename T>
struct Pod {
  alignas(T) unsigned char bytes[sizeof(T)];
  Pod() { memset(bytes, 0, sizeof(bytes)); }
  T &operator*() { return *reinterpret_cast<T *>(bytes); }
  T *operator->() { return reinterpret_cast<T *>(bytes); }
  const T &get() const { return *reinterpret_cast<const T *>(bytes); }
};
using IV = Pod<InputVector>;
using OV = Pod<OutputVector>;

constexpr int kDimWidths[NUM_DIM_WIDTH] = {16, 8, 4, 2, 1};

int data_type_bytes(int t) {
  switch (t) {
    case Bool: case Int8: case Uint8: return 1;
    case Int16: case Uint16: return 2;
    case Int32: case Uint32: case Float32: return 4;
    case Int64: case Uint64: case Float64: case GeoPoint: return 8;
    case UUID: return 16;
    default: throw AbiError("unknown data type");
  }
}

// query/common/dim_util.go: (value offset, validity offset) of dimension `dimIndex`
void dimension_start_offsets(const uint8_t ndw[NUM_DIM_WIDTH], int dimIndex, int64_t capacity, int64_t *valueOff,
                             int64_t *nullOff) {
  int64_t before = 0, all = 0;
  int d = 0, total = 0;
  for (int w = 0; w < NUM_DIM_WIDTH; w++) total += ndw[w];
  for (int w = 0; w < NUM_DIM_WIDTH; w++)
    for (int j = 0; j < ndw[w]; j++, d++) {
      if (d < dimIndex) before += kDimWidths[w];
      all += kDimWidths[w];
    }
  *valueOff = before * capacity;
  *nullOff = all * capacity + static_cast<int64_t>(dimIndex) * capacity;
  (void)total;
}

struct Plan {
  std::vector<AresPlanNode> nodes;
  std::vector<int> filters, foreignFilters, dimNodes, dimTypes;
  int measureNode, aggFunc, measureType;
  bool useHashReduction;
  bool useFusedExtension;
  struct Foreign {
    AresForeignTable t;
    std::vector<VectorPartySlice> slices;
    std::vector<int> dataTypes;
  };
  std::vector<Foreign> foreign;
  bool hasGeo = false;
  AresGeoIntersection geo;

  explicit Plan(const AresQueryPlan &p)
      : nodes(p.nodes, p.nodes + p.numNodes), filters(p.filters, p.filters + p.numFilters),
        foreignFilters(p.foreignFilters, p.foreignFilters + p.numForeignFilters),
        dimNodes(p.dimNodes, p.dimNodes + p.numDims), dimTypes(p.dimTypes, p.dimTypes + p.numDims),
        measureNode(p.measureNode), aggFunc(p.aggFunc), measureType(p.measureType),
        useHashReduction(p.useHashReduction != 0), useFusedExtension(p.useFusedExtension != 0) {
    for (int i = 0; i < p.numForeignTables; i++) {
      Foreign f;
      f.t = p.foreignTables[i];
      f.slices.assign(f.t.slices, f.t.slices + static_cast<size_t>(f.t.numColumns) * f.t.numBatches);
      f.dataTypes.assign(f.t.dataTypes, f.t.dataTypes + f.t.numColumns);
      foreign.push_back(std::move(f));
    }
    memset(&geo, 0, sizeof(geo));
    if (p.geo) {
      hasGeo = true;
      geo = *p.geo;
    }
  }
  int measureBytes() const { return data_type_bytes(measureType); }
  bool isHLL() const { return aggFunc == AGGR_HLL; }  // OOPKContext.IsHLL, query/aql_context.go:421-424
  int dimRowBytes() const {
    int b = 0;
    for (int t : dimTypes) b += data_type_bytes(t) + 1;
    return b;
  }
};

}  // namespace

struct AresQuery {
  AbiLibrary *lib;
  Plan plan;
  int device;
  void *stream;
  // the Go host owns two streams per query and swaps them after every batch
  // (query/aql_processor.go:66-67, :218, :247): batch k's calls go to one, batch k+1's to the other
  void *otherStream = nullptr;
  // oopkBatchContext
  uint8_t ndw[NUM_DIM_WIDTH] = {0, 0, 0, 0, 0};
  std::vector<int> dimVectorIndex;  // query dimension -> position in the width-ordered vector
  int resultSize = 0, resultCapacity = 0;
  uint8_t *dimVec[2] = {nullptr, nullptr};
  uint8_t *measureVec[2] = {nullptr, nullptr};
  // HyperLogLog queries (query/aql_context.go:289-294): allocated by the library on the last batch
  uint8_t *hllVector = nullptr;
  uint16_t *hllDimRegIDCount = nullptr;
  size_t hllVectorSize = 0;
  bool isLastBatch = false;
  uint32_t *geoPredicateVec = nullptr;
  int sizeBeforeGeoFilter = 0;
  std::vector<void *> ownedColumns;  // device allocations of the batch's columns handed over by the caller
  uint64_t *hashVec[2] = {nullptr, nullptr};
  uint32_t *dimIndexVec[2] = {nullptr, nullptr};
  int size = 0;
  uint32_t *indexVec = nullptr;
  uint8_t *predVec = nullptr;
  uint32_t *baseCounts = nullptr;
  uint32_t startRow = 0;
  const VectorPartySlice *columns = nullptr;
  int numColumns = 0;
  std::vector<uint8_t *> stack;
  std::vector<RecordID *> foreignRids;
  long calls = 0, fusedBatches = 0;
  bool fusedDeclined = false;  // the library rejected the plan once: stop asking

  AresQuery(AbiLibrary *l, const AresQueryPlan &p, int dev, void *s) : lib(l), plan(p), device(dev), stream(s) {
    // query/aql_compiler.go:1341-1362: dimensions ordered by width 16..1, then query order
    const int n = static_cast<int>(plan.dimTypes.size());
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      return data_type_bytes(plan.dimTypes[a]) > data_type_bytes(plan.dimTypes[b]);
    });
    dimVectorIndex.assign(n, 0);
    for (int p2 = 0; p2 < n; p2++) dimVectorIndex[order[p2]] = p2;
    for (int t : plan.dimTypes) {
      const int w = data_type_bytes(t);
      for (int k = 0; k < NUM_DIM_WIDTH; k++)
        if (kDimWidths[k] == w) ndw[k]++;
    }
  }

  // -- device_allocator.go semantics: every byte is allocated and freed by the host through libmem --
  template <typename T = uint8_t>
  T *alloc(size_t bytes) {
    return reinterpret_cast<T *>(check(lib->DeviceAllocate(bytes ? bytes : 1, device)));
  }
  void release(void *p) {
    if (p) check(lib->DeviceFree(p, device));
  }
  void wait() { check(lib->WaitForCudaStream(stream, device)); }
  void d2d(void *dst, void *src, size_t bytes) {
    if (bytes) check(lib->AsyncCopyDeviceToDevice(dst, src, bytes, stream, device));
  }

  // -- aql_processor.go:726-739 --
  void prepareForFiltering(const VectorPartySlice *cols, int ncols, int n, uint32_t *bc, uint32_t start) {
    columns = cols;
    numColumns = ncols;
    size = n;
    baseCounts = bc;
    startRow = start;
    indexVec = alloc<uint32_t>(static_cast<size_t>(n) * 4);
    predVec = alloc<uint8_t>(static_cast<size_t>(n));
  }

  // -- aql_processor.go:743-804: grow the result buffers by 12.5 % and carry the old results over --
  void prepareForDimAndMeasureEval() {
    if (resultSize + size <= resultCapacity) return;
    const int oldCapacity = resultCapacity;
    resultCapacity = resultSize + size;
    resultCapacity += resultCapacity / 8;
    const int64_t cap = resultCapacity;
    std::vector<int> widths;
    for (int k = 0; k < NUM_DIM_WIDTH; k++)
      for (int j = 0; j < ndw[k]; j++) widths.push_back(kDimWidths[k]);
    {
      uint8_t *old0 = dimVec[0], *old1 = dimVec[1];
      const size_t unit = std::max(plan.dimRowBytes(), 1);
      dimVec[0] = alloc(cap * unit);
      dimVec[1] = alloc(cap * unit);
      if (old0 && resultSize) {  // asyncCopyDimensionVector: per-dimension strided D2D copies
        for (size_t d = 0; d < widths.size(); d++) {
          int64_t nv, nn, ov, on;
          dimension_start_offsets(ndw, static_cast<int>(d), cap, &nv, &nn);
          dimension_start_offsets(ndw, static_cast<int>(d), oldCapacity, &ov, &on);
          d2d(dimVec[0] + nv, old0 + ov, static_cast<size_t>(resultSize) * widths[d]);
          d2d(dimVec[0] + nn, old0 + on, static_cast<size_t>(resultSize));
        }
      }
      if (old0 || old1) wait();
      release(old0);
      release(old1);
    }
    if (plan.isHLL() || !plan.useHashReduction) {
      release(dimIndexVec[0]); release(dimIndexVec[1]);
      dimIndexVec[0] = alloc<uint32_t>(cap * 4); dimIndexVec[1] = alloc<uint32_t>(cap * 4);
      uint64_t *old0 = hashVec[0], *old1 = hashVec[1];
      hashVec[0] = alloc<uint64_t>(cap * 8); hashVec[1] = alloc<uint64_t>(cap * 8);
      // HyperLogLog keeps the merged keys of the earlier batches in hash vector [0]
      if (plan.isHLL() && old0 && resultSize) {
        d2d(hashVec[0], old0, static_cast<size_t>(resultSize) * 8);
        wait();
      }
      release(old0); release(old1);
    }
    {
      const int mb = plan.measureBytes();
      uint8_t *old0 = measureVec[0], *old1 = measureVec[1];
      measureVec[0] = alloc(cap * mb);
      measureVec[1] = alloc(cap * mb);
      if (old0 && resultSize) d2d(measureVec[0], old0, static_cast<size_t>(resultSize) * mb);
      if (old0 || old1) wait();
      release(old0);
      release(old1);
    }
  }

  // -- time_series_aggregate.go:745-769 --
  uint8_t *allocateStackFrame(int dataType, uint32_t *nullsOffset) {
    const int w = (dataType == Int64 || dataType == Uint64 || dataType == Float64 || dataType == GeoPoint) ? 8
                  : dataType == UUID ? 16 : 4;
    uint8_t *values = alloc(static_cast<size_t>(w + 1) * size);
    stack.push_back(values);
    *nullsOffset = static_cast<uint32_t>(w) * size;
    return values;
  }
  void shrinkStackFrame() {
    std::swap(stack[stack.size() - 1], stack[stack.size() - 2]);
    release(stack.back());
    stack.pop_back();
  }

  void cleanupBeforeAggregation() {
    // the batch's input columns go first (query/aql_processor.go:695-699)
    for (void *p : ownedColumns) release(p);
    ownedColumns.clear();
    release(indexVec); indexVec = nullptr;
    release(predVec); predVec = nullptr;
    for (RecordID *p : foreignRids) release(p);
    foreignRids.clear();
    release(geoPredicateVec); geoPredicateVec = nullptr;
    for (uint8_t *p : stack) release(p);
    stack.clear();
    foreignSliceArrays.clear();
  }
  void swapResultBuffers() {
    size = 0;
    std::swap(dimVec[0], dimVec[1]);
    std::swap(measureVec[0], measureVec[1]);
    std::swap(hashVec[0], hashVec[1]);
  }
  void releaseAll() {
    for (int i = 0; i < 2; i++) {
      release(dimVec[i]); release(measureVec[i]); release(hashVec[i]); release(dimIndexVec[i]);
      dimVec[i] = nullptr; measureVec[i] = nullptr; hashVec[i] = nullptr; dimIndexVec[i] = nullptr;
    }
    release(hllVector); release(hllDimRegIDCount);
    hllVector = nullptr; hllDimRegIDCount = nullptr; hllVectorSize = 0;
    resultCapacity = 0;
  }

  // -- processExpression (time_series_aggregate.go:491-593) --
  std::vector<std::vector<VectorPartySlice>> foreignSliceArrays;  // Go slices handed to the callee for one call

  IV columnInput(const AresPlanNode &n) {
    IV iv;
    if (n.table == 0) {
      if (n.column < 0 || n.column >= numColumns) throw AbiError("column index out of range");
      iv->Vector.VP = columns[n.column];
      iv->Type = VectorPartyInput;
      return iv;
    }
    const Plan::Foreign &ft = plan.foreign.at(n.table - 1);
    foreignSliceArrays.emplace_back(ft.slices.begin() + static_cast<size_t>(n.column) * ft.t.numBatches,
                                    ft.slices.begin() + static_cast<size_t>(n.column + 1) * ft.t.numBatches);
    ForeignColumnVector &f = iv->Vector.ForeignVP;
    f.RecordIDs = foreignRids.at(n.table - 1);
    f.Batches = foreignSliceArrays.back().data();
    f.BaseBatchID = ft.t.baseBatchID;
    f.NumBatches = ft.t.numBatches;
    f.NumRecordsInLastBatch = ft.t.numRecordsInLastBatch;
    f.TimezoneLookupSize = 0;  // TimezoneLookup stays NULL (zeroed box)
    f.DataType = static_cast<DataType>(ft.dataTypes[n.column]);
    iv->Type = ForeignColumnInput;
    return iv;
  }
  static IV constantInput(const AresPlanNode &n) {
    IV iv;
    if (n.kind == ARES_NODE_CONST_FLOAT) {
      iv->Vector.Constant.Value.FloatVal = n.fval;
      iv->Vector.Constant.DataType = ConstFloat;
    } else {
      iv->Vector.Constant.Value.IntVal = n.ival;
      iv->Vector.Constant.DataType = ConstInt;
    }
    iv->Vector.Constant.IsValid = true;
    iv->Type = ConstantInput;
    return iv;
  }
  static IV scratchInput(uint8_t *values, uint32_t nullsOffset, int dataType) {
    IV iv;
    iv->Vector.ScratchSpace.Values = values;
    iv->Vector.ScratchSpace.NullsOffset = nullsOffset;
    iv->Vector.ScratchSpace.DataType = static_cast<DataType>(dataType);
    iv->Type = ScratchSpaceInput;
    return iv;
  }
  static OV scratchOutput(uint8_t *values, uint32_t nullsOffset, int dataType) {
    OV ov;
    ov->Vector.ScratchSpace.Values = values;
    ov->Vector.ScratchSpace.NullsOffset = nullsOffset;
    ov->Vector.ScratchSpace.DataType = static_cast<DataType>(dataType);
    ov->Type = ScratchSpaceOutput;
    return ov;
  }

  // root action: what to do with the operands of the expression's root
  struct Action {
    enum Kind { NONE, FILTER, MEASURE, DIMENSION } kind = NONE;
    int dimType = 0;
    int64_t valueOff = 0, nullOff = 0;
    int prevResultSize = 0;
  };

  void runAction(const Action &a, int functor, IV *in, int arity) {
    if (size <= 0) return;
    calls++;
    if (a.kind == Action::FILTER) {
      RecordID **vecs = foreignRids.empty() ? nullptr : foreignRids.data();
      const int nf = static_cast<int>(foreignRids.size());
      const intptr_t n =
          arity == 1 ? check(lib->UnaryFilter(in[0].get(), indexVec, predVec, size, vecs, nf, baseCounts, startRow,
                                              static_cast<UnaryFunctorType>(functor), stream, device))
                     : check(lib->BinaryFilter(in[0].get(), in[1].get(), indexVec, predVec, size, vecs, nf, baseCounts, startRow,
                                               static_cast<BinaryFunctorType>(functor), stream, device));
      size = static_cast<int>(n);
      return;
    }
    OV ov;
    if (a.kind == Action::MEASURE) {
      ov->Vector.Measure.Values = reinterpret_cast<uint32_t *>(measureVec[0] + static_cast<size_t>(resultSize) * plan.measureBytes());
      // hll values of the batch go to measure vector [1] (time_series_aggregate.go:404-408)
      if (plan.isHLL()) ov->Vector.Measure.Values = reinterpret_cast<uint32_t *>(measureVec[1]);
      ov->Vector.Measure.DataType = static_cast<DataType>(plan.measureType);
      ov->Vector.Measure.AggFunc = static_cast<AggregateFunction>(plan.aggFunc);
      ov->Type = MeasureOutput;
    } else {
      const int w = data_type_bytes(a.dimType);
      ov->Vector.Dimension.DimValues = dimVec[0] + a.valueOff + static_cast<int64_t>(w) * a.prevResultSize;
      ov->Vector.Dimension.DimNulls = dimVec[0] + a.nullOff + a.prevResultSize;
      ov->Vector.Dimension.DataType = static_cast<DataType>(a.dimType);
      ov->Type = DimensionOutput;
    }
    if (arity == 1)
      check(lib->UnaryTransform(in[0].get(), ov.get(), indexVec, size, baseCounts, startRow, static_cast<UnaryFunctorType>(functor),
                                stream, device));
    else
      check(lib->BinaryTransform(in[0].get(), in[1].get(), ov.get(), indexVec, size, baseCounts, startRow,
                                 static_cast<BinaryFunctorType>(functor), stream, device));
  }

  // returns the node's value as an InputVector (leaf or scratch frame) when action is NONE,
  // otherwise performs the root action
  IV processExpression(int nodeIdx, const Action &action) {
    const AresPlanNode &n = plan.nodes.at(nodeIdx);
    IV none;
    switch (n.kind) {
      case ARES_NODE_COLUMN:
      case ARES_NODE_CONST_INT:
      case ARES_NODE_CONST_FLOAT: {
        IV iv = n.kind == ARES_NODE_COLUMN ? columnInput(n) : constantInput(n);
        if (action.kind != Action::NONE) {
          runAction(action, Noop, &iv, 1);
          return none;
        }
        return iv;
      }
      case ARES_NODE_UNARY: {
        IV in = processExpression(n.lhs, Action());
        if (action.kind != Action::NONE) {
          runAction(action, n.op, &in, 1);
          return none;
        }
        uint32_t nullsOff;
        uint8_t *values = allocateStackFrame(n.outType, &nullsOff);
        calls++;
        check(lib->UnaryTransform(in.get(), scratchOutput(values, nullsOff, n.outType).get(), indexVec, size, baseCounts,
                                  startRow, static_cast<UnaryFunctorType>(n.op), stream, device));
        if (in->Type == ScratchSpaceInput) shrinkStackFrame();
        return scratchInput(values, nullsOff, n.outType);
      }
      case ARES_NODE_BINARY: {
        IV in[2];
        in[0] = processExpression(n.lhs, Action());
        in[1] = processExpression(n.rhs, Action());
        if (action.kind != Action::NONE) {
          runAction(action, n.op, in, 2);
          return none;
        }
        uint32_t nullsOff;
        uint8_t *values = allocateStackFrame(n.outType, &nullsOff);
        calls++;
        check(lib->BinaryTransform(in[0].get(), in[1].get(), scratchOutput(values, nullsOff, n.outType).get(), indexVec, size,
                                   baseCounts, startRow, static_cast<BinaryFunctorType>(n.op), stream, device));
        if (in[1]->Type == ScratchSpaceInput) shrinkStackFrame();
        if (in[0]->Type == ScratchSpaceInput) shrinkStackFrame();
        return scratchInput(values, nullsOff, n.outType);
      }
      default:
        throw AbiError("unsupported expression node");
    }
  }

  DimensionVector dimensionVector(int which) const {
    DimensionVector dv;
    memset(&dv, 0, sizeof(dv));
    dv.DimValues = dimVec[which];
    dv.HashValues = hashVec[which];
    dv.IndexVector = dimIndexVec[which];
    dv.VectorCapacity = resultCapacity;
    memcpy(dv.NumDimsPerDimWidth, ndw, sizeof(ndw));
    return dv;
  }

  // ---- BatchExecutorImpl.Run (aql_batchexecutor.go:103-273) ----
  void preExec() {
    if (indexVec && size > 0) {
      calls++;
      check(lib->InitIndexVector(indexVec, 0, size, stream, device));
    }
  }
  void filter() {
    Action a;
    a.kind = Action::FILTER;
    for (int f : plan.filters) processExpression(f, a);
  }
  void join() {
    for (const Plan::Foreign &ft : plan.foreign) {
      RecordID *rids = alloc<RecordID>(static_cast<size_t>(std::max(size, 1)) * 8);
      foreignRids.push_back(rids);
      if (size > 0) {
        AresPlanNode key;
        memset(&key, 0, sizeof(key));
        key.kind = ARES_NODE_COLUMN;
        key.table = 0;
        key.column = ft.t.joinColumn;
        calls++;
        check(lib->HashLookup(columnInput(key).get(), rids, indexVec, size, baseCounts, startRow, ft.t.index, stream, device));
      }
    }
    Action a;
    a.kind = Action::FILTER;
    for (int f : plan.foreignFilters) processExpression(f, a);
    // geo intersection (query/aql_batchexecutor.go:146-165, query/time_series_aggregate.go:636-660)
    const int words = (plan.geo.numShapes + 31) / 32;
    if (plan.hasGeo) geoPredicateVec = alloc<uint32_t>(static_cast<size_t>(std::max(size, 1)) * 4 * words);
    sizeBeforeGeoFilter = size;
    if (plan.hasGeo && size > 0 && plan.geo.shapeLatLongs) {
      GeoShapeBatch shapes;
      shapes.LatLongs = const_cast<uint8_t *>(plan.geo.shapeLatLongs);
      shapes.TotalNumPoints = plan.geo.totalNumPoints;
      shapes.TotalWords = static_cast<uint8_t>(words);
      AresPlanNode pointNode;
      memset(&pointNode, 0, sizeof(pointNode));
      pointNode.kind = ARES_NODE_COLUMN;
      pointNode.table = plan.geo.pointTable;
      pointNode.column = plan.geo.pointColumn;
      const int nf = static_cast<int>(foreignRids.size());
      calls++;
      size = static_cast<int>(check(lib->GeoBatchIntersects(shapes, columnInput(pointNode).get(), indexVec, size, startRow,
                                                            nf ? foreignRids.data() : nullptr, nf, geoPredicateVec,
                                                            plan.geo.inOrOut != 0, stream, device)));
    }
  }
  void project() {
    prepareForDimAndMeasureEval();
    const int prev = resultSize;
    for (size_t i = 0; i < plan.dimNodes.size(); i++) {
      Action a;
      a.kind = Action::DIMENSION;
      a.dimType = plan.dimTypes[i];
      a.prevResultSize = prev;
      dimension_start_offsets(ndw, dimVectorIndex[i], resultCapacity, &a.valueOff, &a.nullOff);
      if (plan.hasGeo && plan.geo.dimIndex == static_cast<int>(i)) {  // the shape number (time_series_aggregate.go:611-634)
        if (size > 0 && plan.geo.shapeLatLongs) {
          DimensionOutputVector dv;
          dv.DimValues = dimVec[0] + a.valueOff + prev;
          dv.DimNulls = dimVec[0] + a.nullOff + prev;
          dv.DataType = Uint8;
          calls++;
          check(lib->WriteGeoShapeDim((plan.geo.numShapes + 31) / 32, dv, sizeBeforeGeoFilter, geoPredicateVec, stream, device));
        }
        continue;
      }
      processExpression(plan.dimNodes[i], a);
    }
    Action m;
    m.kind = Action::MEASURE;
    processExpression(plan.measureNode, m);
    wait();
    cleanupBeforeAggregation();
  }
  void reduce() {
    const int length = resultSize + size;
    const int mb = plan.measureBytes();
    if (plan.isHLL()) {  // query/aql_batchexecutor.go:221-233, query/time_series_aggregate.go:661-680
      calls += 3;
      check(lib->InitIndexVector(dimIndexVec[0], 0, resultSize, stream, device));
      check(lib->InitIndexVector(dimIndexVec[1], static_cast<uint32_t>(resultSize), length, stream, device));
      uint8_t *vec = nullptr;
      uint16_t *counts = nullptr;
      size_t vecSize = 0;
      resultSize = static_cast<int>(check(lib->HyperLogLog(
          dimensionVector(0), dimensionVector(1), reinterpret_cast<uint32_t *>(measureVec[0]),
          reinterpret_cast<uint32_t *>(measureVec[1]), resultSize, size, isLastBatch, &vec, &vecSize, &counts, stream, device)));
      if (vec || counts) {
        release(hllVector); release(hllDimRegIDCount);
        hllVector = vec; hllDimRegIDCount = counts; hllVectorSize = vecSize;
      }
    } else if (plan.useHashReduction) {
      calls++;
      resultSize = static_cast<int>(check(lib->HashReduce(dimensionVector(0), measureVec[0], dimensionVector(1), measureVec[1],
                                                          mb, length, static_cast<AggregateFunction>(plan.aggFunc), stream,
                                                          device)));
    } else {
      calls += 3;
      check(lib->InitIndexVector(dimIndexVec[0], 0, length, stream, device));
      check(lib->Sort(dimensionVector(0), length, stream, device));
      resultSize = static_cast<int>(check(lib->Reduce(dimensionVector(0), measureVec[0], dimensionVector(1), measureVec[1], mb,
                                                      length, static_cast<AggregateFunction>(plan.aggFunc), stream, device)));
    }
    wait();
  }
  void postExec() { swapResultBuffers(); }
  void swapStreams() {
    if (otherStream) std::swap(stream, otherStream);
  }

  // ---- fused extension (include/ares_extensions.h): one call per batch --------------------------
  bool fusedExpr(int nodeIdx, int outType, AresFusedExpr *e) {
    const AresPlanNode &n = plan.nodes.at(nodeIdx);
    memset(e, 0, sizeof(*e));
    e->outType = static_cast<DataType>(outType);
    auto leafColumn = [&](const AresPlanNode &c, InputVector *iv) {
      if (c.kind != ARES_NODE_COLUMN || c.table != 0) return false;
      memcpy(iv, &columnInput(c).get(), sizeof(InputVector));
      return true;
    };
    if (n.kind == ARES_NODE_COLUMN) {
      e->arity = 1;
      e->functor = Noop;
      return leafColumn(n, &e->lhs);
    }
    if (n.kind != ARES_NODE_BINARY) return false;
    const AresPlanNode &l = plan.nodes.at(n.lhs), &r = plan.nodes.at(n.rhs);
    if (r.kind != ARES_NODE_CONST_INT && r.kind != ARES_NODE_CONST_FLOAT) return false;
    if (!leafColumn(l, &e->lhs)) return false;
    memcpy(&e->rhs, &constantInput(r).get(), sizeof(InputVector));
    e->arity = 2;
    e->functor = n.op;
    return true;
  }

  // returns false when this batch must take the ordinary sequence
  bool runBatchFused(const VectorPartySlice *cols, int ncols, int n) {
    if (!plan.useFusedExtension || !plan.useHashReduction || fusedDeclined || !lib->FusedFilterHashReduce) return false;
    if (plan.hasGeo || plan.isHLL()) return false;
    if (!plan.foreign.empty() || !plan.foreignFilters.empty() || baseCounts) return false;
    if (plan.dimNodes.empty() || plan.dimNodes.size() > 4 || plan.filters.size() > 4) return false;
    columns = cols;
    numColumns = ncols;
    Pod<AresFusedQuery> q;
    q->numFilters = static_cast<int>(plan.filters.size());
    for (size_t i = 0; i < plan.filters.size(); i++)
      if (!fusedExpr(plan.filters[i], Bool, &q->filters[i])) return false;
    // dimensions in vector order (all 4 bytes wide: vector order == query order)
    q->numDims = static_cast<int>(plan.dimNodes.size());
    for (size_t i = 0; i < plan.dimNodes.size(); i++) {
      if (data_type_bytes(plan.dimTypes[i]) != 4) return false;
      if (!fusedExpr(plan.dimNodes[i], plan.dimTypes[i], &q->dims[dimVectorIndex[i]])) return false;
    }
    if (!fusedExpr(plan.measureNode, plan.measureType, &q->measure)) return false;
    q->aggFunc = static_cast<AggregateFunction>(plan.aggFunc);
    size = n;
    prepareForDimAndMeasureEval();  // result buffers sized for resultSize + n, previous results carried over
    calls++;
    CGoCallResHandle h = lib->FusedFilterHashReduce(&q.get(), n, dimensionVector(0), measureVec[0], resultSize,
                                                    dimensionVector(1), measureVec[1], stream, device);
    if (h.pStrErr && strncmp(h.pStrErr, "not fusable", 11) == 0) {
      free(const_cast<char *>(h.pStrErr));
      fusedDeclined = true;
      size = 0;
      return false;
    }
    resultSize = static_cast<int>(check(h));
    wait();
    for (void *p : ownedColumns) release(p);  // the batch's input columns (query/aql_processor.go:695-699)
    ownedColumns.clear();
    swapResultBuffers();
    fusedBatches++;
    return true;
  }

  void runBatch(const VectorPartySlice *cols, int ncols, int n, uint32_t *bc, uint32_t start) {
    baseCounts = bc;
    if (runBatchFused(cols, ncols, n)) {
      swapStreams();
      return;
    }
    prepareForFiltering(cols, ncols, n, bc, start);
    preExec();
    filter();
    join();
    project();
    reduce();
    postExec();
    swapStreams();
  }
};

// ---- C API ---------------------------------------------------------------------------------------
namespace {
void set_err(char *err, int errLen, const char *msg) {
  if (err && errLen > 0) {
    strncpy(err, msg, static_cast<size_t>(errLen) - 1);
    err[errLen - 1] = 0;
  }
}
}  // namespace

extern "C" {

void *AresDriverOpen(const char *libalgorithmPath, const char *libmemPath, char *err, int errLen) {
  try {
    return new AbiLibrary(libalgorithmPath, libmemPath);
  } catch (std::exception &e) {
    set_err(err, errLen, e.what());
    return nullptr;
  }
}

void AresDriverClose(void *driver) { delete static_cast<AbiLibrary *>(driver); }

AresQuery *AresQueryCreate(void *driver, const AresQueryPlan *plan, int device, void *stream, char *err, int errLen) {
  try {
    return new AresQuery(static_cast<AbiLibrary *>(driver), *plan, device, stream);
  } catch (std::exception &e) {
    set_err(err, errLen, e.what());
    return nullptr;
  }
}

int AresQueryRunBatch(AresQuery *q, const VectorPartySlice *columns, int numColumns, int size, uint32_t *baseCounts,
                      uint32_t startRow, char *err, int errLen) {
  try {
    q->runBatch(columns, numColumns, size, baseCounts, startRow);
    return 0;
  } catch (std::exception &e) {
    // the Go host recovers the panic, frees every device buffer and fails the query
    // (query/aql_processor.go:50-64, :263-269)
    set_err(err, errLen, e.what());
    try {
      q->cleanupBeforeAggregation();
    } catch (...) {
    }
    return -1;
  }
}

// Every batch of a shard that is resident on the device, one after the other like ProcessQuery's batch loop
// (query/aql_processor.go:138-161): the host side of a whole query in ONE call, so that a harness written in an
// interpreted language adds nothing between batches (bench.py's timed region: one call per step).
int AresQueryRunResidentBatches(AresQuery *q, const VectorPartySlice *columns, int numColumns, const int *sizes, int numBatches,
                                char *err, int errLen) {
  for (int b = 0; b < numBatches; b++) {
    q->ownedColumns.clear();
    q->isLastBatch = false;
    const int rc = AresQueryRunBatch(q, columns + static_cast<size_t>(b) * numColumns, numColumns, sizes[b], nullptr, 0, err, errLen);
    if (rc != 0) return rc;
  }
  return 0;
}

int AresQueryResultSize(const AresQuery *q) { return q->resultSize; }
int AresQueryResultCapacity(const AresQuery *q) { return q->resultCapacity; }
uint8_t *AresQueryDimensionVector(const AresQuery *q) { return q->dimVec[0]; }
uint8_t *AresQueryMeasureVector(const AresQuery *q) { return q->measureVec[0]; }
long AresQueryNumCalls(const AresQuery *q) { return q->calls; }
long AresQueryNumFusedBatches(const AresQuery *q) { return q->fusedBatches; }
void AresQuerySetLastBatch(AresQuery *q, int isLast) { q->isLastBatch = isLast != 0; }
void AresQuerySetSecondStream(AresQuery *q, void *stream) { q->otherStream = stream; }
void AresQueryAdoptColumns(AresQuery *q, void *const *allocations, int count) {
  q->ownedColumns.assign(allocations, allocations + count);
}
int64_t AresQueryHLLVectorSize(const AresQuery *q) { return static_cast<int64_t>(q->hllVectorSize); }

int AresQueryFetchHLL(AresQuery *q, uint16_t *regCounts, uint8_t *hllVector, char *err, int errLen) {
  try {
    const int n = q->resultSize;
    if (n && q->hllDimRegIDCount)
      check(q->lib->AsyncCopyDeviceToHost(regCounts, q->hllDimRegIDCount, static_cast<size_t>(n) * 2, q->stream, q->device));
    if (n && q->hllVector)
      check(q->lib->AsyncCopyDeviceToHost(hllVector, q->hllVector, q->hllVectorSize, q->stream, q->device));
    q->wait();
    return 0;
  } catch (std::exception &e) {
    set_err(err, errLen, e.what());
    return -1;
  }
}

int AresQueryFetch(AresQuery *q, uint8_t *dims, uint8_t *measures, char *err, int errLen) {
  try {
    const int n = q->resultSize;
    std::vector<int> widths;
    for (int k = 0; k < NUM_DIM_WIDTH; k++)
      for (int j = 0; j < q->ndw[k]; j++) widths.push_back(kDimWidths[k]);
    uint8_t *out = dims;
    // asyncCopyDimensionVector to the host (query/aql_processor.go:641-671): the host layout uses
    // resultSize as its stride
    for (size_t d = 0; d < widths.size() && n; d++) {
      int64_t vo, no;
      dimension_start_offsets(q->ndw, static_cast<int>(d), q->resultCapacity, &vo, &no);
      check(q->lib->AsyncCopyDeviceToHost(out, q->dimVec[0] + vo, static_cast<size_t>(n) * widths[d], q->stream, q->device));
      out += static_cast<size_t>(n) * widths[d];
    }
    for (size_t d = 0; d < widths.size() && n; d++) {
      int64_t vo, no;
      dimension_start_offsets(q->ndw, static_cast<int>(d), q->resultCapacity, &vo, &no);
      check(q->lib->AsyncCopyDeviceToHost(out, q->dimVec[0] + no, static_cast<size_t>(n), q->stream, q->device));
      out += n;
    }
    if (n && measures) check(q->lib->AsyncCopyDeviceToHost(measures, q->measureVec[0], static_cast<size_t>(n) * q->plan.measureBytes(),
                                               q->stream, q->device));
    q->wait();
    return 0;
  } catch (std::exception &e) {
    set_err(err, errLen, e.what());
    return -1;
  }
}

void AresQueryDestroy(AresQuery *q) {
  if (!q) return;
  try {
    q->cleanupBeforeAggregation();
    q->releaseAll();
  } catch (...) {
  }
  delete q;
}

}  // extern "C"

// ---- shards across devices -----------------------------------------------------------------------------
struct AresComm {
  int rank = 0, nranks = 1;
  AresAllGatherFn allGather = nullptr;
  void *user = nullptr;
  // RCCL binding (AresCommCreateRccl)
  void *rcclHandle = nullptr, *rcclComm = nullptr;
  int (*ncclAllGather)(const void *, void *, size_t, int, void *, void *) = nullptr;
  int (*ncclSend)(const void *, size_t, int, int, void *, void *) = nullptr;
  int (*ncclRecv)(void *, size_t, int, int, void *, void *) = nullptr;
  int (*ncclGroupStart)() = nullptr;
  int (*ncclGroupEnd)() = nullptr;
  AresAllToAllFn allToAll = nullptr;  // optional: without it blocks travel by padded all-gathers
  void *localRank = nullptr;          // AresCommCreateLocal: this rank's end of the in-process rendezvous
  int (*ncclCommDestroy)(void *) = nullptr;
  const char *(*ncclGetErrorString)(int) = nullptr;
};

namespace {
struct NcclUniqueId { char internal[128]; };

int rccl_all_gather(void *user, const void *send, void *recv, size_t bytesPerRank, void *stream) {
  AresComm *c = static_cast<AresComm *>(user);
  return c->ncclAllGather(send, recv, bytesPerRank, /*ncclUint8*/ 1, c->rcclComm, stream);
}

// all-to-all of byte blocks over RCCL: one grouped send / receive pair per peer
int rccl_all_to_all(void *user, const void *send, const size_t *sendBytes, const size_t *sendOffsets, void *recv,
                    const size_t *recvBytes, const size_t *recvOffsets, void *stream) {
  AresComm *c = static_cast<AresComm *>(user);
  int rc = c->ncclGroupStart();
  for (int r = 0; r < c->nranks && rc == 0; r++) {
    if (sendBytes[r]) rc = c->ncclSend(static_cast<const uint8_t *>(send) + sendOffsets[r], sendBytes[r], /*ncclUint8*/ 1, r, c->rcclComm, stream);
    if (rc == 0 && recvBytes[r]) rc = c->ncclRecv(static_cast<uint8_t *>(recv) + recvOffsets[r], recvBytes[r], 1, r, c->rcclComm, stream);
  }
  const int end = c->ncclGroupEnd();
  return rc ? rc : end;
}

void *open_rccl(std::string *why) {
  for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    if (void *h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) return h;
  }
  *why = std::string("cannot load librccl.so: ") + dlerror();
  return nullptr;
}

bool select_device(int device, std::string *why) {
  void *hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
  auto set = hip ? reinterpret_cast<int (*)(int)>(dlsym(hip, "hipSetDevice")) : nullptr;
  if (!set || set(device) != 0) {
    *why = "hipSetDevice failed";
    return false;
  }
  return true;
}

// all_gather of `bytes` host bytes per rank through device staging buffers of the query's allocator
void gather_words(AresQuery *q, AresComm *c, const void *mine, void *all, size_t bytes) {
  uint8_t *send = q->alloc(bytes), *recv = q->alloc(bytes * c->nranks);
  check(q->lib->AsyncCopyHostToDevice(send, const_cast<void *>(mine), bytes, q->stream, q->device));
  q->lib->noteWrite(q->device, recv, bytes * c->nranks);
  if (c->allGather(c->user, send, recv, bytes, q->stream) != 0) throw AbiError("all-gather of the partial sizes failed");
  check(q->lib->AsyncCopyDeviceToHost(all, recv, bytes * c->nranks, q->stream, q->device));
  q->wait();
  q->release(send);
  q->release(recv);
}
}  // namespace

extern "C" {

AresComm *AresCommCreate(int rank, int nranks, AresAllGatherFn allGather, void *user) {
  if (!allGather || nranks < 1 || rank < 0 || rank >= nranks) return nullptr;
  AresComm *c = new AresComm;
  c->rank = rank;
  c->nranks = nranks;
  c->allGather = allGather;
  c->user = user;
  return c;
}

int AresCommRcclUniqueId(uint8_t id[128], char *err, int errLen) {
  std::string why;
  void *h = open_rccl(&why);
  auto get = h ? reinterpret_cast<int (*)(NcclUniqueId *)>(dlsym(h, "ncclGetUniqueId")) : nullptr;
  NcclUniqueId u;
  if (!get || get(&u) != 0) {
    set_err(err, errLen, h ? "ncclGetUniqueId failed" : why.c_str());
    return -1;
  }
  memcpy(id, u.internal, 128);
  return 0;
}

AresComm *AresCommCreateRccl(const uint8_t id[128], int rank, int nranks, int device, char *err, int errLen) {
  std::string why;
  void *h = open_rccl(&why);
  if (!h || !select_device(device, &why)) {
    set_err(err, errLen, why.c_str());
    return nullptr;
  }
  auto init = reinterpret_cast<int (*)(void **, int, NcclUniqueId, int)>(dlsym(h, "ncclCommInitRank"));
  AresComm *c = new AresComm;
  c->rank = rank;
  c->nranks = nranks;
  c->rcclHandle = h;
  c->ncclAllGather = reinterpret_cast<decltype(c->ncclAllGather)>(dlsym(h, "ncclAllGather"));
  c->ncclCommDestroy = reinterpret_cast<decltype(c->ncclCommDestroy)>(dlsym(h, "ncclCommDestroy"));
  c->ncclGetErrorString = reinterpret_cast<decltype(c->ncclGetErrorString)>(dlsym(h, "ncclGetErrorString"));
  NcclUniqueId u;
  memcpy(u.internal, id, 128);
  int rc = -1;
  if (!init || !c->ncclAllGather || !c->ncclCommDestroy || (rc = init(&c->rcclComm, nranks, u, rank)) != 0) {
    set_err(err, errLen, (std::string("ncclCommInitRank failed: ") +
                          (c->ncclGetErrorString && rc > 0 ? c->ncclGetErrorString(rc) : "missing symbol")).c_str());
    delete c;
    return nullptr;
  }
  c->allGather = &rccl_all_gather;
  c->user = c;
  c->ncclSend = reinterpret_cast<decltype(c->ncclSend)>(dlsym(h, "ncclSend"));
  c->ncclRecv = reinterpret_cast<decltype(c->ncclRecv)>(dlsym(h, "ncclRecv"));
  c->ncclGroupStart = reinterpret_cast<decltype(c->ncclGroupStart)>(dlsym(h, "ncclGroupStart"));
  c->ncclGroupEnd = reinterpret_cast<decltype(c->ncclGroupEnd)>(dlsym(h, "ncclGroupEnd"));
  if (c->ncclSend && c->ncclRecv && c->ncclGroupStart && c->ncclGroupEnd) c->allToAll = &rccl_all_to_all;
  return c;
}

// ---- ranks as threads of ONE process: the reference's process model (one device_manager device per shard inside
// the server process, query/device_manager.go:185-218) ---------------------------------------------------------
// The ranks rendezvous in host memory; a rank copies every peer's block itself — hipMemcpyAsync (peer-to-peer
// over xGMI through unified addressing) on its own stream for device memory, memcpy for host memory.
namespace {
struct LocalHub {
  std::mutex m;
  std::condition_variable cv;
  int nranks = 0, arrived = 0, refs = 0;
  uint64_t phase = 0;
  std::vector<const void *> send;
  int (*copyAsync)(void *, const void *, size_t, int, void *) = nullptr;  // hipMemcpyAsync(dst, src, n, hipMemcpyDefault, stream)
  int (*streamSync)(void *) = nullptr;
  // a rank that failed inside a collective says so here: every rank passes both barriers of the collective whatever
  // happened to it (a rank that left early would leave the others waiting forever) and returns non-zero when any failed
  std::atomic<int> failed{0};
  void barrier() {
    std::unique_lock<std::mutex> lock(m);
    const uint64_t my = phase;
    if (++arrived == nranks) {
      arrived = 0;
      phase++;
      cv.notify_all();
    } else {
      cv.wait(lock, [&] { return phase != my; });
    }
  }
};
struct LocalRank {
  LocalHub *hub;
  int rank;
};

int local_all_gather(void *user, const void *send, void *recv, size_t bytesPerRank, void *stream) {
  LocalRank *me = static_cast<LocalRank *>(user);
  LocalHub *h = me->hub;
  int rc = 0;
  if (h->streamSync && h->streamSync(stream) != 0) rc = 1;  // what this rank contributes has been produced
  if (rc) h->failed.store(1);
  h->send[me->rank] = send;
  h->barrier();
  const bool go = h->failed.load() == 0;  // (written before the barrier every rank has passed)
  for (int r = 0; r < h->nranks && rc == 0 && go; r++) {
    uint8_t *dst = static_cast<uint8_t *>(recv) + bytesPerRank * r;
    if (h->copyAsync) rc = h->copyAsync(dst, h->send[r], bytesPerRank, /*hipMemcpyDefault*/ 4, stream);
    else memcpy(dst, h->send[r], bytesPerRank);
  }
  if (rc == 0 && go && h->streamSync) rc = h->streamSync(stream);  // nobody frees a block a peer is still reading
  if (rc) h->failed.store(1);
  h->barrier();
  return (rc || h->failed.load()) ? 1 : 0;  // sticky: a communicator that failed once stays failed
}
}  // namespace

int AresCommCreateLocal(int nranks, int deviceMemory, AresComm **out, char *err, int errLen) {
  if (nranks < 1 || !out) {
    set_err(err, errLen, "AresCommCreateLocal: bad arguments");
    return -1;
  }
  LocalHub *hub = new LocalHub;
  hub->nranks = nranks;
  hub->refs = nranks;
  hub->send.assign(nranks, nullptr);
  if (deviceMemory) {
    void *hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    hub->copyAsync = hip ? reinterpret_cast<decltype(hub->copyAsync)>(dlsym(hip, "hipMemcpyAsync")) : nullptr;
    hub->streamSync = hip ? reinterpret_cast<decltype(hub->streamSync)>(dlsym(hip, "hipStreamSynchronize")) : nullptr;
    if (!hub->copyAsync || !hub->streamSync) {
      delete hub;
      set_err(err, errLen, "AresCommCreateLocal: cannot bind hipMemcpyAsync / hipStreamSynchronize");
      return -1;
    }
  }
  for (int r = 0; r < nranks; r++) {
    AresComm *c = new AresComm;
    c->rank = r;
    c->nranks = nranks;
    c->allGather = &local_all_gather;
    c->user = new LocalRank{hub, r};
    c->localRank = c->user;
    out[r] = c;
  }
  return 0;
}

void AresCommSetAllToAll(AresComm *c, AresAllToAllFn allToAll) {
  if (c) c->allToAll = allToAll;
}

void AresCommDestroy(AresComm *c) {
  if (!c) return;
  if (c->rcclComm && c->ncclCommDestroy) c->ncclCommDestroy(c->rcclComm);
  if (c->localRank) {
    LocalRank *lr = static_cast<LocalRank *>(c->localRank);
    bool last;
    {
      std::lock_guard<std::mutex> lock(lr->hub->m);
      last = --lr->hub->refs == 0;
    }
    if (last) delete lr->hub;
    delete lr;
  }
  delete c;
}

// HyperLogLog shard merge.  A shard that has NOT been finalised (every batch ran with isLastBatch = 0) holds its
// state as entries (dimension row, value = rho << 16 | register), one per (group, register) it has seen, in
// rows [0, resultSize) of dimension / measure vector [0].  The ranks exchange those entries in one all-gather;
// every rank then feeds the OTHER ranks' entries to the library's own HyperLogLog call as one more batch — with
// isLastBatch = 1 — behind its own state: the call keeps the maximum rho per (group, register), which is the
// register-max merge of the broker (broker/result_merge.go:95-104, query/common/hll.go:148), and encodes the
// final dense / sparse vector.  Every rank ends with the same set of (group, register) entries and therefore the same
// encoded registers per group; the ORDER of the groups in the result is rank dependent (a rank's own groups come first).
static int merge_shards_hll(AresQuery *q, AresComm *c) {
  // precondition, agreed on by every rank before anybody leaves: a rank that threw ahead of the first collective would
  // leave the others waiting in it
  const bool unfit = q->isLastBatch || q->hllVector;
  std::vector<int> widths;
  for (int k = 0; k < NUM_DIM_WIDTH; k++)
    for (int j = 0; j < q->ndw[k]; j++) widths.push_back(kDimWidths[k]);
  const int nd = static_cast<int>(widths.size()), world = c->nranks, mb = 4;
  int64_t valueBytes = 0;
  for (int w : widths) valueBytes += w;
  const int64_t rowBytes = valueBytes + nd + mb;
  std::vector<int64_t> sizes(world, 0);
  const int64_t mine = q->resultSize;
  {
    std::vector<int64_t> pairs(2 * static_cast<size_t>(world), 0);
    const int64_t me[2] = {mine, unfit ? 1 : 0};
    gather_words(q, c, me, pairs.data(), 2 * sizeof(int64_t));
    bool anyUnfit = false;
    for (int r = 0; r < world; r++) {
      sizes[r] = pairs[2 * r];
      anyUnfit = anyUnfit || pairs[2 * r + 1] != 0;
    }
    if (anyUnfit)
      throw AbiError(unfit ? "HyperLogLog shard merge needs the shard's intermediate entries: run every batch with isLastBatch = 0, the merge finalises"
                           : "HyperLogLog shard merge: another rank's shard was already finalised (isLastBatch = 1 before the merge)");
  }
  int64_t gmax = 1, others = 0;
  for (int r = 0; r < world; r++) {
    gmax = std::max(gmax, sizes[r]);
    if (r != c->rank) others += sizes[r];
  }
  if (others + mine > INT32_MAX) throw AbiError("merged result exceeds 2^31 rows");
  std::vector<int64_t> sect;
  int64_t off = 0;
  for (int w : widths) { sect.push_back(off); off += gmax * w; }
  for (int d = 0; d < nd; d++) { sect.push_back(off); off += gmax; }
  sect.push_back(off);
  const size_t packedBytes = static_cast<size_t>(gmax * rowBytes);
  uint8_t *packed = q->alloc(packedBytes), *gathered = q->alloc(packedBytes * world);
  for (int d = 0; d < nd && mine; d++) {
    int64_t vo, no;
    dimension_start_offsets(q->ndw, d, q->resultCapacity, &vo, &no);
    q->d2d(packed + sect[d], q->dimVec[0] + vo, static_cast<size_t>(mine) * widths[d]);
    q->d2d(packed + sect[nd + d], q->dimVec[0] + no, static_cast<size_t>(mine));
  }
  if (mine) q->d2d(packed + sect[2 * nd], q->measureVec[0], static_cast<size_t>(mine) * mb);
  q->lib->noteWrite(q->device, gathered, packedBytes * world);
  if (c->allGather(c->user, packed, gathered, packedBytes, q->stream) != 0) throw AbiError("all-gather of the HyperLogLog entries failed");
  q->wait();
  // the other ranks' entries are this rank's last batch: rows [resultSize, resultSize + others) of dimension vector [0],
  // their values in measure vector [1] (where a batch's hll values go, query/time_series_aggregate.go:404-408)
  q->size = static_cast<int>(others);
  q->prepareForDimAndMeasureEval();
  int64_t base = 0;
  for (int r = 0; r < world; r++) {
    if (r == c->rank || !sizes[r]) continue;
    const uint8_t *src = gathered + packedBytes * r;
    for (int d = 0; d < nd; d++) {
      int64_t vo, no;
      dimension_start_offsets(q->ndw, d, q->resultCapacity, &vo, &no);
      q->d2d(q->dimVec[0] + vo + (q->resultSize + base) * widths[d], const_cast<uint8_t *>(src) + sect[d], static_cast<size_t>(sizes[r]) * widths[d]);
      q->d2d(q->dimVec[0] + no + q->resultSize + base, const_cast<uint8_t *>(src) + sect[nd + d], static_cast<size_t>(sizes[r]));
    }
    q->d2d(q->measureVec[1] + base * mb, const_cast<uint8_t *>(src) + sect[2 * nd], static_cast<size_t>(sizes[r]) * mb);
    base += sizes[r];
  }
  q->wait();
  q->release(packed);
  q->release(gathered);
  q->isLastBatch = true;
  q->reduce();
  q->postExec();
  return 0;
}

int AresQueryMergeShards(AresQuery *q, AresComm *c, char *err, int errLen) {
  try {
    if (!c || !c->allGather) throw AbiError("no communicator");
    if (q->plan.isHLL()) return merge_shards_hll(q, c);
    std::vector<int> widths;
    for (int k = 0; k < NUM_DIM_WIDTH; k++)
      for (int j = 0; j < q->ndw[k]; j++) widths.push_back(kDimWidths[k]);
    const int nd = static_cast<int>(widths.size()), mb = q->plan.measureBytes(), world = c->nranks;
    int64_t valueBytes = 0;
    for (int w : widths) valueBytes += w;
    const int64_t rowBytes = valueBytes + nd + mb;
    // 1. sizes
    std::vector<int64_t> sizes(world, 0);
    const int64_t mine = q->resultSize;
    gather_words(q, c, &mine, sizes.data(), sizeof(int64_t));
    int64_t gmax = 1, total = 0;
    for (int64_t sz : sizes) {
      gmax = std::max(gmax, sz);
      total += sz;
    }
    if (total > INT32_MAX) throw AbiError("merged result exceeds 2^31 rows");
    // 2. columnar partial, padded to gmax rows: [dim values...][dim validity...][measures]
    std::vector<int64_t> sect;
    int64_t off = 0;
    for (int w : widths) { sect.push_back(off); off += gmax * w; }
    for (int d = 0; d < nd; d++) { sect.push_back(off); off += gmax; }
    sect.push_back(off);
    const size_t packedBytes = static_cast<size_t>(gmax * rowBytes);
    uint8_t *packed = q->alloc(packedBytes), *gathered = q->alloc(packedBytes * world);
    const int64_t g = q->resultSize;
    for (int d = 0; d < nd && g; d++) {
      int64_t vo, no;
      dimension_start_offsets(q->ndw, d, q->resultCapacity, &vo, &no);
      q->d2d(packed + sect[d], q->dimVec[0] + vo, static_cast<size_t>(g) * widths[d]);
      q->d2d(packed + sect[nd + d], q->dimVec[0] + no, static_cast<size_t>(g));
    }
    if (g) q->d2d(packed + sect[2 * nd], q->measureVec[0], static_cast<size_t>(g) * mb);
    // 3. one exchange
    q->lib->noteWrite(q->device, gathered, packedBytes * world);
    if (c->allGather(c->user, packed, gathered, packedBytes, q->stream) != 0) throw AbiError("all-gather of the partial group tables failed");
    // 4. append every rank's partial into fresh result vectors and re-reduce
    q->wait();
    q->releaseAll();
    q->resultSize = 0;
    q->size = static_cast<int>(total);
    q->prepareForDimAndMeasureEval();  // capacity total + 12.5 %, both buffers of every vector
    int64_t base = 0;
    for (int r = 0; r < world; r++) {
      const uint8_t *src = gathered + packedBytes * r;
      for (int d = 0; d < nd && sizes[r]; d++) {
        int64_t vo, no;
        dimension_start_offsets(q->ndw, d, q->resultCapacity, &vo, &no);
        q->d2d(q->dimVec[0] + vo + base * widths[d], const_cast<uint8_t *>(src) + sect[d], static_cast<size_t>(sizes[r]) * widths[d]);
        q->d2d(q->dimVec[0] + no + base, const_cast<uint8_t *>(src) + sect[nd + d], static_cast<size_t>(sizes[r]));
      }
      if (sizes[r]) q->d2d(q->measureVec[0] + base * mb, const_cast<uint8_t *>(src) + sect[2 * nd], static_cast<size_t>(sizes[r]) * mb);
      base += sizes[r];
    }
    q->wait();
    q->release(packed);
    q->release(gathered);
    if (total > 0) {
      q->reduce();  // HashReduce or InitIndexVector + Sort + Reduce over rows [0, size) of vector [0] into [1]
      q->postExec();
    } else {
      q->size = 0;
    }
    return 0;
  } catch (std::exception &e) {
    set_err(err, errLen, e.what());
    return -1;
  }
}

// Option (B) of SURVEY.md 8e: every rank ends with the groups whose 64-bit row hash falls into its share
// of the hash range, so no rank ever holds the whole table.  Built from the library's own entry points:
// Sort gives the row hashes and the order, Reduce lays the table out in that order (a shard's rows are
// distinct, so it only gathers), the slice for rank r is then contiguous; blocks travel by one
// all-to-all (or padded all-gathers when the communicator has none) and the receiver re-reduces what
// arrived with the query's own reduction.
int AresQueryMergeShardsPartitioned(AresQuery *q, AresComm *c, int64_t *totalGroups, char *err, int errLen) {
  try {
    if (!c || !c->allGather) throw AbiError("no communicator");
    if (q->plan.isHLL()) throw AbiError("HyperLogLog shards merge through AresQueryMergeShards (all-gather of the entries + register-max)");
    std::vector<int> widths;
    for (int k = 0; k < NUM_DIM_WIDTH; k++)
      for (int j = 0; j < q->ndw[k]; j++) widths.push_back(kDimWidths[k]);
    const int nd = static_cast<int>(widths.size()), mb = q->plan.measureBytes(), world = c->nranks;
    int64_t valueBytes = 0;
    for (int w : widths) valueBytes += w;
    const int64_t rowBytes = valueBytes + nd + mb;
    const int n = q->resultSize;
    // 1. the local table in hash order: Sort (hashes + order) and Reduce (gather) of vector [0] into [1]
    std::vector<int64_t> split(world + 1, 0);
    if (n > 0) {
      const int64_t cap = q->resultCapacity;
      const bool ownVectors = q->hashVec[0] == nullptr;
      if (ownVectors) {
        for (int i = 0; i < 2; i++) {
          q->hashVec[i] = q->alloc<uint64_t>(static_cast<size_t>(cap) * 8);
          q->dimIndexVec[i] = q->alloc<uint32_t>(static_cast<size_t>(cap) * 4);
        }
      }
      check(q->lib->InitIndexVector(q->dimIndexVec[0], 0, n, q->stream, q->device));
      check(q->lib->Sort(q->dimensionVector(0), n, q->stream, q->device));
      const int kept = static_cast<int>(check(q->lib->Reduce(q->dimensionVector(0), q->measureVec[0], q->dimensionVector(1),
                                                             q->measureVec[1], mb, n, static_cast<AggregateFunction>(q->plan.aggFunc),
                                                             q->stream, q->device)));
      if (kept != n) throw AbiError("the shard's group table holds rows with equal 64-bit hashes");
      // 2. rank r takes hashes in [r * 2^64 / world, (r + 1) * 2^64 / world): binary searches in the sorted hashes
      auto hashAt = [&](int64_t i) {
        uint64_t h = 0;
        check(q->lib->AsyncCopyDeviceToHost(&h, q->hashVec[0] + i, 8, q->stream, q->device));
        q->wait();
        return h;
      };
      for (int r = 1; r < world; r++) {
        const uint64_t bound = static_cast<uint64_t>((static_cast<unsigned __int128>(r) << 64) / static_cast<unsigned>(world));
        int64_t lo = split[r - 1], hi = n;  // first position with hash >= bound
        while (lo < hi) {
          const int64_t mid = (lo + hi) / 2;
          if (hashAt(mid) < bound) lo = mid + 1;
          else hi = mid;
        }
        split[r] = lo;
      }
      split[world] = n;
      q->wait();
      q->swapResultBuffers();  // vector [0] is the table in hash order now
      q->size = 0;
      if (ownVectors) {
        for (int i = 0; i < 2; i++) {
          q->release(q->hashVec[i]); q->release(q->dimIndexVec[i]);
          q->hashVec[i] = nullptr; q->dimIndexVec[i] = nullptr;
        }
      }
    }
    // 3. who sends how much to whom
    std::vector<int64_t> mine(world), all(static_cast<size_t>(world) * world);
    for (int r = 0; r < world; r++) mine[r] = split[r + 1] - split[r];
    gather_words(q, c, mine.data(), all.data(), sizeof(int64_t) * world);
    std::vector<int64_t> recvRows(world);
    int64_t total = 0, grand = 0, maxBlock = 1;
    for (int src = 0; src < world; src++) {
      recvRows[src] = all[static_cast<size_t>(src) * world + c->rank];
      total += recvRows[src];
      for (int dst = 0; dst < world; dst++) {
        grand += all[static_cast<size_t>(src) * world + dst];
        maxBlock = std::max(maxBlock, all[static_cast<size_t>(src) * world + dst]);
      }
    }
    if (total > INT32_MAX) throw AbiError("a rank's share of the merged result exceeds 2^31 rows");
    // 4. pack one block per destination: [dim values...][dim validity...][measures] of its rows
    auto sections = [&](int64_t rows, std::vector<int64_t> &sect) {
      sect.clear();
      int64_t off = 0;
      for (int w : widths) { sect.push_back(off); off += rows * w; }
      for (int d = 0; d < nd; d++) { sect.push_back(off); off += rows; }
      sect.push_back(off);
    };
    std::vector<size_t> sendBytes(world), sendOff(world), recvBytes(world), recvOff(world);
    size_t sendTotal = 0, recvTotal = 0;
    for (int r = 0; r < world; r++) {
      sendBytes[r] = static_cast<size_t>(mine[r] * rowBytes); sendOff[r] = sendTotal; sendTotal += sendBytes[r];
      recvBytes[r] = static_cast<size_t>(recvRows[r] * rowBytes); recvOff[r] = recvTotal; recvTotal += recvBytes[r];
    }
    uint8_t *packed = q->alloc(sendTotal), *arrived = q->alloc(recvTotal);
    std::vector<int64_t> sect;
    for (int r = 0; r < world; r++) {
      if (!mine[r]) continue;
      sections(mine[r], sect);
      uint8_t *block = packed + sendOff[r];
      for (int d = 0; d < nd; d++) {
        int64_t vo, no;
        dimension_start_offsets(q->ndw, d, q->resultCapacity, &vo, &no);
        q->d2d(block + sect[d], q->dimVec[0] + vo + split[r] * widths[d], static_cast<size_t>(mine[r]) * widths[d]);
        q->d2d(block + sect[nd + d], q->dimVec[0] + no + split[r], static_cast<size_t>(mine[r]));
      }
      q->d2d(block + sect[2 * nd], q->measureVec[0] + split[r] * mb, static_cast<size_t>(mine[r]) * mb);
    }
    // 5. the exchange
    q->lib->noteWrite(q->device, arrived, recvTotal);
    if (c->allToAll) {
      if (c->allToAll(c->user, packed, sendBytes.data(), sendOff.data(), arrived, recvBytes.data(), recvOff.data(), q->stream) != 0)
        throw AbiError("all-to-all of the group table blocks failed");
    } else {  // destination by destination: everybody contributes its block for rank r, rank r keeps them
      const size_t pad = static_cast<size_t>(maxBlock * rowBytes);
      uint8_t *one = q->alloc(pad), *every = q->alloc(pad * world);
      for (int r = 0; r < world; r++) {
        if (sendBytes[r]) q->d2d(one, packed + sendOff[r], sendBytes[r]);
        q->lib->noteWrite(q->device, every, pad * world);
        if (c->allGather(c->user, one, every, pad, q->stream) != 0) throw AbiError("all-gather of the group table blocks failed");
        if (r == c->rank)
          for (int src = 0; src < world; src++)
            if (recvBytes[src]) q->d2d(arrived + recvOff[src], every + pad * src, recvBytes[src]);
        q->wait();  // `one` is refilled for the next destination
      }
      q->release(one);
      q->release(every);
    }
    // 6. append what arrived into fresh result vectors and re-reduce
    q->wait();
    q->releaseAll();
    q->resultSize = 0;
    q->size = static_cast<int>(total);
    if (total > 0) q->prepareForDimAndMeasureEval();
    int64_t base = 0;
    for (int src = 0; src < world; src++) {
      if (!recvRows[src]) continue;
      sections(recvRows[src], sect);
      const uint8_t *block = arrived + recvOff[src];
      for (int d = 0; d < nd; d++) {
        int64_t vo, no;
        dimension_start_offsets(q->ndw, d, q->resultCapacity, &vo, &no);
        q->d2d(q->dimVec[0] + vo + base * widths[d], const_cast<uint8_t *>(block) + sect[d], static_cast<size_t>(recvRows[src]) * widths[d]);
        q->d2d(q->dimVec[0] + no + base, const_cast<uint8_t *>(block) + sect[nd + d], static_cast<size_t>(recvRows[src]));
      }
      q->d2d(q->measureVec[0] + base * mb, const_cast<uint8_t *>(block) + sect[2 * nd], static_cast<size_t>(recvRows[src]) * mb);
      base += recvRows[src];
    }
    q->wait();
    q->release(packed);
    q->release(arrived);
    if (total > 0) {
      q->reduce();
      q->postExec();
    } else {
      q->size = 0;
    }
    // 7. the size of the whole result: every rank's share
    if (totalGroups) {
      std::vector<int64_t> shares(world, 0);
      const int64_t share = q->resultSize;
      gather_words(q, c, &share, shares.data(), sizeof(int64_t));
      *totalGroups = 0;
      for (int64_t v : shares) *totalGroups += v;
    }
    (void)grand;
    return 0;
  } catch (std::exception &e) {
    set_err(err, errLen, e.what());
    return -1;
  }
}

// ---- host batches: transfer pipeline + device-resident column cache -------------------------------------
}  // extern "C"

struct AresColumnCache {
  AbiLibrary *lib;
  int device;
  size_t budget, bytes = 0;
  uint64_t clock = 0;
  struct Entry {
    void *ptr;
    size_t bytes;
    uint64_t lastUse;
    int pins;
  };
  std::map<uint64_t, Entry> entries;
  std::mutex mu;

  // a resident column (pinned until unpin) or nullptr
  void *find(uint64_t key) {
    std::lock_guard<std::mutex> lock(mu);
    auto it = entries.find(key);
    if (it == entries.end()) return nullptr;
    it->second.lastUse = ++clock;
    it->second.pins++;
    return it->second.ptr;
  }
  // takes ownership of `ptr` (an uploaded column) if the budget allows, evicting the least recently used
  // unpinned entries; returns false when the caller keeps ownership
  bool insert(uint64_t key, void *ptr, size_t n) {
    std::vector<void *> evicted;
    {
      std::lock_guard<std::mutex> lock(mu);
      if (n > budget || entries.count(key)) return false;
      while (bytes + n > budget) {
        auto victim = entries.end();
        for (auto it = entries.begin(); it != entries.end(); ++it)
          if (it->second.pins == 0 && (victim == entries.end() || it->second.lastUse < victim->second.lastUse)) victim = it;
        if (victim == entries.end()) return false;
        bytes -= victim->second.bytes;
        evicted.push_back(victim->second.ptr);
        entries.erase(victim);
      }
      entries[key] = Entry{ptr, n, ++clock, 1};
      bytes += n;
    }
    for (void *p : evicted) check(lib->DeviceFree(p, device));
    return true;
  }
  void unpin(uint64_t key) {
    std::lock_guard<std::mutex> lock(mu);
    auto it = entries.find(key);
    if (it != entries.end() && it->second.pins > 0) it->second.pins--;
  }
};

extern "C" {

AresColumnCache *AresColumnCacheCreate(void *driver, int device, size_t budgetBytes) {
  AresColumnCache *c = new AresColumnCache;
  c->lib = static_cast<AbiLibrary *>(driver);
  c->device = device;
  c->budget = budgetBytes;
  return c;
}

void AresColumnCacheDestroy(AresColumnCache *c) {
  if (!c) return;
  for (auto &kv : c->entries) {
    try {
      check(c->lib->DeviceFree(kv.second.ptr, c->device));
    } catch (...) {
    }
  }
  delete c;
}

int AresQueryRunHostBatches(AresQuery *q, const AresHostColumn *columns, int numColumns, const int *batchSizes, int numBatches,
                            AresColumnCache *cache, uint64_t stats[4], char *err, int errLen) {
  uint64_t uploadedBytes = 0, uploads = 0, hits = 0;
  struct Staged {
    std::vector<VectorPartySlice> slices;
    std::vector<void *> owned;
    std::vector<uint64_t> pinned;
    int size = 0;
  };
  std::string workerError;
  auto run_staged = [&](Staged *b) {  // the executor of one batch, on the stream its transfer used
    try {
      q->ownedColumns = b->owned;
      q->runBatch(b->slices.data(), static_cast<int>(b->slices.size()), b->size, nullptr, 0);
    } catch (std::exception &e) {
      workerError = e.what();
      try {
        q->cleanupBeforeAggregation();
      } catch (...) {
      }
    }
    if (cache)
      for (uint64_t k : b->pinned) cache->unpin(k);
  };
  // ONE executor thread for the whole call (a thread per batch pays for its thread-local state in the library — a mapped pinned
  // result slot from hipHostMalloc among it — on every batch): batches are handed over one at a time
  struct Executor {
    std::mutex mu;
    std::condition_variable cv;
    Staged *job = nullptr;
    bool busy = false, quit = false;
    std::thread thread;
  } ex;
  auto submit = [&](Staged *b) {
    std::lock_guard<std::mutex> lock(ex.mu);
    ex.job = b;
    ex.busy = true;
    ex.cv.notify_all();
  };
  auto wait_idle = [&] {
    std::unique_lock<std::mutex> lock(ex.mu);
    ex.cv.wait(lock, [&] { return !ex.busy; });
  };
  auto stop = [&] {
    {
      std::lock_guard<std::mutex> lock(ex.mu);
      ex.quit = true;
      ex.cv.notify_all();
    }
    if (ex.thread.joinable()) ex.thread.join();
  };
  try {
    Staged prev, cur;
    bool havePrev = false;
    auto start_executor = [&] {  // (on the first batch that is handed over: a query whose columns all sit in the cache never starts it)
      ex.thread = std::thread([&] {
        for (;;) {
          Staged *b = nullptr;
          {
            std::unique_lock<std::mutex> lock(ex.mu);
            ex.cv.wait(lock, [&] { return ex.quit || ex.job; });
            if (!ex.job) return;
            b = ex.job;
            ex.job = nullptr;
          }
          run_staged(b);
          {
            std::lock_guard<std::mutex> lock(ex.mu);
            ex.busy = false;
            ex.cv.notify_all();
          }
        }
      });
    };
    for (int k = 0; k < numBatches; k++) {
      // (query/aql_processor.go:850-881) async transfer of batch k ...
      void *xfer = havePrev && q->otherStream ? q->otherStream : q->stream;
      cur = Staged();
      cur.size = batchSizes[k];
      // columns the cache holds first: a batch that needs no upload has nothing to overlap with — batch k-1 then runs on
      // THIS thread (handing it to the executor and waking up again costs ~0.1 ms per batch of condition-variable latency)
      std::vector<void *> devs(static_cast<size_t>(numColumns), nullptr);
      int misses = 0;
      for (int c = 0; c < numColumns; c++) {
        const AresHostColumn &hc = columns[static_cast<size_t>(k) * numColumns + c];
        devs[c] = (cache && hc.cacheKey) ? cache->find(hc.cacheKey) : nullptr;
        if (devs[c]) {
          hits++;
          cur.pinned.push_back(hc.cacheKey);
        } else {
          misses++;
        }
      }
      const bool handOver = havePrev && misses > 0;
      if (handOver) {
        if (!ex.thread.joinable()) start_executor();
        submit(&prev);  // ... while batch k-1 executes
      }
      try {
        for (int c = 0; c < numColumns; c++) {
          const AresHostColumn &hc = columns[static_cast<size_t>(k) * numColumns + c];
          void *dev = devs[c];
          if (!dev) {
            dev = reinterpret_cast<void *>(check(q->lib->DeviceAllocate(hc.bytes ? hc.bytes : 1, q->device)));
            check(q->lib->AsyncCopyHostToDevice(dev, const_cast<void *>(hc.host), hc.bytes, xfer, q->device));
            uploadedBytes += hc.bytes;
            uploads++;
            if (cache && hc.cacheKey && cache->insert(hc.cacheKey, dev, hc.bytes)) cur.pinned.push_back(hc.cacheKey);
            else cur.owned.push_back(dev);
          }
          VectorPartySlice vp = hc.slice;
          vp.BasePtr = static_cast<uint8_t *>(dev) + reinterpret_cast<uintptr_t>(hc.slice.BasePtr);
          cur.slices.push_back(vp);
        }
        if (misses > 0) check(q->lib->WaitForCudaStream(xfer, q->device));  // wait for the data transfer of the current batch
      } catch (...) {
        if (handOver) wait_idle();
        stop();
        throw;
      }
      if (handOver) wait_idle();
      else if (havePrev) run_staged(&prev);
      if (!workerError.empty()) throw AbiError(workerError);
      prev = cur;
      havePrev = true;
    }
    if (havePrev) run_staged(&prev);  // (nothing left to overlap with)
    stop();
    if (!workerError.empty()) throw AbiError(workerError);
    if (stats) {
      stats[0] = uploadedBytes;
      stats[1] = uploads;
      stats[2] = hits;
      stats[3] = cache ? cache->bytes : 0;
    }
    return 0;
  } catch (std::exception &e) {
    stop();
    set_err(err, errLen, e.what());
    return -1;
  }
}

}  // extern "C"
