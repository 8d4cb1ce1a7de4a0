// Host-side binding of the ABI's tagged unions (InputVector / OutputVector) to the plain
// device descriptors of device_model.hpp.  Restates the type rules of the reference binder
// (query/binder.hpp:102-264, :308-426; query/transform.hpp:112-171 and the three
// *_transform.cu output binders) as run-time checks instead of template recursion.
#pragma once

#include <memory>
#include <vector>

#include "common.hpp"
#include "device_model.hpp"

namespace ares {

inline int kind_of_datatype(int t) {
  switch (t) {
    case Bool: return K_BOOL;
    case Int8: case Int16: case Int32: return K_I32;
    case Uint8: case Uint16: case Uint32: return K_U32;
    case Float32: return K_F32;
    case Int64: return K_I64;
    case UUID: return K_UUID;
    case GeoPoint: return K_GEO;
    default: return K_NONE;
  }
}

inline int step_in_bytes(int t) {  // query/utils.hpp:186-204
  switch (t) {
    case Bool: case Int8: case Uint8: return 1;
    case Int16: case Uint16: return 2;
    case Int32: case Uint32: case Float32: return 4;
    case GeoPoint: case Int64: case Uint64: return 8;
    case UUID: return 16;
    default: throw std::invalid_argument("Unsupported data type for VectorPartyInput");
  }
}

inline bool is_wide(int kind) { return kind == K_I64 || kind == K_UUID || kind == K_GEO; }

inline int common_kind(int a, int b) {  // query/utils.hpp:83-126
  if (a == K_GEO || a == K_UUID) return a;
  if (a == K_F32 || b == K_F32) return K_F32;
  if (a == K_I64 || b == K_I64) return K_I64;
  if (a == K_I32 || b == K_I32) return K_I32;
  return K_U32;
}

inline void set_default(OperandD &op, const DefaultValue &dv, int kind) {
  op.cok = dv.HasDefault ? 1u : 0u;
  op.cbits = 0;
  op.c64[0] = op.c64[1] = 0;
  switch (kind) {
    case K_BOOL: op.cbits = dv.Value.BoolVal ? 1u : 0u; break;
    case K_I32: op.cbits = static_cast<uint32_t>(dv.Value.Int32Val); break;
    case K_U32: op.cbits = dv.Value.Uint32Val; break;
    case K_F32: memcpy(&op.cbits, &dv.Value.FloatVal, 4); break;
    case K_I64: memcpy(&op.c64[0], &dv.Value.Int64Val, 8); break;
    case K_UUID: memcpy(&op.c64[0], &dv.Value.UUIDVal, 16); break;
    case K_GEO: memcpy(&op.c64[0], &dv.Value.GeoPointVal, 8); break;
    default: break;
  }
}

// Device copies of per-call host arrays (foreign batch descriptors) kept alive until the kernels
// that read them have been enqueued; freed stream-ordered.
struct CallTemps {
  std::vector<std::unique_ptr<StreamBuffer>> buffers;
  std::vector<std::vector<ForeignBatchD>> hostCopies;  // must outlive the async H2D
};

inline void bind_operand(const InputVector &in, bool first, hipStream_t stream, OperandD &op, CallTemps &temps) {
  memset(&op, 0, sizeof(op));
  switch (in.Type) {
    case ConstantInput: {
      const ConstantVector &c = in.Vector.Constant;
      op.type = OP_CONST;
      op.cok = c.IsValid ? 1u : 0u;
      switch (c.DataType) {
        case ConstInt: op.kind = K_I32; op.cbits = static_cast<uint32_t>(c.Value.IntVal); return;
        case ConstFloat: op.kind = K_F32; memcpy(&op.cbits, &c.Value.FloatVal, 4); return;
        case ConstGeoPoint: op.kind = K_GEO; memcpy(&op.c64[0], &c.Value.GeoPointVal, 8); return;
        case ConstUUID: op.kind = K_UUID; memcpy(&op.c64[0], &c.Value.UUIDVal, 16); return;
      }
      throw std::invalid_argument("Unsupported constant data type");
    }
    case ScratchSpaceInput: {
      const ScratchSpaceVector &s = in.Vector.ScratchSpace;
      op.type = OP_SCRATCH;
      op.base = s.Values;
      op.nullsOff = s.NullsOffset;
      switch (s.DataType) {
        case Int32: op.kind = K_I32; return;
        case Uint32: op.kind = K_U32; return;
        case Float32: op.kind = K_F32; return;
        case UUID: op.kind = K_UUID; return;
        case GeoPoint: op.kind = K_GEO; return;
        default: throw std::invalid_argument("Unsupported data type for ScratchSpaceInput");
      }
    }
    case VectorPartyInput: {
      const VectorPartySlice &vp = in.Vector.VP;
      const int kind = kind_of_datatype(vp.DataType);
      if (kind == K_NONE || (!first && is_wide(kind)))
        throw std::invalid_argument("Unsupported data type for VectorPartyInput");
      op.kind = kind;
      if (vp.BasePtr == nullptr) {  // mode 0: the column is its default value
        op.type = OP_CONST;
        set_default(op, vp.DefaultValue, kind);
        return;
      }
      op.type = OP_COLUMN;
      op.base = vp.BasePtr;
      op.nullsOff = vp.NullsOffset;
      op.valuesOff = vp.ValuesOffset;
      op.length = vp.Length;
      op.bitOff = vp.StartingIndex;
      op.step = static_cast<uint8_t>(step_in_bytes(vp.DataType));
      op.mode = vp.ValuesOffset == 0 ? 1 : (vp.NullsOffset == 0 ? 2 : 3);
      if (kind == K_GEO && op.mode == 3) op.mode = 2;  // geo columns are never run-length decoded
      if (kind == K_GEO) op.nullsOff = 0;              // ... and keep validity at BasePtr (iterator.hpp:318-325)
      return;
    }
    case ForeignColumnInput: {
      const ForeignColumnVector &f = in.Vector.ForeignVP;
      const int kind = kind_of_datatype(f.DataType);
      if (kind == K_NONE || kind == K_GEO || (!first && is_wide(kind)))
        throw std::invalid_argument("Unsupported data type for VectorPartyInput");
      op.type = OP_FOREIGN;
      op.kind = kind;
      op.step = static_cast<uint8_t>(step_in_bytes(f.DataType));
      op.rids = f.RecordIDs;
      op.baseBatchID = f.BaseBatchID;
      op.numBatches = f.NumBatches;
      op.numRecLast = f.NumRecordsInLastBatch;
      op.tz = f.TimezoneLookup;
      op.tzSize = f.TimezoneLookupSize;
      set_default(op, f.DefaultValue, kind);
      // per-call upload of the batch descriptors (reference binder.hpp:591-633)
      temps.hostCopies.emplace_back(static_cast<size_t>(f.NumBatches > 0 ? f.NumBatches : 0));
      std::vector<ForeignBatchD> &h = temps.hostCopies.back();
      for (int i = 0; i < f.NumBatches; i++) {
        const VectorPartySlice &vp = f.Batches[i];
        h[i].base = vp.BasePtr;
        h[i].nullsOff = vp.NullsOffset;
        h[i].valuesOff = vp.ValuesOffset;
        h[i].bitOff = vp.StartingIndex;
        h[i].isConst = vp.BasePtr == nullptr;
      }
      temps.buffers.emplace_back(new StreamBuffer(sizeof(ForeignBatchD) * h.size() + 16, stream));
      if (!h.empty())
        hip_check(hipMemcpyAsync(temps.buffers.back()->get(), h.data(), sizeof(ForeignBatchD) * h.size(),
                                 hipMemcpyHostToDevice, stream),
                  "upload foreign batches");
      op.batches = temps.buffers.back()->as<ForeignBatchD>();
      return;
    }
    default:
      // array columns bind as the FIRST operand only, through bind_array (binder.hpp:385-426)
      throw std::invalid_argument("Unsupported input vector type for this operand");
  }
}

// identity of an aggregate in the measure's own representation (query/utils.hpp:169-184;
// note AGGR_MAX_FLOAT -> FLT_MIN is a reference quirk that parity preserves)
inline uint64_t identity_bits(int agg, int dtype) {
  double d = 0;
  int64_t l = 0;
  bool fl = false;
  switch (agg) {
    case AGGR_MIN_UNSIGNED: l = static_cast<int64_t>(UINT32_MAX); break;
    case AGGR_MIN_SIGNED: l = INT32_MAX; break;
    case AGGR_MIN_FLOAT: fl = true; d = FLT_MAX; break;
    case AGGR_MAX_SIGNED: l = INT32_MIN; break;
    case AGGR_MAX_FLOAT: fl = true; d = FLT_MIN; break;
    default: break;
  }
  uint64_t out = 0;
  switch (dtype) {
    case Int32: { int32_t x = fl ? static_cast<int32_t>(d) : static_cast<int32_t>(l); memcpy(&out, &x, 4); break; }
    case Uint32: { uint32_t x = fl ? static_cast<uint32_t>(d) : static_cast<uint32_t>(l); memcpy(&out, &x, 4); break; }
    case Float32: { float x = fl ? static_cast<float>(d) : static_cast<float>(l); memcpy(&out, &x, 4); break; }
    case Int64: { int64_t x = fl ? static_cast<int64_t>(d) : l; memcpy(&out, &x, 8); break; }
    case Float64: { double x = fl ? d : static_cast<double>(l); memcpy(&out, &x, 8); break; }
    default: break;
  }
  return out;
}

inline void bind_sink(const OutputVector &out, const uint32_t *baseCounts, SinkD &s) {
  memset(&s, 0, sizeof(s));
  switch (out.Type) {
    case ScratchSpaceOutput: {
      const ScratchSpaceVector &v = out.Vector.ScratchSpace;
      const int t = v.DataType;
      if (!(t == Int32 || t == Uint32 || t == Float32 || t == Int64 || t == UUID || t == GeoPoint))
        throw std::invalid_argument("Unsupported data type for ScratchSpaceOutput");
      s.type = SINK_SCRATCH;
      s.dtype = t;
      s.width = step_in_bytes(t);
      s.values = v.Values;
      s.nulls = v.Values + v.NullsOffset;
      return;
    }
    case DimensionOutput: {
      const DimensionOutputVector &v = out.Vector.Dimension;
      const int t = v.DataType;
      if (t == Uint64 || t == Float64 || kind_of_datatype(t) == K_NONE)
        throw std::invalid_argument("Unsupported data type for DimensionOutput");
      s.type = SINK_DIM;
      s.dtype = t;
      s.width = step_in_bytes(t);
      s.values = v.DimValues;
      s.nulls = v.DimNulls;
      return;
    }
    case MeasureOutput: {
      const MeasureOutputVector &v = out.Vector.Measure;
      const int t = v.DataType;
      if (!(t == Int32 || t == Uint32 || t == Float32 || t == Int64 || t == Float64))
        throw std::invalid_argument("Unsupported data type for MeasureOutput");
      s.type = SINK_MEASURE;
      s.dtype = t;
      s.width = (t == Int64 || t == Float64) ? 8 : 4;
      s.values = reinterpret_cast<uint8_t *>(v.Values);
      s.agg = v.AggFunc;
      s.identity = identity_bits(v.AggFunc, t);
      s.baseCounts = baseCounts;
      return;
    }
    default:
      throw std::invalid_argument("Unsupported output vector type");
  }
}

inline void bind_pred_sink(uint8_t *pred, SinkD &s) {
  memset(&s, 0, sizeof(s));
  s.type = SINK_PRED;
  s.dtype = Bool;
  s.width = 1;
  s.values = pred;
}

// validity rules for the second operand of a binary call (query/binder.hpp:266-306, filter.cu:88-103)
inline void check_binary_kinds(const OperandD &a, const OperandD &b, const InputVector &rhs) {
  if (a.kind == K_I64) throw std::invalid_argument("int64 data type is only supported in UnaryTransform");
  if (a.kind == K_GEO && !(rhs.Type == ConstantInput && b.kind == K_GEO))
    throw std::invalid_argument("Unsupported data type when value type of first input iterator is GeoPoint");
  if (a.kind == K_UUID && !(rhs.Type == ConstantInput && b.kind == K_UUID))
    throw std::invalid_argument("Unsupported data type when value type of first input iterator is UUID");
  if (!is_wide(a.kind) && is_wide(b.kind)) throw std::invalid_argument("Unsupported data type combination");
}

}  // namespace ares
