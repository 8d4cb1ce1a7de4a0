// Geo intersection for MI355X (gfx950): point-in-polygon filter / join against a batch of shapes.
//
// Reference: query/geo_intersects.cu (GeoBatchIntersects, WriteGeoShapeDim),
// query/iterator.hpp:1260-1452 (GeoPredicateIterator, GeoBatchIntersectIterator).
//
// The reference launches one thread per (entry, polygon point) pair and toggles predicate bits with
// atomicXor.  Here one lane owns one entry and a wavefront walks the polygon edges 64 at a time with
// register broadcasts (see geo_intersect_kernel); the crossing parity accumulates in registers (up
// to 8 words = 256 shapes): no atomics, no LDS, one read-modify-write of the predicate words per
// entry.  The same pass writes a keep byte per entry; the index vector and the RecordID vectors are
// then compacted by the filter's chain-free compaction kernels (transform.hip).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <vector>

#include "binding.hpp"
#include "common.hpp"
#include "device_model.hpp"
#include "lookback.hpp"

namespace ares {

int compact_by_predicate(const uint8_t *pred, uint32_t *indexVector, RecordID **recordIDVectors, int numForeignTables,
                         int n, hipStream_t stream);

namespace {

constexpr int kBlock = 256;

// where the geo points of the entries come from
struct GeoPointsD {
  int foreign;  // 0: main-table column through the index vector, 1: joined column through RecordIDs
  // main table (VectorPartyIterator<GeoPointT>, query/iterator.hpp:291-352)
  const uint8_t *base;
  uint32_t valuesOff;
  uint32_t bitOff;
  const uint32_t *index;
  // join (RecordIDJoinIterator<GeoPointT>, query/iterator.hpp:911-930)
  const RecordID *rids;
  const ForeignBatchD *batches;
  int32_t baseBatchID, numBatches, numRecLast;
  float defaultLat, defaultLong;
  uint32_t defaultOk;
};

struct GeoPointV {
  float lat, lng;
  bool ok;
};

__device__ __forceinline__ GeoPointV load_point(const GeoPointsD &P, int64_t i) {
  GeoPointV r;
  r.lat = 0.f;
  r.lng = 0.f;
  r.ok = false;
  if (!P.foreign) {
    const uint32_t row = P.index[i];
    const float2 v = reinterpret_cast<const float2 *>(P.base + P.valuesOff)[row];
    r.lat = v.x;
    r.lng = v.y;
    r.ok = P.valuesOff == 0 ? true : get_bit(P.base, row + P.bitOff) != 0;
    return r;
  }
  const RecordID rid = P.rids[i];
  if (rid.batchID != 0 &&
      (rid.batchID - P.baseBatchID < P.numBatches - 1 || rid.index < static_cast<uint32_t>(P.numRecLast))) {
    const ForeignBatchD b = P.batches[rid.batchID - P.baseBatchID];
    if (b.isConst) {
      r.lat = P.defaultLat;
      r.lng = P.defaultLong;
      r.ok = P.defaultOk != 0;
      return r;
    }
    const float2 v = reinterpret_cast<const float2 *>(b.base + b.valuesOff)[rid.index];
    r.lat = v.x;
    r.lng = v.y;
    r.ok = b.valuesOff == 0 ? true : get_bit(b.base, rid.index + b.bitOff) != 0;
  }
  return r;
}

// GeoPredicateIterator (query/iterator.hpp:1263-1315): first set bit as an int8 — shape numbers
// 128..255 wrap to negative values, which every consumer reads as "no shape"
template <int W>
__device__ __forceinline__ int first_shape(const uint32_t (&words)[W], int totalWords) {
#pragma unroll
  for (int w = 0; w < W; w++)
    if (w < totalWords && words[w]) return static_cast<int8_t>(w * 32 + __builtin_ctz(words[w]));
  return -1;
}

// One wavefront = 64 entries x all polygon edges, 64 edges at a time:
//   * lane l fetches polygon points e0+l and e0+l+1 (edge l of the chunk) — coalesced, cached;
//   * hot loop, no memory and no branches: for every polygon point k of the chunk the longitude is
//     broadcast with v_readlane and compared with the lane's own test longitude; the 65 results
//     form a per-lane bit string G, and "edge k straddles the test longitude" is G ^ (G >> 1),
//     masked with the wave-uniform set of real edges (same shape, no ring separator);
//   * only the few straddling edges of each lane reach the exact test: the lane pulls that edge's
//     coordinates out of the owning lane with ds_bpermute and evaluates the reference's
//     expression (same operations, same order, IEEE division) — every lane works on its own edge.
template <int W>
__global__ __launch_bounds__(kBlock) void geo_intersect_kernel(GeoPointsD P, const float *lats, const float *longs,
                                                               const uint8_t *shape, int numPoints, uint32_t *pred,
                                                               uint8_t *keep, int n, int totalWords, int inOrOut) {
  // a null point: "the first edge writes the verdict" (query/iterator.hpp:1372-1381) — if there is one
  const bool firstEdge = numPoints >= 2 && shape[0] == shape[1];
  const int numEdges = numPoints - 1;
  const int lane = threadIdx.x & 63;
  for (int64_t blockStart = static_cast<int64_t>(blockIdx.x) * kBlock; blockStart < n;
       blockStart += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int64_t i = blockStart + threadIdx.x;
    GeoPointV pt;
    pt.lat = pt.lng = 0.f;
    pt.ok = false;
    if (i < n) pt = load_point(P, i);
    // a lane without a valid point compares false against everything
    const float testLong = pt.ok ? pt.lng : __int_as_float(0x7fc00000);
    uint32_t acc[W];
#pragma unroll
    for (int w = 0; w < W; w++) acc[w] = 0;
    for (int e0 = 0; e0 < numEdges; e0 += 64) {
      const int p = e0 + lane;
      const bool have = p < numEdges;
      float lat1 = 0.f, lat2 = 0.f, long1 = 0.f, long2 = 0.f;
      int s1 = -1, s2 = -2;
      if (p < numPoints) {  // the chunk's last edge ends on a point that starts no edge of its own
        lat1 = lats[p];
        long1 = longs[p];
        s1 = shape[p];
      }
      if (have) {
        lat2 = lats[p + 1];
        long2 = longs[p + 1];
        s2 = shape[p + 1];
      }
      const uint64_t edges = __ballot(have && s1 == s2 && lat1 < FLT_MAX && lat2 < FLT_MAX);
      if (edges == 0) continue;
      // G: bit k = (longitude of polygon point e0+k > test longitude), k = 0..64
      const int long1Bits = __float_as_int(long1);
      uint32_t gLo = 0, gHi = 0;
#pragma unroll
      for (int k = 0; k < 32; k++) {
        const float a = __int_as_float(__builtin_amdgcn_readlane(long1Bits, k));
        const float b = __int_as_float(__builtin_amdgcn_readlane(long1Bits, k + 32));
        gLo |= a > testLong ? (1u << k) : 0u;
        gHi |= b > testLong ? (1u << k) : 0u;
      }
      const float last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(long2), 63));
      const uint64_t g = (static_cast<uint64_t>(gHi) << 32) | gLo;
      const uint64_t gNext = (g >> 1) | (last > testLong ? (1ull << 63) : 0ull);
      uint64_t cross = (g ^ gNext) & edges;
      const float dLat = __fsub_rn(lat2, lat1);
      while (__ballot(cross != 0)) {
        const bool active = cross != 0;
        const int k = active ? __builtin_ctzll(cross) : 0;
        cross &= cross - 1;
        const float eLat1 = __shfl(lat1, k), eLong1 = __shfl(long1, k), eLong2 = __shfl(long2, k);
        const float eDLat = __shfl(dLat, k);
        const int s = __shfl(s1, k);
        if (active) {
          // (lat2 - lat1) * (testLong - long1) / (long2 - long1) + lat1, in the reference's order
          const float t = __fadd_rn(
              __fdiv_rn(__fmul_rn(eDLat, __fsub_rn(testLong, eLong1)), __fsub_rn(eLong2, eLong1)), eLat1);
          if (pt.lat < t) {
#pragma unroll
            for (int w = 0; w < W; w++)
              if ((s >> 5) == w) acc[w] ^= 1u << (s & 31);
          }
        }
      }
    }
    if (i < n) {
      uint32_t words[W];
      uint32_t *mine = pred + static_cast<size_t>(i) * totalWords;
#pragma unroll
      for (int w = 0; w < W; w++) {
        words[w] = 0;
        if (w < totalWords) {
          uint32_t v = mine[w];
          if (pt.ok) {
            v ^= acc[w];
            if (acc[w]) mine[w] = v;
          } else if (firstEdge) {
            v = inOrOut ? 0u : 1u;
            mine[w] = v;
          }
          words[w] = v;
        }
      }
      const bool none = first_shape<W>(words, totalWords) < 0;
      keep[i] = (inOrOut != 0) == none ? 0 : 1;  // GeoRemoveFilter (query/geo_intersects.cu:214-226)
    }
  }
}

// ---------------------------------------------------------------------------------------------
// WriteGeoShapeDim: first shape of every entry that is inside some shape, compacted in order
// ---------------------------------------------------------------------------------------------
constexpr int kDimKPT = 4;
constexpr int kDimTile = kBlock * kDimKPT;

struct GeoDimParams {
  const uint32_t *pred;
  int totalWords;
  int n;
  int numTiles;
  uint8_t *values;
  uint8_t *nulls;
  unsigned int *ticket;
  uint32_t *error;
  uint64_t *status;
};

__global__ __launch_bounds__(kBlock) void geo_shape_dim_kernel(GeoDimParams p) {
  __shared__ uint32_t sWave[kBlock / 64];
  __shared__ uint32_t sTileExcl;
  __shared__ int sTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    if (threadIdx.x == 0) sTile = static_cast<int>(atomicAdd(p.ticket, 1u));
    __syncthreads();
    const int tile = sTile;
    if (tile >= p.numTiles) break;
    const int64_t first = static_cast<int64_t>(tile) * kDimTile + static_cast<int64_t>(threadIdx.x) * kDimKPT;
    int shapeOf[kDimKPT];
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < kDimKPT; j++) {
      const int64_t i = first + j;
      shapeOf[j] = -1;
      if (i < p.n) {
        const uint32_t *words = p.pred + static_cast<size_t>(i) * p.totalWords;
        for (int w = 0; w < p.totalWords; w++) {
          const uint32_t v = words[w];
          if (v) {
            shapeOf[j] = static_cast<int8_t>(w * 32 + __builtin_ctz(v));
            break;
          }
        }
      }
      mine += shapeOf[j] >= 0 ? 1u : 0u;
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    uint32_t waveBase = 0, tileSum = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) {
      if (w < wave) waveBase += sWave[w];
      tileSum += sWave[w];
    }
    if (wave == 0) {
      if (lane == 0) st_status(p.status + tile, (tile == 0 ? kFlagInclusive : kFlagAggregate) | tileSum);
      uint64_t excl = 0;
      if (tile > 0) {
        excl = lookback_wave(p.status, tile, lane, p.error);
        if (lane == 0) st_status(p.status + tile, kFlagInclusive | (excl + tileSum));
      }
      if (lane == 0) sTileExcl = static_cast<uint32_t>(excl);
    }
    __syncthreads();
    uint32_t at = sTileExcl + waveBase + (incl - mine);
#pragma unroll
    for (int j = 0; j < kDimKPT; j++)
      if (shapeOf[j] >= 0) {
        p.values[at] = static_cast<uint8_t>(shapeOf[j]);
        p.nulls[at] = 1;
        at++;
      }
    __syncthreads();
  }
}

int geo_batch_intersects(const GeoShapeBatch &shapes, const InputVector &points, uint32_t *indexVector, int n,
                         RecordID **recordIDVectors, int numForeignTables, uint32_t *outputPredicate, bool inOrOut,
                         hipStream_t stream, int device) {
  GeoPointsD P;
  memset(&P, 0, sizeof(P));
  std::vector<ForeignBatchD> hostBatches;
  std::unique_ptr<StreamBuffer> deviceBatches;
  if (points.Type == VectorPartyInput) {
    const VectorPartySlice &vp = points.Vector.VP;
    if (vp.DataType != GeoPoint) throw std::invalid_argument("only geo point column are allowed in geo_intersects");
    if (vp.BasePtr == nullptr) return 0;
    P.base = vp.BasePtr;
    P.valuesOff = vp.ValuesOffset;
    P.bitOff = vp.StartingIndex;
    P.index = indexVector;
  } else if (points.Type == ForeignColumnInput) {
    const ForeignColumnVector &f = points.Vector.ForeignVP;
    if (f.DataType != GeoPoint) throw std::invalid_argument("only geo point column are allowed in geo_intersects");
    P.foreign = 1;
    P.rids = f.RecordIDs;
    P.baseBatchID = f.BaseBatchID;
    P.numBatches = f.NumBatches;
    P.numRecLast = f.NumRecordsInLastBatch;
    P.defaultLat = f.DefaultValue.Value.GeoPointVal.Lat;
    P.defaultLong = f.DefaultValue.Value.GeoPointVal.Long;
    P.defaultOk = f.DefaultValue.HasDefault ? 1u : 0u;
    hostBatches.resize(static_cast<size_t>(f.NumBatches > 0 ? f.NumBatches : 0));
    for (int b = 0; b < f.NumBatches; b++) {  // the Batches array is host memory (Go slice)
      const VectorPartySlice &vp = f.Batches[b];
      hostBatches[b].base = vp.BasePtr;
      hostBatches[b].nullsOff = vp.NullsOffset;
      hostBatches[b].valuesOff = vp.ValuesOffset;
      hostBatches[b].bitOff = vp.StartingIndex;
      hostBatches[b].isConst = vp.BasePtr == nullptr;
    }
    deviceBatches.reset(new StreamBuffer(sizeof(ForeignBatchD) * hostBatches.size() + 16, stream));
    if (!hostBatches.empty())
      hip_check(hipMemcpyAsync(deviceBatches->get(), hostBatches.data(), sizeof(ForeignBatchD) * hostBatches.size(),
                               hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
    P.batches = deviceBatches->as<ForeignBatchD>();
  } else {
    throw std::invalid_argument("Unsupported data type for geo intersection contexts");
  }
  if (numForeignTables < 0 || numForeignTables > 8) throw std::invalid_argument("only support up to 8 foreign tables");
  invalidate_filter_journal(device, indexVector);  // the index vector is compacted by something a fused scan cannot replay
  if (n <= 0) return 0;
  const int N = shapes.TotalNumPoints, W = shapes.TotalWords;
  if (W > 8) throw std::invalid_argument("geo intersection supports up to 256 shapes");
  const float *lats = reinterpret_cast<const float *>(shapes.LatLongs);
  const float *longs = lats + N;
  const uint8_t *shapeIdx = shapes.LatLongs + static_cast<size_t>(N) * 8;
  StreamBuffer keep(static_cast<size_t>(n) + 16, stream);
  const int grid = capped_grid((static_cast<int64_t>(n) + kBlock - 1) / kBlock, 256 * 8);
  const int io = inOrOut ? 1 : 0;
  if (W <= 1)
    ARES_LAUNCH("geo_intersect_kernel<1>", geo_intersect_kernel<1>, grid, kBlock, stream, P, lats, longs, shapeIdx, N,
                outputPredicate, keep.as<uint8_t>(), n, W, io);
  else if (W <= 2)
    ARES_LAUNCH("geo_intersect_kernel<2>", geo_intersect_kernel<2>, grid, kBlock, stream, P, lats, longs, shapeIdx, N,
                outputPredicate, keep.as<uint8_t>(), n, W, io);
  else if (W <= 4)
    ARES_LAUNCH("geo_intersect_kernel<4>", geo_intersect_kernel<4>, grid, kBlock, stream, P, lats, longs, shapeIdx, N,
                outputPredicate, keep.as<uint8_t>(), n, W, io);
  else
    ARES_LAUNCH("geo_intersect_kernel<8>", geo_intersect_kernel<8>, grid, kBlock, stream, P, lats, longs, shapeIdx, N,
                outputPredicate, keep.as<uint8_t>(), n, W, io);
  return compact_by_predicate(keep.as<uint8_t>(), indexVector, recordIDVectors, numForeignTables, n, stream);
}

void write_geo_shape_dim(int shapeTotalWords, const DimensionOutputVector &dimOut, int n, uint32_t *outputPredicate,
                         hipStream_t stream) {
  if (n <= 0) return;
  GeoDimParams p;
  p.pred = outputPredicate;
  p.totalWords = static_cast<uint8_t>(shapeTotalWords);
  p.n = n;
  p.numTiles = (n + kDimTile - 1) / kDimTile;
  p.values = dimOut.DimValues;
  p.nulls = dimOut.DimNulls;
  StreamBuffer ws(16 + sizeof(uint64_t) * static_cast<size_t>(p.numTiles), stream);
  hip_check(hipMemsetAsync(ws.get(), 0, 16 + sizeof(uint64_t) * static_cast<size_t>(p.numTiles), stream), "hipMemsetAsync");
  p.ticket = ws.as<unsigned int>();
  p.error = ws.as<uint32_t>() + 1;
  p.status = reinterpret_cast<uint64_t *>(ws.as<uint8_t>() + 16);
  ARES_LAUNCH("geo_shape_dim_kernel", geo_shape_dim_kernel, capped_grid(p.numTiles), kBlock, stream, p);
}

}  // namespace

}  // namespace ares

using namespace ares;

extern "C" {

CGoCallResHandle GeoBatchIntersects(GeoShapeBatch geoShapeBatch, InputVector points, uint32_t *indexVector,
                                    int indexVectorLength, uint32_t startCount, RecordID **recordIDVectors,
                                    int numForeignTables, uint32_t *outputPredicate, bool inOrOut, void *cudaStream,
                                    int device) {
  ARES_ABI_BEGIN(device)
  materialize_index_vector(device, indexVector);
  (void)startCount;  // geo columns are never run-length decoded (query/geo_intersects.cu:160-163)
  if (indexVectorLength > 0) {
    const size_t n = static_cast<size_t>(indexVectorLength);
    mem_note_write(device, outputPredicate, 4 * n * (geoShapeBatch.TotalWords ? geoShapeBatch.TotalWords : 1));
    mem_note_write(device, indexVector, 4 * n);
    for (int t = 0; t < numForeignTables && t < 8; t++) mem_note_write(device, recordIDVectors[t], sizeof(RecordID) * n);
  }
  resHandle.res = int_result(geo_batch_intersects(geoShapeBatch, points, indexVector, indexVectorLength, recordIDVectors,
                                                  numForeignTables, outputPredicate, inOrOut,
                                                  reinterpret_cast<hipStream_t>(cudaStream), device));
  ARES_ABI_END("GeoBatchIntersects")
}

CGoCallResHandle WriteGeoShapeDim(int shapeTotalWords, DimensionOutputVector dimOut, int indexVectorLengthBeforeGeo,
                                  uint32_t *outputPredicate, void *cudaStream, int device) {
  ARES_ABI_BEGIN(device)
  grouped_note_write(device, dimOut.DimValues, static_cast<size_t>(indexVectorLengthBeforeGeo > 0 ? indexVectorLengthBeforeGeo : 1));
  if (indexVectorLengthBeforeGeo > 0) {
    mem_note_write(device, dimOut.DimValues, static_cast<size_t>(indexVectorLengthBeforeGeo));
    mem_note_write(device, dimOut.DimNulls, static_cast<size_t>(indexVectorLengthBeforeGeo));
  }
  write_geo_shape_dim(shapeTotalWords, dimOut, indexVectorLengthBeforeGeo, outputPredicate,
                      reinterpret_cast<hipStream_t>(cudaStream));
  ARES_ABI_END("WriteGeoShapeDim")
}

}  // extern "C"
