// HyperLogLog for MI355X (gfx950): the count-distinct aggregate of the batch pipeline.
//
// Reference: query/hll.cu:62-290 (sortCurrentBatch, reduceCurrentBatch, merge, makeHLLVector,
// copyDim), query/functor.hpp:1296-1374, query/iterator.hpp:1169-1257.  Per batch:
//   1. key = dim-row hash with its low 16 bits replaced by the register id; stable radix sort of
//      the batch's entries (sort_reduce.hip's one-sweep sort, payload = entry position);
//   2. runs of equal keys -> (key, first row, max value), appended behind the previous results;
//   3. stable merge of previous and current entries by (key ascending, value descending): one
//      merge-path partition per 2048-entry output tile, then every tile ranks its two input
//      segments against each other in LDS;
//   4. last batch: dimension / register head flags with one chained scan, registers per dimension,
//      byte offsets per dimension with a second chained scan, sparse (4 B per register) or dense
//      (16384 B) encoding; the two result buffers are allocated with libmem's deviceMalloc because
//      the host releases them with DeviceFree (query/time_series_aggregate.go:661-680);
//   5. gather of the surviving entries' dimension rows.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <string>

#include "ares_extensions.h"
#include "common.hpp"
#include "dim_layout.hpp"
#include "lookback.hpp"
#include "sort_reduce.hpp"

namespace ares {

namespace {

constexpr int kBlock = 256;

// HLLMergeComparator (query/functor.hpp:1316-1328)
__device__ __forceinline__ bool hll_less(uint64_t h1, uint32_t v1, uint64_t h2, uint32_t v2) {
  return h1 == h2 ? v1 > v2 : h1 < h2;
}

// ---------------------------------------------------------------------------------------------
// step 3: merge
// ---------------------------------------------------------------------------------------------
constexpr int kMergeTile = 2048;

// split[t] = number of entries of A among the first min(t * kMergeTile, nA + nB) outputs
__global__ __launch_bounds__(kBlock) void hll_merge_split_kernel(const uint64_t *hash, const uint32_t *values, int nA,
                                                                 int nB, int numTiles, int *split) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t > numTiles) return;
  const int64_t total = static_cast<int64_t>(nA) + nB;
  const int64_t diag = static_cast<int64_t>(t) * kMergeTile < total ? static_cast<int64_t>(t) * kMergeTile : total;
  const uint64_t *hb = hash + nA;
  const uint32_t *vb = values + nA;
  int lo = static_cast<int>(diag > nB ? diag - nB : 0), hi = static_cast<int>(diag < nA ? diag : nA);
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int64_t j = diag - 1 - mid;  // B entry that competes with A[mid] for the diagonal
    // ties go to A (a stable merge emits the first range first)
    if (!hll_less(hb[j], vb[j], hash[mid], values[mid])) lo = mid + 1; else hi = mid;
  }
  split[t] = lo;
}

__global__ __launch_bounds__(kBlock) void hll_merge_kernel(const uint64_t *hash, const uint32_t *values,
                                                           const uint32_t *index, int nA, int nB, const int *split,
                                                           uint64_t *hashOut, uint32_t *valuesOut, uint32_t *indexOut) {
  __shared__ uint64_t sHash[kMergeTile];
  __shared__ uint32_t sVal[kMergeTile];
  const int tile = blockIdx.x;
  const int64_t total = static_cast<int64_t>(nA) + nB;
  const int64_t outBase = static_cast<int64_t>(tile) * kMergeTile;
  const int count = static_cast<int>(total - outBase < kMergeTile ? total - outBase : kMergeTile);
  const int a0 = split[tile], a1 = split[tile + 1];
  const int b0 = static_cast<int>(outBase - a0);
  const int na = a1 - a0, nb = count - na;
  // LDS: [A segment][B segment]
  for (int i = threadIdx.x; i < count; i += kBlock) {
    const int64_t src = i < na ? a0 + i : static_cast<int64_t>(nA) + b0 + (i - na);
    sHash[i] = hash[src];
    sVal[i] = values[src];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < count; i += kBlock) {
    const uint64_t h = sHash[i];
    const uint32_t v = sVal[i];
    int rank;
    int64_t src;
    if (i < na) {  // entries of B strictly before this one
      int lo = 0, hi = nb;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (hll_less(sHash[na + mid], sVal[na + mid], h, v)) lo = mid + 1; else hi = mid;
      }
      rank = i + lo;
      src = a0 + i;
    } else {  // entries of A that are not after this one
      int lo = 0, hi = na;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (!hll_less(h, v, sHash[mid], sVal[mid])) lo = mid + 1; else hi = mid;
      }
      rank = (i - na) + lo;
      src = static_cast<int64_t>(nA) + b0 + (i - na);
    }
    hashOut[outBase + rank] = h;
    valuesOut[outBase + rank] = v;
    indexOut[outBase + rank] = index[src];
  }
}

// ---------------------------------------------------------------------------------------------
// step 4: the HLL vector
// ---------------------------------------------------------------------------------------------
constexpr int kHeadKPT = 8;
constexpr int kHeadTile = kBlock * kHeadKPT;
constexpr int kCountShift = 31;  // packed scan value: dimension heads << 31 | register heads

struct HeadParams {
  const uint64_t *hash;
  const uint32_t *index;
  int n;
  int numTiles;
  uint32_t *entryDim;   // [n] dimension number of every entry
  uint32_t *entryReg;   // [n] register heads before the entry
  uint32_t *regStart;   // [dims] register heads before the dimension's first entry
  uint32_t *headIndex;  // [dims] dimension row of the dimension's first entry
  unsigned int *ticket;
  uint32_t *totals;  // {dimensions, registers}
  uint32_t *error;
  uint64_t *status;
};

__global__ __launch_bounds__(kBlock) void hll_heads_kernel(HeadParams p) {
  __shared__ uint64_t sWave[kBlock / 64];
  __shared__ uint64_t sTileExcl;
  __shared__ int sTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    if (threadIdx.x == 0) sTile = static_cast<int>(atomicAdd(p.ticket, 1u));
    __syncthreads();
    const int tile = sTile;
    if (tile >= p.numTiles) break;
    const int64_t first = static_cast<int64_t>(tile) * kHeadTile + static_cast<int64_t>(threadIdx.x) * kHeadKPT;
    uint64_t h[kHeadKPT];
    uint64_t prev = 0;
    if (first > 0 && first < p.n) prev = p.hash[first - 1];
#pragma unroll
    for (int j = 0; j < kHeadKPT; j++) h[j] = first + j < p.n ? p.hash[first + j] : 0;
    uint32_t dimHeads = 0, regHeads = 0;
#pragma unroll
    for (int j = 0; j < kHeadKPT; j++) {
      const int64_t i = first + j;
      const uint64_t before = j == 0 ? prev : h[j - 1];
      if (i < p.n) {
        dimHeads |= static_cast<uint32_t>(i == 0 || (h[j] >> 16) != (before >> 16)) << j;
        regHeads |= static_cast<uint32_t>(i == 0 || h[j] != before) << j;
      }
    }
    const uint64_t mine = (static_cast<uint64_t>(__popc(dimHeads)) << kCountShift) | __popc(regHeads);
    uint64_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint64_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    uint64_t waveBase = 0, tileSum = 0;
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
      if (lane == 0) {
        sTileExcl = excl;
        if (tile == p.numTiles - 1) {
          const uint64_t all = excl + tileSum;
          p.totals[0] = static_cast<uint32_t>(all >> kCountShift);
          p.totals[1] = static_cast<uint32_t>(all & ((1ull << kCountShift) - 1));
        }
      }
    }
    __syncthreads();
    const uint64_t start = sTileExcl + waveBase + (incl - mine);
    uint32_t dim = static_cast<uint32_t>(start >> kCountShift);  // dimension heads before the lane's first entry
    uint32_t reg = static_cast<uint32_t>(start & ((1ull << kCountShift) - 1));
#pragma unroll
    for (int j = 0; j < kHeadKPT; j++) {
      const int64_t i = first + j;
      if (i < p.n) {
        if ((dimHeads >> j) & 1u) {
          p.regStart[dim] = reg;
          p.headIndex[dim] = p.index[i];
          dim++;
        }
        p.entryDim[i] = dim - 1;
        p.entryReg[i] = reg;
        reg += (regHeads >> j) & 1u;
      }
    }
    __syncthreads();
  }
}

// registers per dimension (truncated to 16 bits like the reference's output vector), encoded size
// per dimension (HLLDimByteCountFunctor, query/functor.hpp:1331-1340) and its exclusive scan
struct DimScanParams {
  const uint32_t *regStart;
  int dims;
  uint32_t registers;
  int numTiles;
  uint16_t *regCount;
  uint64_t *offsets;  // [dims + 1]
  unsigned int *ticket;
  uint32_t *error;
  uint64_t *status;
};

__global__ __launch_bounds__(kBlock) void hll_dim_scan_kernel(DimScanParams p) {
  __shared__ uint64_t sWave[kBlock / 64];
  __shared__ uint64_t sTileExcl;
  __shared__ int sTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    if (threadIdx.x == 0) sTile = static_cast<int>(atomicAdd(p.ticket, 1u));
    __syncthreads();
    const int tile = sTile;
    if (tile >= p.numTiles) break;
    const int64_t d = static_cast<int64_t>(tile) * kBlock + threadIdx.x;
    uint64_t bytes = 0;
    if (d < p.dims) {
      const uint32_t next = d + 1 < p.dims ? p.regStart[d + 1] : p.registers;
      const uint16_t count = static_cast<uint16_t>(next - p.regStart[d]);
      p.regCount[d] = count;
      bytes = count < HLL_DENSE_THRESHOLD ? static_cast<uint64_t>(count) * 4 : static_cast<uint64_t>(HLL_DENSE_SIZE);
    }
    uint64_t incl = bytes;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint64_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    uint64_t waveBase = 0, tileSum = 0;
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
      if (lane == 0) {
        sTileExcl = excl;
        if (tile == p.numTiles - 1) p.offsets[p.dims] = excl + tileSum;
      }
    }
    __syncthreads();
    if (d < p.dims) p.offsets[d] = sTileExcl + waveBase + (incl - bytes);
    __syncthreads();
  }
}

// CopyHLLFunctor (query/functor.hpp:1351-1374) over HLLValueOutputIterator (query/iterator.hpp:1197-1257)
__global__ __launch_bounds__(kBlock) void hll_write_kernel(const uint64_t *hash, const uint32_t *values, int n,
                                                           const uint32_t *entryDim, const uint32_t *entryReg,
                                                           const uint32_t *regStart, const uint16_t *regCount,
                                                           const uint64_t *offsets, uint8_t *hllVector) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint64_t h = hash[i];
    if (i > 0 && hash[i - 1] == h) continue;  // not the first (= largest) value of its register
    const uint32_t dim = entryDim[i];
    const uint32_t value = values[i];
    const uint32_t regID = value & 0x3FFFu;
    const uint32_t rho = (((value >> 16) & 0xFFu) + 1u) & 0xFFu;  // uint8 arithmetic in the reference
    const uint64_t off = offsets[dim];
    if (regCount[dim] < HLL_DENSE_THRESHOLD)
      *reinterpret_cast<uint32_t *>(hllVector + off + static_cast<uint64_t>(entryReg[i] - regStart[dim]) * 4) =
          rho << 16 | regID;
    else
      hllVector[off + regID] = static_cast<uint8_t>(rho);
  }
}

// step 5: copyDim (query/hll.cu:169-187) — both strides are the input capacity
__global__ __launch_bounds__(kBlock) void hll_gather_dims_kernel(const uint8_t *dimIn, uint8_t *dimOut, DimLayoutD L,
                                                                 size_t capacity, const uint32_t *index, int n) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    copy_dim_row(dimIn, capacity, dimOut, capacity, L, index[i], static_cast<uint32_t>(i));
}

// ---------------------------------------------------------------------------------------------
// result buffers come from libmem.so: the host frees them with DeviceFree
// ---------------------------------------------------------------------------------------------
using DeviceMallocFn = CGoCallResHandle (*)(void **, size_t);

DeviceMallocFn libmem_device_malloc() {
  static const DeviceMallocFn fn = []() -> DeviceMallocFn {
    // the libmem.so that sits next to this library (the reference's lib/ directory layout) ...
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(&AresFlushDeferred), &info) && info.dli_fname) {
      std::string path(info.dli_fname);
      const size_t slash = path.rfind('/');
      path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/libmem.so";
      if (void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL))
        if (void *s = dlsym(h, "deviceMalloc")) return reinterpret_cast<DeviceMallocFn>(s);
    }
    // ... else whatever the host linked (-lalgorithm -lmem puts the symbol in the global scope)
    return reinterpret_cast<DeviceMallocFn>(dlsym(RTLD_DEFAULT, "deviceMalloc"));
  }();
  return fn;
}

void *result_alloc(size_t bytes) {
  const DeviceMallocFn fn = libmem_device_malloc();
  if (!fn) throw AlgorithmError("HyperLogLog: libmem.so (deviceMalloc) not found next to libalgorithm.so");
  void *p = nullptr;
  const CGoCallResHandle h = fn(&p, bytes ? bytes : 1);
  if (h.pStrErr) {
    std::string msg(h.pStrErr);
    free(const_cast<char *>(h.pStrErr));
    throw AlgorithmError(msg);
  }
  return p;
}

int make_hll_vector(const DimensionVector &cur, const uint32_t *curValues, int n, uint8_t **hllVectorPtr,
                    size_t *hllVectorSizePtr, uint16_t **regCountPtr, hipStream_t stream) {
  const int numTiles = (n + kHeadTile - 1) / kHeadTile;
  // workspace: [ticket, totals[2], error | ticket2, error2, pad][status][entryDim][entryReg][regStart][headIndex]
  const size_t offStatus = 64, statusBytes = sizeof(uint64_t) * static_cast<size_t>(numTiles);
  const size_t offEntryDim = (offStatus + statusBytes + 255) & ~size_t(255);
  const size_t vec = sizeof(uint32_t) * static_cast<size_t>(n);
  StreamBuffer ws(offEntryDim + 4 * vec + 256, stream);
  uint8_t *base = ws.as<uint8_t>();
  hip_check(hipMemsetAsync(base, 0, offStatus + statusBytes, stream), "hipMemsetAsync");
  HeadParams hp;
  hp.hash = cur.HashValues;
  hp.index = cur.IndexVector;
  hp.n = n;
  hp.numTiles = numTiles;
  hp.entryDim = reinterpret_cast<uint32_t *>(base + offEntryDim);
  hp.entryReg = hp.entryDim + n;
  hp.regStart = hp.entryReg + n;
  hp.headIndex = hp.regStart + n;
  hp.ticket = reinterpret_cast<unsigned int *>(base);
  hp.totals = reinterpret_cast<uint32_t *>(base) + 1;
  hp.error = reinterpret_cast<uint32_t *>(base) + 3;
  hp.status = reinterpret_cast<uint64_t *>(base + offStatus);
  ARES_LAUNCH("hll_heads_kernel", hll_heads_kernel, capped_grid(numTiles), kBlock, stream, hp);
  uint32_t totals[3] = {0, 0, 0};  // {dimensions, registers, error}
  read_back_u32(hp.totals, totals, 3, stream);
  if (totals[2]) throw AlgorithmError("ERROR: HyperLogLog: inter-tile scan timed out");
  const int dims = static_cast<int>(totals[0]);

  uint16_t *regCount = static_cast<uint16_t *>(result_alloc(sizeof(uint16_t) * static_cast<size_t>(dims)));
  *regCountPtr = regCount;
  const int dimTiles = (dims + kBlock - 1) / kBlock;
  StreamBuffer ws2(64 + sizeof(uint64_t) * (static_cast<size_t>(dimTiles) + dims + 1), stream);
  uint8_t *base2 = ws2.as<uint8_t>();
  hip_check(hipMemsetAsync(base2, 0, 64 + sizeof(uint64_t) * static_cast<size_t>(dimTiles), stream), "hipMemsetAsync");
  DimScanParams dp;
  dp.regStart = hp.regStart;
  dp.dims = dims;
  dp.registers = totals[1];
  dp.numTiles = dimTiles;
  dp.regCount = regCount;
  dp.ticket = reinterpret_cast<unsigned int *>(base2);
  dp.error = reinterpret_cast<uint32_t *>(base2) + 1;
  dp.status = reinterpret_cast<uint64_t *>(base2 + 64);
  dp.offsets = dp.status + dimTiles;
  ARES_LAUNCH("hll_dim_scan_kernel", hll_dim_scan_kernel, capped_grid(dimTiles), kBlock, stream, dp);
  uint32_t tail[2] = {0, 0};
  read_back_u32(reinterpret_cast<const uint32_t *>(dp.offsets + dims), tail, 2, stream);
  uint32_t err = 0;
  read_back_u32(dp.error, &err, 1, stream);
  if (err) throw AlgorithmError("ERROR: HyperLogLog: inter-tile scan timed out");
  const size_t total = static_cast<size_t>(tail[0]) | (static_cast<size_t>(tail[1]) << 32);

  uint8_t *hllVector = static_cast<uint8_t *>(result_alloc(total));
  *hllVectorPtr = hllVector;
  *hllVectorSizePtr = total;
  hip_check(hipMemsetAsync(hllVector, 0, total, stream), "hipMemsetAsync");
  ARES_LAUNCH("hll_write_kernel", hll_write_kernel, capped_grid((static_cast<int64_t>(n) + kBlock - 1) / kBlock, 256 * 8),
              kBlock, stream, cur.HashValues, curValues, n, hp.entryDim, hp.entryReg, hp.regStart, regCount, dp.offsets,
              hllVector);
  // thrust::remove_if on the dimension heads (query/hll.cu:240-245)
  hip_check(hipMemcpyAsync(cur.IndexVector, hp.headIndex, sizeof(uint32_t) * static_cast<size_t>(dims),
                           hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync");
  return dims;
}

int hyperloglog(const DimensionVector &prev, const DimensionVector &cur, uint32_t *prevValues, uint32_t *curValues,
                int prevResultSize, int curBatchSize, bool isLastBatch, uint8_t **hllVectorPtr,
                size_t *hllVectorSizePtr, uint16_t **regCountPtr, hipStream_t stream) {
  if (prevResultSize < 0 || curBatchSize < 0) throw std::invalid_argument("HyperLogLog: negative size");
  if (static_cast<int64_t>(prevResultSize) + curBatchSize >= (1ll << 30))
    throw std::invalid_argument("HyperLogLog supports up to 2^30 - 1 entries per call");
  const DimLayoutD L = make_dim_layout(cur.NumDimsPerDimWidth);
  const size_t capacity = static_cast<size_t>(cur.VectorCapacity);
  const int P = prevResultSize, n = curBatchSize;

  int runs = 0;
  if (n > 0) {
    StreamBuffer positions(sizeof(uint32_t) * static_cast<size_t>(n), stream);
    sort_rows(prev.DimValues, L, capacity, cur.IndexVector, curValues, cur.HashValues, positions.as<uint32_t>(), true,
              n, stream);
    runs = hll_reduce_sorted(cur.HashValues, positions.as<uint32_t>(), cur.IndexVector, curValues, prev.HashValues + P,
                             prev.IndexVector + P, prevValues + P, n, stream);
  }
  int resSize = P + runs;
  if (resSize > 0) {
    const int tiles = (resSize + kMergeTile - 1) / kMergeTile;
    StreamBuffer split(sizeof(int) * (static_cast<size_t>(tiles) + 1), stream);
    ARES_LAUNCH("hll_merge_split_kernel", hll_merge_split_kernel, (tiles + 1 + kBlock - 1) / kBlock, kBlock, stream,
                prev.HashValues, prevValues, P, runs, tiles, split.as<int>());
    ARES_LAUNCH("hll_merge_kernel", hll_merge_kernel, tiles, kBlock, stream, prev.HashValues, prevValues,
                prev.IndexVector, P, runs, split.as<int>(), cur.HashValues, curValues, cur.IndexVector);
    if (isLastBatch)
      resSize = make_hll_vector(cur, curValues, resSize, hllVectorPtr, hllVectorSizePtr, regCountPtr, stream);
    ARES_LAUNCH("hll_gather_dims_kernel", hll_gather_dims_kernel,
                capped_grid((static_cast<int64_t>(resSize) + kBlock - 1) / kBlock, 256 * 8), kBlock, stream, prev.DimValues,
                cur.DimValues, make_dim_layout(prev.NumDimsPerDimWidth), static_cast<size_t>(prev.VectorCapacity),
                cur.IndexVector, resSize);
  }
  return resSize;
}

}  // namespace

}  // namespace ares

using namespace ares;

extern "C" CGoCallResHandle HyperLogLog(DimensionVector prevDimOut, DimensionVector curDimOut, uint32_t *prevValuesOut,
                                        uint32_t *curValuesOut, int prevResultSize, int curBatchSize, bool isLastBatch,
                                        uint8_t **hllVectorPtr, size_t *hllVectorSizePtr,
                                        uint16_t **hllDimRegIDCountPtr, void *cudaStream, int device) {
  ARES_ABI_BEGIN(device)
  settle_dimension_vector(device, prevDimOut);
  settle_dimension_vector(device, curDimOut);
  {
    const size_t capV = static_cast<size_t>(prevDimOut.VectorCapacity > curDimOut.VectorCapacity ? prevDimOut.VectorCapacity : curDimOut.VectorCapacity);
    materialize_fills_for_read(device, prevValuesOut, 4 * capV);
    materialize_fills_for_read(device, curValuesOut, 4 * capV);
  }
  flush_deferred_for_vector(device, prevDimOut, nullptr, 0);
  flush_deferred_for_vector(device, curDimOut, nullptr, 0);
  grouped_note_write(device, prevDimOut);
  grouped_note_write(device, curDimOut);
  {  // the sort / run-max / merge stages rewrite both vectors, their hash and index vectors and both value vectors
    mem_note_vector_all(device, prevDimOut);
    mem_note_vector_all(device, curDimOut);
    const size_t cap = static_cast<size_t>(prevDimOut.VectorCapacity > curDimOut.VectorCapacity ? prevDimOut.VectorCapacity
                                                                                                    : curDimOut.VectorCapacity);
    mem_note_write(device, prevValuesOut, 4 * cap);
    mem_note_write(device, curValuesOut, 4 * cap);
  }
  resHandle.res = reinterpret_cast<void *>(static_cast<intptr_t>(
      hyperloglog(prevDimOut, curDimOut, prevValuesOut, curValuesOut, prevResultSize, curBatchSize, isLastBatch,
                  hllVectorPtr, hllVectorSizePtr, hllDimRegIDCountPtr, reinterpret_cast<hipStream_t>(cudaStream))));
  ARES_ABI_END("HyperLogLog")
}
