// Sort, Reduce and Expand for MI355X (gfx950): the sort-based group-by path.
//
// Reference: query/sort_reduce.cu:118-133 (hash rows + thrust::stable_sort_by_key),
// :135-249 (thrust::reduce_by_key + permuted dim gather), :252-314 (Expand).
//
// Sort   = one fused "hash rows + all eight digit histograms" pass, then eight passes of a
//          single-pass LSD radix sort ("onesweep"): each tile ranks its keys with wavefront
//          ballot matching (stable), publishes its 256-bin histogram through the chained-scan
//          protocol of lookback.hpp and scatters through LDS so that global writes are runs of
//          consecutive addresses.  64-bit keys are sorted on all 64 bits: the order of the output
//          is observable (Reduce emits groups in ascending hash order).
// Reduce = one pass: head flags from neighbouring hashes, chained scan of the head counts for the
//          group numbering, in-register folds per lane, a wavefront segmented scan for runs that
//          span lanes, atomics only for runs that span wavefronts.
#include <hip/hip_runtime.h>

#include "aggregate.hpp"
#include "common.hpp"
#include "device_model.hpp"
#include "dim_layout.hpp"
#include "lookback.hpp"
#include "sort_reduce_fused.hpp"

namespace ares {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

// ---------------------------------------------------------------------------------------------
// Sort step 1: hashes + global digit histograms
// ---------------------------------------------------------------------------------------------
// hllValues != nullptr: the key is the HyperLogLog sort key (query/functor.hpp:1296-1305): the dim
// hash with its low 16 bits replaced by the register id of entry i.  iotaOut != nullptr: the
// payload that travels with the keys is the entry's position.
__global__ __launch_bounds__(kBlock) void sort_hash_hist_kernel(const uint8_t *dimValues, DimLayoutD L, size_t capacity,
                                                                const uint32_t *indexVector, uint64_t *hashes, int n,
                                                                uint32_t *globalHist /* [8][256] */,
                                                                const uint32_t *hllValues, uint32_t *iotaOut) {
  __shared__ uint32_t sHist[8 * 256];
  for (int i = threadIdx.x; i < 8 * 256; i += kBlock) sHist[i] = 0;
  __syncthreads();
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    Murmur128Stream ms(0);
    hash_dim_row(ms, dimValues, L, capacity, indexVector[i]);
    uint64_t h = ms.finish();
    if (hllValues) h = (h & 0xFFFFFFFFFFFF0000ull) | (hllValues[i] & 0x3FFFu);
    if (iotaOut) iotaOut[i] = static_cast<uint32_t>(i);
    hashes[i] = h;
#pragma unroll
    for (int p = 0; p < 8; p++) atomicAdd(&sHist[p * 256 + ((h >> (8 * p)) & 255)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 256; i += kBlock) {
    const uint32_t c = sHist[i];
    if (c) atomicAdd(&globalHist[i], c);
  }
}

// exclusive scan of each pass's 256 bins: digitStart[p][d] = number of keys with a smaller digit
__global__ __launch_bounds__(256) void digit_start_kernel(uint32_t *hist /* in: counts, out: starts */) {
  __shared__ uint32_t s[256];
  uint32_t *h = hist + blockIdx.x * 256;
  const uint32_t c = h[threadIdx.x];
  s[threadIdx.x] = c;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
    __syncthreads();
    s[threadIdx.x] += t;
    __syncthreads();
  }
  h[threadIdx.x] = s[threadIdx.x] - c;
}

// ---------------------------------------------------------------------------------------------
// Sort step 2: one LSD pass (8-bit digit), stable, single pass over the data
// ---------------------------------------------------------------------------------------------
// (keys per lane is a template parameter of the pass: 16 by default — a 4096-key tile, 1024 contiguous keys per wavefront)
constexpr uint32_t kFlagAgg32 = 1u << 30, kFlagInc32 = 2u << 30, kFlagMask32 = 3u << 30;

__device__ __forceinline__ uint32_t ld_status32(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status32(uint32_t *p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int KPT>
__global__ __launch_bounds__(kBlock) void radix_pass_kernel(const uint64_t *keysIn, const uint32_t *valsIn,
                                                            uint64_t *keysOut, uint32_t *valsOut, int n, int shift,
                                                            const uint32_t *digitStart, unsigned int *ticket,
                                                            uint32_t *error, uint32_t *status /* [numTiles][256] */,
                                                            int numTiles) {
  __shared__ uint64_t sKeys[(kBlock * KPT)];
  __shared__ uint32_t sVals[(kBlock * KPT)];
  __shared__ uint32_t sHist[kWaves][256];  // per-wave digit counts -> per-wave bases inside the tile
  __shared__ uint32_t sTileStart[256];     // first tile-local slot of each digit
  __shared__ uint32_t sBase[256];          // global position = sBase[digit] + tile-local slot
  __shared__ uint32_t sWaveTotals[kWaves];
  __shared__ int sTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t ltMask = (1ull << lane) - 1;
  for (;;) {
    if (threadIdx.x == 0) sTile = static_cast<int>(atomicAdd(ticket, 1u));
    for (int i = threadIdx.x; i < kWaves * 256; i += kBlock) (&sHist[0][0])[i] = 0;
    __syncthreads();
    const int tile = sTile;
    if (tile >= numTiles) break;
    const int64_t tileBase = static_cast<int64_t>(tile) * (kBlock * KPT);
    const int tileCount = static_cast<int>(n - tileBase < (kBlock * KPT) ? n - tileBase : (kBlock * KPT));

    uint64_t key[KPT];
    uint32_t val[KPT];
    uint16_t rank[KPT];
#pragma unroll
    for (int r = 0; r < KPT; r++) {
      const int local = wave * (64 * KPT) + r * 64 + lane;
      if (local < tileCount) {
        key[r] = keysIn[tileBase + local];
        val[r] = valsIn[tileBase + local];
      } else {
        key[r] = ~0ull;
        val[r] = 0;
      }
    }
    // stable ranking inside the wavefront's chunk: lanes holding the same digit find each other
    // with 8 ballots; the per-wave histogram row lives in LDS and is only touched by this wave
#pragma unroll
    for (int r = 0; r < KPT; r++) {
      const int local = wave * (64 * KPT) + r * 64 + lane;
      const bool valid = local < tileCount;
      const uint32_t digit = static_cast<uint32_t>(key[r] >> shift) & 255u;
      uint64_t peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const uint64_t m = __ballot((digit >> b) & 1u);
        peers &= ((digit >> b) & 1u) ? m : ~m;
      }
      const uint32_t before = __popcll(peers & ltMask);
      uint32_t old = 0;
      if (valid) old = sHist[wave][digit];
      __builtin_amdgcn_wave_barrier();
      if (valid && before == 0) sHist[wave][digit] = old + __popcll(peers);
      __builtin_amdgcn_wave_barrier();
      rank[r] = static_cast<uint16_t>(old + before);
    }
    __syncthreads();
    // thread d owns digit d: bases of the waves inside the tile, tile histogram, chained scan
    const int d = threadIdx.x;
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) {
      const uint32_t c = sHist[w][d];
      sHist[w][d] = total;
      total += c;
    }
    st_status32(status + static_cast<size_t>(tile) * 256 + d, (tile == 0 ? kFlagInc32 : kFlagAgg32) | total);
    // exclusive scan of the tile histogram over the 256 digits
    uint32_t incl = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) sWaveTotals[wave] = incl;
    __syncthreads();
    uint32_t waveBase = 0;
    for (int w = 0; w < wave; w++) waveBase += sWaveTotals[w];
    const uint32_t tileStart = waveBase + incl - total;
    sTileStart[d] = tileStart;
    uint32_t exclusive = 0;
    if (tile > 0) {
      for (int t = tile - 1; t >= 0; --t) {
        uint32_t w = ld_status32(status + static_cast<size_t>(t) * 256 + d);
        uint32_t spins = 0;
        while ((w & kFlagMask32) == 0) {
          __builtin_amdgcn_s_sleep(2);
          w = ld_status32(status + static_cast<size_t>(t) * 256 + d);
          if (++spins > kMaxSpins) {
            atomicOr(error, 1u);
            w = kFlagInc32;
          }
        }
        exclusive += w & ~kFlagMask32;
        if ((w & kFlagMask32) == kFlagInc32) break;
      }
      st_status32(status + static_cast<size_t>(tile) * 256 + d, kFlagInc32 | (exclusive + total));
    }
    sBase[d] = digitStart[d] + exclusive - tileStart;
    __syncthreads();
    // reorder through LDS: tile-local slot = digit start + wave base + rank in wave
#pragma unroll
    for (int r = 0; r < KPT; r++) {
      const int local = wave * (64 * KPT) + r * 64 + lane;
      if (local < tileCount) {
        const uint32_t digit = static_cast<uint32_t>(key[r] >> shift) & 255u;
        const uint32_t slot = sTileStart[digit] + sHist[wave][digit] + rank[r];
        sKeys[slot] = key[r];
        sVals[slot] = val[r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < KPT; m++) {
      const int j = m * kBlock + threadIdx.x;
      if (j < tileCount) {
        const uint64_t k = sKeys[j];
        const uint32_t digit = static_cast<uint32_t>(k >> shift) & 255u;
        const uint32_t dst = sBase[digit] + static_cast<uint32_t>(j);
        keysOut[dst] = k;
        valsOut[dst] = sVals[j];
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Sort step 3: fix-up after sorting the TOP 32 bits only
// ---------------------------------------------------------------------------------------------
// The order of Reduce's output — ascending 64-bit hash — is observable, so the sort has to be by all 64 bits.  With
// (pseudo-)random hashes the top 32 bits already decide the order of nearly every pair (of 50 M distinct hashes about
// 50e6^2 / 2 / 2^32 = 290 k pairs share their top half): four stable passes over bits 32..63, then
//   detect   one thread per element i: a DESCENT (same top half as element i - 1, smaller low half) means its segment —
//            the run of equal top halves — is out of order.  The thread of a segment's FIRST descent walks to the
//            segment's ends (bounded by kFixupMaxRun) and lists [start, end).  Segments without a descent — every run of
//            equal keys however long: a hot group — cost one comparison per element and are never walked;
//   sort     one thread per listed segment: stable insertion sort by the low half, in place (segments are disjoint).
// Equal top halves keep their input order through the four passes and equal keys theirs through the insertion sort:
// the result is the stable sort by the whole key.  A listed segment longer than kFixupMaxRun, or a full list, raises
// `fallback`: the caller runs all eight passes over the data as it is (a stable permutation of the input, so the
// result is the same).  Logic first checked on the CPU (tools/prototypes/sort_topbits_fixup.hpp, tests/test_prototypes.py).
constexpr int kFixupMaxRun = 64;
struct FixupSegment {
  uint32_t start, end;
};
// The list is kept per WORKGROUP of the detect kernel ([workgroup][capPerGroup] + one counter each): C4's 50 M distinct
// hashes list ~290 k segments per call, and that many returning atomics on ONE counter took 2.3 ms — a single address
// sustains < 100 of them per microsecond (tools/ubench_atomics.hip) — against 0.3 ms for everything else in the kernel.
// LONG segments (more than kFixupMaxRun entries, out of order): two groups whose hashes share their top half, each with
// many rows — 200 k groups over 64 Mi rows: ~5 such pairs per call, 670 entries each.  Too long for one thread's
// insertion sort, far too few to justify four more passes over everything (the first version's fallback: 12.9 instead
// of 10.4 ms on that configuration).  The thread that meets such a segment finds its ends by binary search (the data is
// sorted by top half), claims it in a small table keyed by its start — several descents of one segment meet there — and a
// workgroup sorts it in LDS by ranking (sort_fixup_long_kernel).  Only a segment beyond kFixupLongMax entries, or a full
// table, still raises `fallback`.
constexpr int kFixupLongMax = 4096;
constexpr int kFixupLongSlots = 1024;  // claim table (open addressing, power of two); at most half of it is used
struct FixupParams {
  uint64_t *keys;
  uint32_t *vals;
  int n;
  FixupSegment *work;     // [groups][capPerGroup]
  uint32_t *groupCounts;  // [groups], zeroed; may exceed capPerGroup (then fallback is set)
  uint32_t capPerGroup;
  uint32_t *fallback;     // zeroed
  uint32_t *longTable;    // [kFixupLongSlots] segment start + 1 (0 = free), zeroed
  FixupSegment *longWork; // [kFixupLongSlots / 2]
  uint32_t *longCount;    // zeroed
};

// the run of equal top halves around position i of an array sorted by top half: [start, end)
__device__ __forceinline__ FixupSegment fixup_bounds(const uint64_t *keys, int n, int i) {
  const uint32_t top = static_cast<uint32_t>(keys[i] >> 32);
  int lo = 0, hi = i;  // first position whose top half is >= top
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (static_cast<uint32_t>(keys[mid] >> 32) < top) lo = mid + 1; else hi = mid;
  }
  const int start = lo;
  lo = i + 1;
  hi = n;  // first position whose top half is > top
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (static_cast<uint32_t>(keys[mid] >> 32) <= top) lo = mid + 1; else hi = mid;
  }
  return FixupSegment{static_cast<uint32_t>(start), static_cast<uint32_t>(lo)};
}

// a descent in a segment that one thread cannot walk: list it once
__device__ __forceinline__ void fixup_claim_long(const FixupParams &p, int i) {
  const FixupSegment seg = fixup_bounds(p.keys, p.n, i);
  if (seg.end - seg.start > static_cast<uint32_t>(kFixupLongMax)) {
    *p.fallback = 1u;
    return;
  }
  const uint32_t tag = seg.start + 1u;
  uint32_t slot = (seg.start * 2654435761u) >> 22;  // 10 bits
  for (int probe = 0; probe < kFixupLongSlots; probe++, slot = (slot + 1u) & (kFixupLongSlots - 1)) {
    const uint32_t seen = atomicCAS(p.longTable + slot, 0u, tag);
    if (seen == tag) return;  // another descent of this segment was here first
    if (seen == 0u) {
      const uint32_t at = atomicAdd(p.longCount, 1u);
      if (at >= static_cast<uint32_t>(kFixupLongSlots / 2)) *p.fallback = 1u;
      else p.longWork[at] = seg;
      return;
    }
  }
  *p.fallback = 1u;
}

__global__ __launch_bounds__(kBlock) void sort_fixup_detect_kernel(FixupParams p) {
  for (int64_t i64 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i64 < p.n; i64 += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int i = static_cast<int>(i64);
    if (i == 0) continue;
    const uint64_t k = p.keys[i], before = p.keys[i - 1];
    if ((k >> 32) != (before >> 32) || static_cast<uint32_t>(k) >= static_cast<uint32_t>(before)) continue;
    const uint32_t top = static_cast<uint32_t>(k >> 32);
    // the first descent of the segment lists it: walk back to the segment's start, giving up at an earlier descent
    int s = i - 1;
    bool mine = true;
    while (s > 0 && static_cast<uint32_t>(p.keys[s - 1] >> 32) == top) {
      if (static_cast<uint32_t>(p.keys[s]) < static_cast<uint32_t>(p.keys[s - 1])) {
        mine = false;
        break;
      }
      s--;
      if (i - s > kFixupMaxRun) {
        fixup_claim_long(p, i);
        mine = false;
        break;
      }
    }
    if (!mine) continue;
    int e = i + 1;
    while (e < p.n && static_cast<uint32_t>(p.keys[e] >> 32) == top) {
      e++;
      if (e - s > kFixupMaxRun) {
        fixup_claim_long(p, i);
        mine = false;
        break;
      }
    }
    if (!mine) continue;
    const uint32_t slot = atomicAdd(p.groupCounts + blockIdx.x, 1u);
    if (slot >= p.capPerGroup) {
      *p.fallback = 1u;
      continue;
    }
    p.work[static_cast<size_t>(blockIdx.x) * p.capPerGroup + slot] = FixupSegment{static_cast<uint32_t>(s), static_cast<uint32_t>(e)};
  }
}

// one workgroup per list of the detect kernel (same grid)
__global__ __launch_bounds__(64) void sort_fixup_sort_kernel(FixupParams p) {
  const uint32_t listed = p.groupCounts[blockIdx.x];
  const uint32_t count = listed < p.capPerGroup ? listed : p.capPerGroup;
  for (uint32_t w = threadIdx.x; w < count; w += 64) {
    const FixupSegment seg = p.work[static_cast<size_t>(blockIdx.x) * p.capPerGroup + w];
    for (uint32_t a = seg.start + 1; a < seg.end; a++) {
      const uint64_t k = p.keys[a];
      const uint32_t v = p.vals[a];
      uint32_t b = a;
      while (b > seg.start && static_cast<uint32_t>(p.keys[b - 1]) > static_cast<uint32_t>(k)) {  // strict: equal keys keep their order
        p.keys[b] = p.keys[b - 1];
        p.vals[b] = p.vals[b - 1];
        b--;
      }
      p.keys[b] = k;
      p.vals[b] = v;
    }
  }
}

// one workgroup per listed long segment: stable sort by the low half through ranks computed in LDS
__global__ __launch_bounds__(kBlock) void sort_fixup_long_kernel(FixupParams p) {
  __shared__ uint32_t sLow[kFixupLongMax];
  __shared__ uint32_t sVal[kFixupLongMax];
  const uint32_t listed = *p.longCount;
  const uint32_t count = listed < static_cast<uint32_t>(kFixupLongSlots / 2) ? listed : static_cast<uint32_t>(kFixupLongSlots / 2);
  for (uint32_t w = blockIdx.x; w < count; w += gridDim.x) {
    const FixupSegment seg = p.longWork[w];
    const int len = static_cast<int>(seg.end - seg.start);
    const uint64_t top = p.keys[seg.start] & 0xFFFFFFFF00000000ull;
    __syncthreads();  // (LDS of the previous segment)
    for (int i = threadIdx.x; i < len; i += kBlock) {
      sLow[i] = static_cast<uint32_t>(p.keys[seg.start + i]);
      sVal[i] = p.vals[seg.start + i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < len; i += kBlock) {
      const uint32_t mine = sLow[i];
      int rank = 0;
      for (int j = 0; j < len; j++) {
        const uint32_t other = sLow[j];
        rank += (other < mine || (other == mine && j < i)) ? 1 : 0;
      }
      p.keys[seg.start + rank] = top | mine;
      p.vals[seg.start + rank] = sVal[i];
    }
  }
}

__global__ __launch_bounds__(kBlock) void fill_u64_kernel(uint64_t *p, uint64_t v, int n) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    p[i] = v;
}

// no dimensions: every row is the empty row, one constant hash (and nothing to sort)
__global__ __launch_bounds__(kBlock) void sort_const_hash_kernel(uint64_t *hashes, int n) {
  Murmur128Stream ms(0);
  const uint64_t h = ms.finish();
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    hashes[i] = h;
}

void sort_rows(const uint8_t *dimValues, const DimLayoutD &L, size_t capacity, const uint32_t *rowIndex,
               const uint32_t *hllValues, uint64_t *keyVector, uint32_t *payload, bool iotaPayload, int length,
               hipStream_t stream) {
  if (length <= 0) return;
  if (static_cast<int64_t>(length) >= (1ll << 30))
    throw std::invalid_argument("Sort supports up to 2^30 - 1 rows per call");
  if (L.numDims == 0 && !hllValues) {  // a stable sort of equal keys leaves the index vector untouched
    ARES_LAUNCH("sort_const_hash_kernel", sort_const_hash_kernel, capped_grid((static_cast<int64_t>(length) + kBlock - 1) / kBlock, 256 * 8),
                kBlock, stream, keyVector, length);
    return;
  }
  // keys per lane and tile of a pass: 16 (4096-key tiles, 54 KB of LDS: two workgroups per compute unit) or 8 (2048-key
  // tiles, 27 KB: five) — ARES_SORT_KPT
  static EnvSwitch<int> keysPerLane("ARES_SORT_KPT", [](const char *e) { return e && atoi(e) == 8 ? 8 : 16; });
  const int kpt = keysPerLane.get();
  const int sortTile = kBlock * kpt;
  const int numTiles = (length + sortTile - 1) / sortTile;
  const size_t histBytes = 8 * 256 * sizeof(uint32_t);
  const size_t statusBytes = static_cast<size_t>(numTiles) * 256 * sizeof(uint32_t);
  // Row hashes are sorted by their top half and fixed up (above); HyperLogLog keys carry the register id in their low
  // 16 bits — equal top halves are the rule there — and keep the eight passes.  ARES_SORT_TOPBITS=0: eight passes always.
  static EnvSwitch<bool> topBits("ARES_SORT_TOPBITS", [](const char *e) { return !(e && e[0] == '0'); });
  const bool topOnly = !hllValues && topBits.get();
  const int fixGrid = capped_grid((static_cast<int64_t>(length) + kBlock - 1) / kBlock, 256 * 16);
  const uint32_t capPerGroup = static_cast<uint32_t>(length / 8 / fixGrid + 64);  // (expected: length^2 / 2^33 / fixGrid each)
  const size_t workCap = topOnly ? static_cast<size_t>(capPerGroup) * fixGrid : 0;
  // workspace: [hist 8 KiB][8 tickets, error, fallback][status][alt keys][alt vals][fix-up: list counters, segments]
  const size_t offTicket = histBytes, offStatus = offTicket + 64, offKeys = (offStatus + statusBytes + 255) & ~size_t(255);
  const size_t offVals = offKeys + sizeof(uint64_t) * static_cast<size_t>(length);
  const size_t offWork = (offVals + sizeof(uint32_t) * static_cast<size_t>(length) + 255) & ~size_t(255);
  // [list counters of the detect workgroups][long-segment claim table, long-segment counter][long segments][short segments]
  const size_t groupCountBytes = (sizeof(uint32_t) * static_cast<size_t>(fixGrid) + 255) & ~size_t(255);
  const size_t longTableBytes = sizeof(uint32_t) * kFixupLongSlots + 256;
  const size_t countBytes = groupCountBytes + longTableBytes;  // (zeroed together)
  const size_t longWorkBytes = sizeof(FixupSegment) * (kFixupLongSlots / 2);
  StreamBuffer ws(offWork + countBytes + longWorkBytes + sizeof(FixupSegment) * workCap + 256, stream);
  uint8_t *base = ws.as<uint8_t>();
  uint32_t *hist = reinterpret_cast<uint32_t *>(base);
  unsigned int *tickets = reinterpret_cast<unsigned int *>(base + offTicket);  // [0..7] tickets, [8] error, [9] fallback
  uint32_t *status = reinterpret_cast<uint32_t *>(base + offStatus);
  uint64_t *altKeys = reinterpret_cast<uint64_t *>(base + offKeys);
  uint32_t *altVals = reinterpret_cast<uint32_t *>(base + offVals);
  hip_check(hipMemsetAsync(base, 0, offStatus, stream), "hipMemsetAsync");

  const int grid = capped_grid((static_cast<int64_t>(length) + kBlock - 1) / kBlock, 256 * 8);
  ARES_LAUNCH("sort_hash_hist_kernel", sort_hash_hist_kernel, grid, kBlock, stream, dimValues, L, capacity, rowIndex,
              keyVector, length, hist, hllValues, iotaPayload ? payload : nullptr);
  ARES_LAUNCH("digit_start_kernel", digit_start_kernel, 8, 256, stream, hist);
  const int passGrid = capped_grid(numTiles, kpt == 8 ? 256 * 6 : 256 * 3);
  auto run_passes = [&](int first) {  // an even number of passes: the data ends where it started (keyVector / payload)
    for (int pass = first; pass < 8; pass++) {
      hip_check(hipMemsetAsync(status, 0, statusBytes, stream), "hipMemsetAsync");
      const bool even = ((pass - first) & 1) == 0;
      if (kpt == 8)
        ARES_LAUNCH("radix_pass_kernel", radix_pass_kernel<8>, passGrid, kBlock, stream, even ? keyVector : altKeys, even ? payload : altVals,
                    even ? altKeys : keyVector, even ? altVals : payload, length, 8 * pass, hist + 256 * pass, tickets + pass, tickets + 8,
                    status, numTiles);
      else
        ARES_LAUNCH("radix_pass_kernel", radix_pass_kernel<16>, passGrid, kBlock, stream, even ? keyVector : altKeys, even ? payload : altVals,
                    even ? altKeys : keyVector, even ? altVals : payload, length, 8 * pass, hist + 256 * pass, tickets + pass, tickets + 8,
                    status, numTiles);
    }
  };
  run_passes(topOnly ? 4 : 0);
  uint32_t back[2] = {0, 0};  // {error, fallback}
  if (topOnly) {
    FixupParams fp;
    fp.keys = keyVector;
    fp.vals = payload;
    fp.n = length;
    fp.groupCounts = reinterpret_cast<uint32_t *>(base + offWork);
    fp.longTable = reinterpret_cast<uint32_t *>(base + offWork + groupCountBytes);
    fp.longCount = fp.longTable + kFixupLongSlots;
    fp.longWork = reinterpret_cast<FixupSegment *>(base + offWork + countBytes);
    fp.work = reinterpret_cast<FixupSegment *>(base + offWork + countBytes + longWorkBytes);
    fp.capPerGroup = capPerGroup;
    fp.fallback = tickets + 9;
    hip_check(hipMemsetAsync(fp.groupCounts, 0, countBytes, stream), "hipMemsetAsync");
    ARES_LAUNCH("sort_fixup_detect_kernel", sort_fixup_detect_kernel, fixGrid, kBlock, stream, fp);
    ARES_LAUNCH("sort_fixup_sort_kernel", sort_fixup_sort_kernel, fixGrid, 64, stream, fp);
    ARES_LAUNCH("sort_fixup_long_kernel", sort_fixup_long_kernel, 64, kBlock, stream, fp);
  }
  read_back_u32(tickets + 8, back, 2, stream);
  if (back[0]) throw AlgorithmError("ERROR: Sort: inter-tile scan timed out");
  if (topOnly && back[1]) {
    // a long run of equal top halves that is out of order (or more segments than the list holds): all eight passes over
    // the data as it is — a stable permutation of the input (the fix-up's partial work included), so the result is the same
    hip_check(hipMemsetAsync(tickets, 0, 64, stream), "hipMemsetAsync");
    run_passes(0);
    uint32_t err = 0;
    read_back_u32(tickets + 8, &err, 1, stream);
    if (err) throw AlgorithmError("ERROR: Sort: inter-tile scan timed out");
  }
}

static void sort_impl(const DimensionVector &keys, int length, hipStream_t stream) {
  sort_rows(keys.DimValues, make_dim_layout(keys.NumDimsPerDimWidth), static_cast<size_t>(keys.VectorCapacity),
            keys.IndexVector, nullptr, keys.HashValues, keys.IndexVector, false, length, stream);
}

void sort_keys_now(const DimensionVector &keys, int length, hipStream_t stream) {
  if (length > 0) {
    const int device = current_device();
    mem_note_write(device, keys.HashValues, 8ull * static_cast<size_t>(length));
    mem_note_write(device, keys.IndexVector, 4ull * static_cast<size_t>(length));
  }
  sort_impl(keys, length, stream);
}

// ---------------------------------------------------------------------------------------------
// Reduce
// ---------------------------------------------------------------------------------------------
constexpr int kReduceKPT = 8;
constexpr int kReduceTile = kBlock * kReduceKPT;

struct ReduceParams {
  const uint64_t *hashes;
  const uint32_t *indexIn;
  const uint8_t *valuesIn;
  uint32_t *indexOut;
  uint8_t *valuesOut;
  const uint8_t *dimIn;
  uint8_t *dimOut;
  DimLayoutD L;
  size_t capacity;
  AggSpec agg;
  int n;
  int numTiles;
  unsigned int *ticket;
  uint32_t *total;
  uint32_t *error;
  uint64_t *status;
  // HyperLogLog's reduceCurrentBatch (query/hll.cu:211-231): indexIn holds entry positions; the
  // value and the dimension row of an entry are valuesIn[pos] and indexSrc[pos], the run's key goes
  // to hashOut and no dimension row is copied
  const uint32_t *indexSrc;
  uint64_t *hashOut;
};

__global__ __launch_bounds__(kBlock) void fill_identity_kernel(uint8_t *values, AggSpec a, int n) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    store_value_bits(values, a, static_cast<size_t>(i), a.identity);
}

template <bool HLL>
__global__ __launch_bounds__(kBlock) void reduce_kernel(ReduceParams p) {
  __shared__ uint32_t sTrailG[kWaves], sTrailWhole[kWaves];
  __shared__ uint64_t sTrailP[kWaves];
  __shared__ uint32_t sWave[kWaves];
  __shared__ uint32_t sTileExcl;
  __shared__ int sTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    if (threadIdx.x == 0) sTile = static_cast<int>(atomicAdd(p.ticket, 1u));
    __syncthreads();
    const int tile = sTile;
    if (tile >= p.numTiles) break;
    // blocked arrangement: lane owns kReduceKPT consecutive sorted positions
    const int64_t first = static_cast<int64_t>(tile) * kReduceTile + static_cast<int64_t>(threadIdx.x) * kReduceKPT;
    uint64_t h[kReduceKPT];
    uint32_t idx[kReduceKPT];
    uint64_t prev = 0;
    if (first > 0 && first < p.n) prev = p.hashes[first - 1];
#pragma unroll
    for (int j = 0; j < kReduceKPT; j++) {
      const int64_t i = first + j;
      h[j] = i < p.n ? p.hashes[i] : 0;
      idx[j] = i < p.n ? p.indexIn[i] : 0;
    }
    uint32_t heads = 0;  // bit j: position first+j starts a group
#pragma unroll
    for (int j = 0; j < kReduceKPT; j++) {
      const int64_t i = first + j;
      const bool head = i < p.n && (i == 0 || h[j] != (j == 0 ? prev : h[j - 1]));
      heads |= static_cast<uint32_t>(head) << j;
    }
    const uint32_t myHeads = __popc(heads);
    // block-wide exclusive scan of the head counts
    uint32_t incl = myHeads;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    uint32_t waveBase = 0, tileHeads = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) {
      if (w < wave) waveBase += sWave[w];
      tileHeads += sWave[w];
    }
    if (wave == 0) {
      if (lane == 0) st_status(p.status + tile, (tile == 0 ? kFlagInclusive : kFlagAggregate) | tileHeads);
      uint64_t excl = 0;
      if (tile > 0) {
        excl = lookback_wave(p.status, tile, lane, p.error);
        if (lane == 0) st_status(p.status + tile, kFlagInclusive | (excl + tileHeads));
      }
      if (lane == 0) {
        sTileExcl = static_cast<uint32_t>(excl);
        if (tile == p.numTiles - 1) *p.total = static_cast<uint32_t>(excl) + tileHeads;
      }
    }
    __syncthreads();
    // group number of the run that is open when this lane starts (may have begun in an earlier lane)
    uint32_t group = sTileExcl + waveBase + (incl - myHeads) - 1;
    uint64_t acc = p.agg.identity;
    bool open = false;  // acc holds a partial of `group`
#pragma unroll
    for (int j = 0; j < kReduceKPT; j++) {
      const int64_t i = first + j;
      if (i >= p.n) break;
      if ((heads >> j) & 1u) {
        if (open) aggregate_slot(p.valuesOut + static_cast<size_t>(p.agg.width) * group,
                                 reinterpret_cast<const uint8_t *>(&acc), p.agg);
        group++;
        if (HLL) {
          p.indexOut[group] = p.indexSrc[idx[j]];
          p.hashOut[group] = h[j];
        } else {
          p.indexOut[group] = idx[j];
          copy_dim_row(p.dimIn, p.capacity, p.dimOut, p.capacity, p.L, idx[j], group);
        }
        acc = load_value_bits(p.valuesIn, p.agg, idx[j]);
      } else {
        const uint64_t v = load_value_bits(p.valuesIn, p.agg, idx[j]);
        acc = open ? combine_bits(p.agg, acc, v) : v;
      }
      open = true;
    }
    // runs that continue across lanes: segmented inclusive scan over the wavefront keyed by the
    // group of each lane's trailing partial; the last lane of every group emits one atomic
    uint32_t g = open ? group : 0xffffffffu;
    uint64_t part = acc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint64_t otherPart = __shfl_up(part, off);
      const uint32_t otherG = __shfl_up(g, off);
      if (lane >= off && otherG == g && g != 0xffffffffu) part = combine_bits(p.agg, otherPart, part);
    }
    const uint32_t nextG = __shfl_down(g, 1);
    // runs that end inside the wavefront: one atomic each (different groups, little contention)
    if (open && lane != 63 && nextG != g)
      aggregate_slot(p.valuesOut + static_cast<size_t>(p.agg.width) * g, reinterpret_cast<const uint8_t *>(&part), p.agg);
    // the wavefront's trailing run may continue in the next one: when whole wavefronts belong to
    // the same group their partials are combined in LDS first, so a group that spans the tile costs
    // ONE atomic per tile instead of one per wavefront (a single hot address sustains < 100
    // atomics per microsecond)
    const uint32_t firstG = __shfl(g, 0);
    const bool wholeWave = __ballot(open && g == firstG) == ~0ull;
    if (lane == 63) {
      sTrailG[wave] = open ? g : 0xffffffffu;
      sTrailP[wave] = part;
      sTrailWhole[wave] = wholeWave ? 1u : 0u;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t carryG = 0xffffffffu;
      uint64_t carryP = 0;
      for (int w = 0; w < kWaves; w++) {
        const uint32_t gw = sTrailG[w];
        if (gw == 0xffffffffu) continue;
        if (carryG == gw && sTrailWhole[w]) {
          carryP = combine_bits(p.agg, carryP, sTrailP[w]);
        } else {
          if (carryG != 0xffffffffu)
            aggregate_slot(p.valuesOut + static_cast<size_t>(p.agg.width) * carryG, reinterpret_cast<const uint8_t *>(&carryP), p.agg);
          carryG = gw;
          carryP = sTrailP[w];
        }
      }
      if (carryG != 0xffffffffu)
        aggregate_slot(p.valuesOut + static_cast<size_t>(p.agg.width) * carryG, reinterpret_cast<const uint8_t *>(&carryP), p.agg);
    }
    __syncthreads();
  }
}

static int reduce_impl(const DimensionVector &in, uint8_t *inputValues, const DimensionVector &out,
                       uint8_t *outputValues, int valueBytes, int length, int aggFunc, hipStream_t stream) {
  ReduceParams p;
  p.agg = make_agg_spec(aggFunc, valueBytes);
  if (length <= 0) return 0;
  p.hashes = in.HashValues;
  p.indexIn = in.IndexVector;
  p.valuesIn = inputValues;
  p.indexOut = out.IndexVector;
  p.valuesOut = outputValues;
  p.dimIn = in.DimValues;
  p.dimOut = out.DimValues;
  p.L = make_dim_layout(in.NumDimsPerDimWidth);
  p.capacity = static_cast<size_t>(in.VectorCapacity);
  p.n = length;
  p.numTiles = (length + kReduceTile - 1) / kReduceTile;
  StreamBuffer ws(16 + sizeof(uint64_t) * static_cast<size_t>(p.numTiles), stream);
  hip_check(hipMemsetAsync(ws.get(), 0, 16 + sizeof(uint64_t) * static_cast<size_t>(p.numTiles), stream),
            "hipMemsetAsync");
  p.ticket = ws.as<unsigned int>();
  p.total = ws.as<uint32_t>() + 1;
  p.error = ws.as<uint32_t>() + 2;
  p.status = reinterpret_cast<uint64_t *>(ws.as<uint8_t>() + 16);
  // every group slot starts from the aggregate's identity; partials are merged with atomics
  const int fillGrid = capped_grid((static_cast<int64_t>(length) + kBlock - 1) / kBlock, 256 * 8);
  ARES_LAUNCH("fill_identity_kernel", fill_identity_kernel, fillGrid, kBlock, stream, outputValues, p.agg, length);
  p.indexSrc = nullptr;
  p.hashOut = nullptr;
  ARES_LAUNCH("reduce_kernel", reduce_kernel<false>, capped_grid(p.numTiles), kBlock, stream, p);
  uint32_t result[2] = {0, 0};  // {groups, error}
  read_back_u32(p.total, result, 2, stream);
  if (result[1]) throw AlgorithmError("ERROR: Reduce: inter-tile scan timed out");
  return static_cast<int>(result[0]);
}

int reduce_now(const DimensionVector &in, uint8_t *inputValues, const DimensionVector &out, uint8_t *outputValues, int valueBytes,
               int length, int aggFunc, hipStream_t stream) {
  if (length > 0) {
    const int device = current_device();
    mem_note_dim_rows(device, out, 0, static_cast<size_t>(length));
    mem_note_write(device, out.IndexVector, 4ull * static_cast<size_t>(length));
    mem_note_write(device, outputValues, static_cast<size_t>(valueBytes) * static_cast<size_t>(length));
  }
  return reduce_impl(in, inputValues, out, outputValues, valueBytes, length, aggFunc, stream);
}

int hll_reduce_sorted(const uint64_t *keys, const uint32_t *positions, const uint32_t *indexSrc,
                      const uint32_t *valuesSrc, uint64_t *hashOut, uint32_t *indexOut, uint32_t *valuesOut,
                      int length, hipStream_t stream) {
  if (length <= 0) return 0;
  ReduceParams p;
  p.agg = make_agg_spec(AGGR_MAX_UNSIGNED, 4);
  p.hashes = keys;
  p.indexIn = positions;
  p.valuesIn = reinterpret_cast<const uint8_t *>(valuesSrc);
  p.indexOut = indexOut;
  p.valuesOut = reinterpret_cast<uint8_t *>(valuesOut);
  p.dimIn = nullptr;
  p.dimOut = nullptr;
  p.L = DimLayoutD{};
  p.capacity = 0;
  p.n = length;
  p.numTiles = (length + kReduceTile - 1) / kReduceTile;
  p.indexSrc = indexSrc;
  p.hashOut = hashOut;
  StreamBuffer ws(16 + sizeof(uint64_t) * static_cast<size_t>(p.numTiles), stream);
  hip_check(hipMemsetAsync(ws.get(), 0, 16 + sizeof(uint64_t) * static_cast<size_t>(p.numTiles), stream),
            "hipMemsetAsync");
  p.ticket = ws.as<unsigned int>();
  p.total = ws.as<uint32_t>() + 1;
  p.error = ws.as<uint32_t>() + 2;
  p.status = reinterpret_cast<uint64_t *>(ws.as<uint8_t>() + 16);
  hip_check(hipMemsetAsync(valuesOut, 0, sizeof(uint32_t) * static_cast<size_t>(length), stream), "hipMemsetAsync");
  ARES_LAUNCH("hll_reduce_kernel", reduce_kernel<true>, capped_grid(p.numTiles), kBlock, stream, p);
  uint32_t result[2] = {0, 0};  // {runs, error}
  read_back_u32(p.total, result, 2, stream);
  if (result[1]) throw AlgorithmError("ERROR: HyperLogLog: inter-tile scan timed out");
  return static_cast<int>(result[0]);
}

// ---------------------------------------------------------------------------------------------
// Expand
// ---------------------------------------------------------------------------------------------
constexpr int kExpandKPT = 4;
constexpr int kExpandTile = kBlock * kExpandKPT;

struct ExpandParams {
  const uint32_t *baseCounts;
  const uint32_t *indexVector;
  int n;
  int numTiles;
  uint64_t *offsets;  // exclusive prefix of the run lengths (n + 1 entries)
  unsigned int *ticket;
  uint32_t *error;
  uint64_t *status;
};

__global__ __launch_bounds__(kBlock) void expand_offsets_kernel(ExpandParams p) {
  __shared__ uint64_t sWave[kWaves];
  __shared__ uint64_t sTileExcl;
  __shared__ int sTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    if (threadIdx.x == 0) sTile = static_cast<int>(atomicAdd(p.ticket, 1u));
    __syncthreads();
    const int tile = sTile;
    if (tile >= p.numTiles) break;
    const int64_t first = static_cast<int64_t>(tile) * kExpandTile + static_cast<int64_t>(threadIdx.x) * kExpandKPT;
    uint32_t c[kExpandKPT];
    uint64_t mine = 0;
#pragma unroll
    for (int j = 0; j < kExpandKPT; j++) {
      const int64_t i = first + j;
      c[j] = 0;
      if (i < p.n) {
        const uint32_t row = p.indexVector[i];
        c[j] = p.baseCounts[row + 1] - p.baseCounts[row];
      }
      mine += c[j];
    }
    uint64_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint64_t t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    uint64_t waveBase = 0, tileTotal = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) {
      if (w < wave) waveBase += sWave[w];
      tileTotal += sWave[w];
    }
    if (wave == 0) {
      if (lane == 0) st_status(p.status + tile, (tile == 0 ? kFlagInclusive : kFlagAggregate) | tileTotal);
      uint64_t excl = 0;
      if (tile > 0) {
        excl = lookback_wave(p.status, tile, lane, p.error);
        if (lane == 0) st_status(p.status + tile, kFlagInclusive | (excl + tileTotal));
      }
      if (lane == 0) {
        sTileExcl = excl;
        if (tile == p.numTiles - 1) p.offsets[p.n] = excl + tileTotal;
      }
    }
    __syncthreads();
    uint64_t run = sTileExcl + waveBase + incl - mine;
#pragma unroll
    for (int j = 0; j < kExpandKPT; j++) {
      const int64_t i = first + j;
      if (i < p.n) p.offsets[i] = run;
      run += c[j];
    }
    __syncthreads();
  }
}

// output row j copies input dim row i where offsets[i] <= j < offsets[i+1]
__global__ __launch_bounds__(kBlock) void expand_copy_kernel(const uint64_t *offsets, int n, const uint8_t *dimIn,
                                                             size_t inCap, uint8_t *dimOut, size_t outCap, DimLayoutD L,
                                                             int outLen, int occupied) {
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; j < outLen;
       j += static_cast<int64_t>(gridDim.x) * kBlock) {
    int lo = 0, hi = n;  // last i with offsets[i] <= j
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (offsets[mid] > static_cast<uint64_t>(j)) hi = mid; else lo = mid + 1;
    }
    copy_dim_row(dimIn, inCap, dimOut, outCap, L, static_cast<uint32_t>(lo - 1), static_cast<uint32_t>(occupied + j));
  }
}

static int expand_impl(const DimensionVector &in, const DimensionVector &out, uint32_t *baseCounts,
                       uint32_t *indexVector, int n, int occupied, hipStream_t stream) {
  if (n <= 0) return occupied;
  ExpandParams p;
  p.baseCounts = baseCounts;
  p.indexVector = indexVector;
  p.n = n;
  p.numTiles = (n + kExpandTile - 1) / kExpandTile;
  const size_t offBytes = sizeof(uint64_t) * (static_cast<size_t>(n) + 1);
  const size_t statusBytes = sizeof(uint64_t) * static_cast<size_t>(p.numTiles);
  StreamBuffer ws(16 + statusBytes + offBytes, stream);
  hip_check(hipMemsetAsync(ws.get(), 0, 16 + statusBytes, stream), "hipMemsetAsync");
  p.ticket = ws.as<unsigned int>();
  p.error = ws.as<uint32_t>() + 2;
  p.status = reinterpret_cast<uint64_t *>(ws.as<uint8_t>() + 16);
  p.offsets = reinterpret_cast<uint64_t *>(ws.as<uint8_t>() + 16 + statusBytes);
  ARES_LAUNCH("expand_offsets_kernel", expand_offsets_kernel, capped_grid(p.numTiles), kBlock, stream, p);
  uint64_t *pinned = pinned_words();
  hip_check(hipMemcpyAsync(pinned, p.offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, stream), "read back total");
  hip_check(hipMemcpyAsync(pinned + 1, p.error, sizeof(uint32_t), hipMemcpyDeviceToHost, stream), "read back error");
  hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
  if (static_cast<uint32_t>(pinned[1])) throw AlgorithmError("ERROR: Expand: inter-tile scan timed out");
  const uint64_t total = pinned[0];
  const int room = out.VectorCapacity - occupied;
  const int outLen = static_cast<int>(total < static_cast<uint64_t>(room > 0 ? room : 0) ? total : (room > 0 ? room : 0));
  if (outLen > 0) {
    const DimLayoutD L = make_dim_layout(in.NumDimsPerDimWidth);
    const int grid = capped_grid((static_cast<int64_t>(outLen) + kBlock - 1) / kBlock, 256 * 8);
    ARES_LAUNCH("expand_copy_kernel", expand_copy_kernel, grid, kBlock, stream, p.offsets, n, in.DimValues,
                       static_cast<size_t>(in.VectorCapacity), out.DimValues, static_cast<size_t>(out.VectorCapacity), L,
                       outLen, occupied);  // `ws` (offsets) is released in stream order, after this kernel
  }
  return outLen + occupied;
}

// ---------------------------------------------------------------------------------------------
// Queries without dimensions (COUNT(*) / SUM / MIN / MAX over the whole table: BASELINE config C2)
// ---------------------------------------------------------------------------------------------
// Every row is the empty dimension row: Sort gives all of them one constant hash and leaves the index vector as
// it was, Reduce folds the whole value vector into ONE group.  Nothing of that needs 8 + 4 bytes per row of hash
// and index traffic, an 8-pass radix sort or a segmented reduction:
//   * Sort over an index vector that is still a lazy iota defines the hash vector as a lazy fill (transform.hip)
//     and returns: no kernel;
//   * Reduce then knows the hashes are equal and the index is the identity: one reduction pass over the value rows
//     that exist — and the rows a constant measure transform only DEFINED (COUNT(*): the literal 1) are added
//     arithmetically: prev + n x c.  Integer SUM / MIN / MAX only (a float sum depends on the order of additions).
// Anything else that looks at those buffers finds them written (every flush point materialises lazy fills).
uint64_t empty_row_hash() {  // murmur3_x64_128 of zero bytes, seed 0: h1 = h2 = 0 -> fmix64(0) twice = 0
  return 0ull;
}

__global__ __launch_bounds__(kBlock) void reduce_all_kernel(const uint8_t *values, int rows, AggSpec a, uint64_t constBits,
                                                             uint32_t constRows, uint8_t *out, uint32_t *indexOut) {
  uint64_t acc = a.identity;
  bool any = false;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < rows; i += static_cast<int64_t>(gridDim.x) * kBlock) {
    acc = combine_bits(a, acc, load_value_bits(values, a, static_cast<size_t>(i)));
    any = true;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (indexOut) indexOut[0] = 0u;
    if (constRows) {  // rows that are only defined: c repeated constRows times
      uint64_t folded = constBits;
      if (a.op == OP_SUM) folded = (a.width == 8 ? constBits * static_cast<uint64_t>(constRows)
                                                 : static_cast<uint64_t>(static_cast<uint32_t>(constBits) * constRows));
      acc = combine_bits(a, acc, folded);
      any = true;
    }
  }
  // wavefront fold, then one atomic per wavefront that saw anything
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint64_t other = __shfl_xor(static_cast<unsigned long long>(acc), off);
    acc = combine_bits(a, acc, other);
  }
  if (__ballot(any) && (threadIdx.x & 63) == 0) aggregate_slot(out, reinterpret_cast<const uint8_t *>(&acc), a);
}

// Reduce of a query without dimensions over lazily defined inputs; false = not this case
static bool reduce_without_dimensions(int device, const DimensionVector &in, const uint8_t *inputValues, const DimensionVector &out,
                                      uint8_t *outputValues, int valueBytes, int length, int aggFunc, hipStream_t stream, int *groups) {
  if (length <= 0) return false;
  const DimLayoutD L = make_dim_layout(in.NumDimsPerDimWidth), LO = make_dim_layout(out.NumDimsPerDimWidth);
  if (L.numDims != 0 || LO.numDims != 0 || aggFunc == AGGR_AVG_FLOAT) return false;
  AggSpec a;
  try {
    a = make_agg_spec(aggFunc, valueBytes);
  } catch (std::exception &) {
    return false;
  }
  if (!(a.vtype == V_U32 || a.vtype == V_I32 || a.vtype == V_U64 || a.vtype == V_I64)) return false;
  uint64_t hashPattern = 0;
  int hashUnit = 0;
  if (!in.HashValues || !in.IndexVector || !virtual_iota_peek(device, in.IndexVector, length) ||
      !pending_fill_exact(device, in.HashValues, 8ull * static_cast<size_t>(length), &hashPattern, &hashUnit) || hashUnit != 8)
    return false;
  int prev = length;
  uint64_t c = 0;
  if (!pending_fill_tail(device, inputValues, a.width, length, &prev, &c)) prev = length;  // every value row exists
  if (prev > 0) {
    // rows that exist are read by the kernel below: a measure transform over a COLUMN (SUM(col), MAX(col), ...) may
    // still sit in the stream's queue — this entry point does not flush — and lazily defined rows below `prev`
    // are written now
    launch_pending_writers(device, inputValues, static_cast<size_t>(a.width) * prev);
    materialize_fills_for_read(device, inputValues, static_cast<size_t>(a.width) * prev);
  }
  retire_fills_for_write(device, outputValues, static_cast<size_t>(a.width));
  if (out.IndexVector) retire_fills_for_write(device, out.IndexVector, 4);
  mem_note_write(device, outputValues, static_cast<size_t>(a.width));
  if (out.IndexVector) mem_note_write(device, out.IndexVector, 4);
  (void)virtual_iota_peek(device, in.IndexVector, length, /*consume=*/true);  // used as what it is defined to be
  ARES_LAUNCH("fill_identity_kernel", fill_identity_kernel, 1, kBlock, stream, outputValues, a, 1);
  const int grid = capped_grid((static_cast<int64_t>(prev) + kBlock * 8 - 1) / (kBlock * 8), 256 * 4);
  ARES_LAUNCH("reduce_all_kernel", reduce_all_kernel, grid, kBlock, stream, inputValues, prev, a, c, static_cast<uint32_t>(length - prev),
              outputValues, out.IndexVector);
  *groups = 1;
  return true;
}

}  // namespace ares

using namespace ares;

extern "C" {

CGoCallResHandle Sort(DimensionVector keys, int length, void *cudaStream, int device) {
  ARES_ABI_BEGIN_NOFLUSH(device)
  // no dimensions and an index vector that is still a lazy iota: the hash vector is DEFINED (one constant), not written
  if (length > 0 && make_dim_layout(keys.NumDimsPerDimWidth).numDims == 0 && keys.HashValues && keys.IndexVector &&
      virtual_iota_peek(device, keys.IndexVector, length) &&
      defer_fill(device, reinterpret_cast<hipStream_t>(cudaStream), keys.HashValues, 8ull * static_cast<size_t>(length), empty_row_hash(), 8))
    return resHandle;
  // dimensions whose transforms are still pending on this stream: Sort is DEFINED — Reduce aggregates by the 64-bit row hash
  // without sorting rows (sort_reduce_fused.hip); anybody else who looks at the hash or index vector makes it run
  if (define_lazy_sort(device, reinterpret_cast<hipStream_t>(cudaStream), keys, length)) return resHandle;
  const bool lazyVectors = lazy_vector_sort_candidate(device, keys, length);
  flush_deferred(device);
  settle_dimension_vector(device, keys, /*rowsOnly=*/lazyVectors);
  flush_deferred_for_vector(device, keys, nullptr, 0);  // rows a HashReduce skipped, should a host sort them after all
  // the rows exist now: Sort is DEFINED all the same — Reduce orders the groups by row hash (fused_sort_reduce_vectors)
  if (lazyVectors && define_lazy_sort_vectors(device, reinterpret_cast<hipStream_t>(cudaStream), keys, length)) return resHandle;
  if (length > 0) {
    mem_note_write(device, keys.HashValues, 8ull * static_cast<size_t>(length));
    mem_note_write(device, keys.IndexVector, 4ull * static_cast<size_t>(length));
  }
  sort_impl(keys, length, reinterpret_cast<hipStream_t>(cudaStream));
  ARES_ABI_END("Sort")
}

CGoCallResHandle Reduce(DimensionVector inputKeys, uint8_t *inputValues, DimensionVector outputKeys,
                        uint8_t *outputValues, int valueBytes, int length, enum AggregateFunction aggFunc,
                        void *cudaStream, int device) {
  ARES_ABI_BEGIN_NOFLUSH(device)
  {
    int groups = 0;
    if (reduce_without_dimensions(device, inputKeys, inputValues, outputKeys, outputValues, valueBytes, length, aggFunc,
                                  reinterpret_cast<hipStream_t>(cudaStream), &groups)) {
      resHandle.res = int_result(groups);
      return resHandle;
    }
  }
  {
    int groups = 0;
    if (length > 0 && fuse_pending_into_sort_reduce(device, reinterpret_cast<hipStream_t>(cudaStream), inputKeys, inputValues, outputKeys,
                                                    outputValues, valueBytes, length, aggFunc, &groups)) {
      resHandle.res = int_result(groups);
      return resHandle;
    }
  }
  flush_deferred(device);
  settle_dimension_vector(device, inputKeys);
  settle_dimension_vector(device, outputKeys);
  retire_fills_for_write(device, outputValues, static_cast<size_t>(valueBytes) * (length > 0 ? length : 0));
  flush_deferred_for_vector(device, inputKeys, inputValues, static_cast<size_t>(valueBytes) * (length > 0 ? length : 0));
  grouped_note_write(device, outputKeys);
  grouped_note_write(device, outputValues, static_cast<size_t>(valueBytes) * (length > 0 ? length : 0));
  drop_skipped_outputs(device, outputKeys.DimValues, 1, outputValues, static_cast<size_t>(valueBytes) * (length > 0 ? length : 0));
  if (length > 0) {  // (at most `length` groups; the identity fill covers `length` values)
    mem_note_dim_rows(device, outputKeys, 0, static_cast<size_t>(length));
    mem_note_write(device, outputKeys.IndexVector, 4ull * static_cast<size_t>(length));
    mem_note_write(device, outputValues, static_cast<size_t>(valueBytes) * static_cast<size_t>(length));
  }
  resHandle.res = int_result(reduce_impl(inputKeys, inputValues, outputKeys, outputValues, valueBytes, length, aggFunc,
                                         reinterpret_cast<hipStream_t>(cudaStream)));
  ARES_ABI_END("Reduce")
}

CGoCallResHandle Expand(DimensionVector inputKeys, DimensionVector outputKeys, uint32_t *baseCounts,
                        uint32_t *indexVector, int indexVectorLen, int outputOccupiedLen, void *cudaStream, int device) {
  ARES_ABI_BEGIN(device)
  settle_dimension_vector(device, inputKeys);
  settle_dimension_vector(device, outputKeys);
  materialize_index_vector(device, indexVector);
  flush_deferred_for_vector(device, inputKeys, nullptr, 0);
  grouped_note_write(device, outputKeys);
  mem_note_vector_all(device, outputKeys);
  resHandle.res = int_result(expand_impl(inputKeys, outputKeys, baseCounts, indexVector, indexVectorLen,
                                         outputOccupiedLen, reinterpret_cast<hipStream_t>(cudaStream)));
  ARES_ABI_END("Expand")
}

}  // extern "C"
