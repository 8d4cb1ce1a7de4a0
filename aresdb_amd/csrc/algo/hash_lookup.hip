// HashLookup for MI355X (gfx950): foreign-key column -> RecordID through the cuckoo index.
//
// Reference: query/hash_lookup.cu:70-157 and HashLookupFunctor, query/functor.hpp:1173-1266.
// Index layout (memstore/cuckoo_index.go:42-48): numBuckets buckets + 1 stash bucket of
// bucketBytes = 8 * (8 + 1 + keyBytes); bucket = [RecordID x8][signature u8 x8][key x8].
//
// One lane probes one row.  Unlike the reference's byte-wise memequal the slot test is done on
// the 8 signature bytes at once (one 8-byte load, the candidate slots as a bit mask) and a key is
// only fetched for slots whose signature matches, so a probe normally touches 64 B of
// signatures/keys plus one 8-byte RecordID instead of walking 104 B byte by byte.
#include <hip/hip_runtime.h>

#include "binding.hpp"
#include "common.hpp"
#include "device_model.hpp"
#include "fast_eval.hpp"

namespace ares {

constexpr int kBlock = 256;

struct LookupParams {
  OperandD a;
  const uint32_t *idx;
  const uint32_t *baseCounts;
  uint32_t startCount;
  int needRow;
  const uint8_t *buckets;
  uint32_t seeds[4];
  int keyBytes;
  int numHashes;
  uint32_t numBuckets;
  uint32_t bucketBytes;
};

struct __attribute__((packed, aligned(1))) KeyWord { uint32_t v; };

__device__ __forceinline__ bool key_equal(const uint8_t *slotKey, const uint32_t (&key)[4], int keyBytes) {
  // keys are stored unaligned (offset 72 + j*keyBytes): word-wise with byte-aligned loads (gfx950 global loads need no
  // alignment), the odd tail byte-wise.  Constant indices into `key`: a run-time index would put it in scratch memory.
  bool same = true;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const int have = keyBytes - 4 * w;
    if (have >= 4) {
      same = same && reinterpret_cast<const KeyWord *>(slotKey)[w].v == key[w];
    } else if (have > 0) {
      uint32_t tail = 0;
      for (int b = 0; b < have; b++) tail |= static_cast<uint32_t>(slotKey[4 * w + b]) << (8 * b);
      same = same && tail == (key[w] & ((1u << (8 * have)) - 1u));
    }
  }
  return same;
}

__global__ __launch_bounds__(kBlock) void hash_lookup_kernel(LookupParams p, RecordID *out, int n) {
  const FastDivisor bucketDiv = make_fast_divisor(p.numBuckets);  // hv % numBuckets by multiply-high
  for (int64_t i64 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i64 < n;
       i64 += static_cast<int64_t>(gridDim.x) * kBlock) {
    const uint32_t i = static_cast<uint32_t>(i64);
    const uint32_t row = p.needRow ? p.idx[i] : i;
    uint32_t key[4] = {0, 0, 0, 0};
    uint32_t ok;
    if (p.a.kind == K_I64 || p.a.kind == K_UUID) {
      // wide keys: read the raw value bytes
      const int w = p.a.kind == K_UUID ? 16 : 8;
      if (p.a.type == OP_CONST) {
        key[0] = static_cast<uint32_t>(p.a.c64[0]); key[1] = static_cast<uint32_t>(p.a.c64[0] >> 32);
        key[2] = static_cast<uint32_t>(p.a.c64[1]); key[3] = static_cast<uint32_t>(p.a.c64[1] >> 32);
        ok = p.a.cok;
      } else {
        const uint32_t pos = locate(p.a, row, p.baseCounts, p.startCount);
        const uint32_t *v = reinterpret_cast<const uint32_t *>(p.a.base + p.a.valuesOff + static_cast<size_t>(w) * pos);
#pragma unroll  // (constant indices keep the key words in registers)
        for (int k = 0; k < 4; k++)
          if (k < (w >> 2)) key[k] = v[k];
        ok = p.a.mode >= 2 ? get_bit(p.a.base + p.a.nullsOff, pos + p.a.bitOff) : 1u;
      }
    } else {
      const DVal v = load32(p.a, i, row, p.baseCounts, p.startCount);
      key[0] = v.bits;
      ok = v.ok;
    }
    // the key is the first keyBytes bytes of the (widened) value: clear everything beyond
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const int have = p.keyBytes - 4 * w;  // bytes of this word that belong to the key
      if (have <= 0) key[w] = 0;
      else if (have < 4) key[w] &= (1u << (8 * have)) - 1u;
    }
    RecordID rid = {0, 0};
    if (ok) {
      bool found = false;
#pragma unroll  // (constant indices into p.seeds)
      for (int h = 0; h < 4; h++) {
        if (h >= p.numHashes || found) break;
        const uint32_t hv = murmur3_32_words<4>(key, p.keyBytes, p.seeds[h]);
        uint32_t bq, bidx;
        fast_divmod(bucketDiv, hv, bq, bidx);
        if (p.numBuckets == 1) bidx = 0;
        const uint8_t *bucket = p.buckets + static_cast<size_t>(bidx) * p.bucketBytes;
        uint32_t sig = hv >> 24;
        if (sig < 1) sig = 1;
        // 8 signature bytes at offset 64 (bucket start is 8-byte aligned: bucketBytes % 8 == 0)
        const uint64_t sigs = *reinterpret_cast<const uint64_t *>(bucket + 64);
        // candidate slots first (pure ALU on the 8 signature bytes), then only those are visited, in
        // slot order: lanes of a wavefront no longer walk all eight slots in lockstep (4.3 -> 2.4 ms
        // per 64 Mi rows; hashing all four bucket choices up front instead was slower: 2.75 ms)
        uint32_t cand = 0;
#pragma unroll
        for (int j = 0; j < HASH_BUCKET_SIZE; j++) cand |= (((sigs >> (8 * j)) & 0xff) == sig ? 1u : 0u) << j;
        while (cand) {
          const int j = __builtin_ctz(cand);
          cand &= cand - 1;
          if (key_equal(bucket + 72 + j * p.keyBytes, key, p.keyBytes)) {
            rid = reinterpret_cast<const RecordID *>(bucket)[j];
            found = true;
            break;
          }
        }
      }
      if (!found) {
        const uint8_t *stash = p.buckets + static_cast<size_t>(p.numBuckets) * p.bucketBytes;
        for (int j = 0; j < HASH_STASH_SIZE; j++) {
          if (stash[64 + j] != 0 && key_equal(stash + 72 + j * p.keyBytes, key, p.keyBytes)) {
            rid = reinterpret_cast<const RecordID *>(stash)[j];
            break;
          }
        }
      }
    }
    out[i] = rid;
  }
}

}  // namespace ares

using namespace ares;

extern "C" CGoCallResHandle HashLookup(InputVector input, RecordID *output, uint32_t *indexVector,
                                       int indexVectorLength, uint32_t *baseCounts, uint32_t startCount,
                                       CuckooHashIndex hashIndex, void *cudaStream, int device) {
  ARES_ABI_BEGIN(device)
  materialize_index_vector(device, indexVector);
  hipStream_t stream = reinterpret_cast<hipStream_t>(cudaStream);
  if (indexVectorLength > 0) {
    LookupParams p;
    memset(&p, 0, sizeof(p));
    CallTemps temps;
    bind_operand(input, true, stream, p.a, temps);
    if (p.a.kind == K_GEO || ((p.a.type == OP_SCRATCH || p.a.type == OP_FOREIGN) && is_wide(p.a.kind)))
      throw std::invalid_argument("Unsupported data type for HashLookup");
    if (hashIndex.keyBytes < 1 || hashIndex.keyBytes > 16 || hashIndex.numBuckets < 1 ||
        hashIndex.numHashes < 0 || hashIndex.numHashes > 4)
      throw std::invalid_argument("Unsupported cuckoo hash index geometry");
    p.idx = indexVector;
    p.baseCounts = baseCounts;
    p.startCount = startCount;
    p.needRow = indexVector != nullptr && p.a.type == OP_COLUMN;
    p.buckets = hashIndex.buckets;
    for (int i = 0; i < 4; i++) p.seeds[i] = hashIndex.seeds[i];
    p.keyBytes = hashIndex.keyBytes;
    p.numHashes = hashIndex.numHashes;
    p.numBuckets = static_cast<uint32_t>(hashIndex.numBuckets);
    p.bucketBytes = HASH_BUCKET_SIZE * (8 + 1 + hashIndex.keyBytes);
    const int grid = capped_grid((static_cast<int64_t>(indexVectorLength) + kBlock - 1) / kBlock, 256 * 16);
    mem_note_write(device, output, sizeof(RecordID) * static_cast<size_t>(indexVectorLength));
    ARES_LAUNCH("hash_lookup_kernel", hash_lookup_kernel, grid, kBlock, stream, p, output, indexVectorLength);
  }
  resHandle.res = int_result(indexVectorLength);
  ARES_ABI_END("HashLookup")
}
