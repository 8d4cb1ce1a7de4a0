// PROTOTYPE for the next round (not part of libalgorithm.so yet).
//
// Sort orders (64-bit row hash, 32-bit payload) pairs by all 64 bits with eight one-sweep radix passes because the
// order of Reduce's output — ascending hash — is observable.  With random hashes the top 32 bits already decide the
// order of nearly every pair: of 50 M distinct hashes about 50e6^2 / 2 / 2^32 = 290 k pairs share their top half.  So:
// four passes over bits 32..63 (stable), then this fix-up — two small kernels:
//   detect   one thread per element i: a DESCENT (same top half as i - 1, smaller low half) means its segment (the run
//            of equal top halves) is out of order.  The thread of a segment's FIRST descent walks to the segment's ends
//            (bounded by maxRun) and appends [start, end) to a work list.  Segments without a descent — every run of
//            equal keys, however long: a hot group — cost one comparison per element and are never walked.
//   sort     one thread per listed segment: stable insertion sort by the low half, in place (segments are disjoint).
// A segment longer than maxRun that contains a descent, or a full work list, raises `fallback`: the caller runs the
// eight full passes on the data as it is (still a stable permutation of the input, so the result is the same).
//
// The functions are host+device so that the logic is tested on the CPU (sort_topbits_fixup_test.cpp, run by
// tests/test_prototypes.py); the kernels that will call them are one line each.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define ARES_HD __host__ __device__
#else
#define ARES_HD
#endif

namespace ares_proto {

struct Segment {
  uint32_t start, end;
};

struct FixupParams {
  uint64_t *keys;
  uint32_t *vals;
  int n;
  Segment *work;
  uint32_t *workCount;  // zeroed; may exceed workCap (then fallback is set)
  uint32_t workCap;
  uint32_t *fallback;  // zeroed
  int maxRun;
};

ARES_HD inline uint32_t top_half(uint64_t k) { return static_cast<uint32_t>(k >> 32); }
ARES_HD inline uint32_t low_half(uint64_t k) { return static_cast<uint32_t>(k); }

// append(counter) -> the slot reserved: atomicAdd on the device, a plain increment in the CPU test
template <typename Append>
ARES_HD inline void fixup_detect(const FixupParams &p, int i, Append append) {
  if (i <= 0 || i >= p.n) return;
  const uint64_t k = p.keys[i], before = p.keys[i - 1];
  if (top_half(k) != top_half(before) || low_half(k) >= low_half(before)) return;
  const uint32_t top = top_half(k);
  // the first descent of the segment lists it: walk back to the segment's start, giving up at an earlier descent
  int s = i - 1;
  while (s > 0 && top_half(p.keys[s - 1]) == top) {
    if (low_half(p.keys[s]) < low_half(p.keys[s - 1])) return;
    s--;
    if (i - s > p.maxRun) {
      *p.fallback = 1u;
      return;
    }
  }
  int e = i + 1;
  while (e < p.n && top_half(p.keys[e]) == top) {
    e++;
    if (e - s > p.maxRun) {
      *p.fallback = 1u;
      return;
    }
  }
  const uint32_t slot = append(p.workCount);
  if (slot >= p.workCap) {
    *p.fallback = 1u;
    return;
  }
  p.work[slot] = Segment{static_cast<uint32_t>(s), static_cast<uint32_t>(e)};
}

ARES_HD inline void fixup_sort(const FixupParams &p, uint32_t w) {
  const Segment seg = p.work[w];
  for (uint32_t a = seg.start + 1; a < seg.end; a++) {
    const uint64_t k = p.keys[a];
    const uint32_t v = p.vals[a];
    uint32_t b = a;
    while (b > seg.start && low_half(p.keys[b - 1]) > low_half(k)) {  // strict: equal keys keep their order
      p.keys[b] = p.keys[b - 1];
      p.vals[b] = p.vals[b - 1];
      b--;
    }
    p.keys[b] = k;
    p.vals[b] = v;
  }
}

}  // namespace ares_proto
