// CPU check of sort_topbits_fixup.hpp: g++ -O2 -std=c++17 -o t sort_topbits_fixup_test.cpp && ./t
// "sorted by the top half only, stably" + fix-up must equal a stable sort by the whole key — or raise the fallback.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#include "sort_topbits_fixup.hpp"

using namespace ares_proto;

struct Case {
  int n;
  uint32_t tops, lows;  // how many distinct top / low halves the keys draw from
  int hotRun;           // > 0: that many copies of one key up front in the input (a hot group)
  int maxRun;
  bool expectFallback;
};

static int run_case(const Case &c, uint32_t seed, int order) {
  std::mt19937_64 rng(seed);
  std::vector<uint64_t> keys(c.n);
  std::vector<uint32_t> vals(c.n);
  for (int i = 0; i < c.n; i++) {
    keys[i] = (static_cast<uint64_t>(rng() % c.tops) << 32) | (rng() % c.lows) * 2654435761u % 0xFFFFFFFFu;
    vals[i] = static_cast<uint32_t>(i);  // payload = original position: stability is visible
  }
  for (int i = 0; i < c.hotRun && i < c.n; i++) keys[(static_cast<size_t>(i) * 7919u) % c.n] = (uint64_t{3} << 32) | 0x80000000u;
  // what the four high passes leave: a stable sort by the top half
  std::vector<int> perm(c.n);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return top_half(keys[a]) < top_half(keys[b]); });
  std::vector<uint64_t> k(c.n);
  std::vector<uint32_t> v(c.n);
  for (int i = 0; i < c.n; i++) { k[i] = keys[perm[i]]; v[i] = vals[perm[i]]; }
  // the answer: a stable sort by the whole key
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return keys[a] < keys[b]; });
  std::vector<Segment> work(c.n / 2 + 1);
  uint32_t count = 0, fallback = 0;
  FixupParams p{k.data(), v.data(), c.n, work.data(), &count, static_cast<uint32_t>(work.size()), &fallback, c.maxRun};
  auto append = [](uint32_t *ctr) { return (*ctr)++; };
  // "threads" in three different orders: detection reads only, so the order must not matter
  if (order == 0) for (int i = 0; i < c.n; i++) fixup_detect(p, i, append);
  if (order == 1) for (int i = c.n - 1; i >= 0; i--) fixup_detect(p, i, append);
  if (order == 2) { std::vector<int> o(c.n); std::iota(o.begin(), o.end(), 0); std::shuffle(o.begin(), o.end(), rng); for (int i : o) fixup_detect(p, i, append); }
  if (fallback) return c.expectFallback ? 0 : (fprintf(stderr, "unexpected fallback (n %d seed %u)\n", c.n, seed), 1);
  if (c.expectFallback) return fprintf(stderr, "fallback expected (n %d seed %u)\n", c.n, seed), 1;
  // segments must be disjoint (they are sorted concurrently)
  std::vector<Segment> segs(work.begin(), work.begin() + count);
  std::sort(segs.begin(), segs.end(), [](const Segment &a, const Segment &b) { return a.start < b.start; });
  for (size_t i = 1; i < segs.size(); i++)
    if (segs[i].start < segs[i - 1].end) return fprintf(stderr, "overlapping segments\n"), 1;
  for (uint32_t w = count; w-- > 0;) fixup_sort(p, w);
  for (int i = 0; i < c.n; i++)
    if (k[i] != keys[perm[i]] || v[i] != vals[perm[i]]) return fprintf(stderr, "mismatch at %d (n %d seed %u order %d)\n", i, c.n, seed, order), 1;
  return 0;
}

int main() {
  const Case cases[] = {
      {100000, 1u << 20, 1u << 30, 0, 64, false},   // sparse collisions: the production regime
      {100000, 5000, 1u << 30, 0, 64, false},       // ~20 keys per segment, all mixed
      {50000, 3000, 7, 0, 64, false},               // many duplicates of few low halves inside every segment
      {20000, 1u << 20, 1u << 30, 6000, 64, false}, // a hot group: 6000 equal keys in one segment, no descent in it unless a neighbour collides
      {20000, 40, 1u << 30, 0, 64, true},           // segments of ~500 mixed keys: beyond maxRun -> fallback
      {20000, 40, 1u << 30, 0, 100000, false},      // the same with an unbounded walk: sorted
      {1, 4, 4, 0, 64, false}, {2, 1, 1u << 30, 0, 64, false}, {0, 4, 4, 0, 64, false},
  };
  int bad = 0, ran = 0;
  for (const Case &c : cases)
    for (uint32_t seed = 1; seed <= 5; seed++)
      for (int order = 0; order < 3; order++) { bad += run_case(c, seed, order); ran++; }
  printf("%d cases, %d failed\n", ran, bad);
  return bad ? 1 : 0;
}
