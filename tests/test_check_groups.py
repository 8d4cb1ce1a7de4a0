"""aresdb_amd/check.py is the key-level checker of bench.py and the scale tests: its exact group-by (torch, with the
small key spaces of the low-cardinality legs spread over sub-slots) against a plain dictionary over the same rows."""
import numpy as np
import pytest
import torch

from aresdb_amd import check, workload


@pytest.mark.parametrize("dims", [("ts", "d1", "d2", "d3"), ("d2", "d3"), ("d1",), ("ts", "d1")])
def test_exact_groups_against_a_dictionary(dims):
    batches = workload.c3_shard(30000, 12000, 5, torch.device("cpu"), null_fraction=0.05)
    code, sums, first, rows = check.exact_groups(batches, dims=dims, d1_below=70)
    want = {}
    offset = 0
    for b in batches:
        cols = {n: (b[n].values().numpy(), None if b[n].valid() is None else b[n].valid().numpy()) for n in ("ts", "d1", "d2", "d3", "m")}
        n = b["m"].length
        for i in range(n):
            d1, d1ok = cols["d1"][0][i], cols["d1"][1] is None or cols["d1"][1][i]
            if not (d1ok and d1 < 70):
                continue
            key = []
            for name in dims:
                v, ok = cols[name]
                valid = ok is None or bool(ok[i])
                x = int(v[i]) // 3600 * 3600 if name == "ts" else int(v[i])  # the stored dimension value: the hourly bucket
                key.append((x, True) if valid else (0, False))
            mok = cols["m"][1] is None or cols["m"][1][i]
            e = want.setdefault(tuple(key), [0.0, offset + i, 0])
            e[0] += float(cols["m"][0][i]) if mok else 0.0
            e[2] += 1
        offset += n
    values, valids = check.decode_codes(code, dims)
    got = {tuple((int(values[d][g]), bool(valids[d][g])) if valids[d][g] else (0, False) for d in range(len(dims))): (sums[g], first[g], rows[g])
           for g in range(len(code))}
    assert set(got) == set(want) and len(want) > (50 if len(dims) > 1 else 20)
    for k, (s, f, r) in got.items():
        assert abs(s - want[k][0]) <= 1e-9 * max(1.0, abs(want[k][0])) and f == want[k][1] and r == want[k][2], k


def test_row_hash_of_the_checker_matches_the_reference_known_answers():
    """The 64-bit row hash the ordered comparison goes by (check.murmur3_128_lo64_rows) against the four hashes of
    SortAndReduceTest.CheckHash (query/algorithm_unittest.cu:1160-1217: one 1-byte dimension, rows {value, validity = 1};
    SURVEY.md 8c) and against the oracle's Sort over random rows of 5, 16, 20, 25 and 33 bytes (tail and multi-block paths)."""
    rows = np.array([[2, 1], [0, 1], [3, 1], [1, 1]], np.uint8)
    assert [int(h) for h in check.murmur3_128_lo64_rows(rows)] == [0x60e187b4814392c4, 0x7cb3f5c58dab264c, 0xb73e42bb654cee53, 0xca410abc0a9d4c6b]
    import harness as H
    from aresdb_amd import abi
    be = H.oracle_backend()
    rng = np.random.default_rng(4)
    for ndw in ((0, 0, 1, 0, 0), (0, 0, 3, 1, 0), (0, 0, 4, 0, 0), (0, 1, 2, 2, 1), (1, 1, 1, 0, 0)):
        n = 300
        dv = H.DimVector(be, n, ndw)
        blob = rng.integers(0, 256, dv.nbytes).astype(np.uint8)
        dv.values.write(blob)
        be.call("InitIndexVector", dv.index.ptr, 0, n, None, 0)
        be.call("Sort", dv.struct(), n, None, 0)
        got = np.sort(dv.hash.read(np.uint64, n))
        packed = np.concatenate([blob[vo:vo + w * n].reshape(n, w) for vo, _, w in dv.dim_offsets()] +
                                [blob[no:no + n].reshape(n, 1) for _, no, _ in dv.dim_offsets()], axis=1)
        assert np.array_equal(np.sort(check.murmur3_128_lo64_rows(packed)), got), ndw
        dv.free()


def test_exact_groups_over_slices_equals_whole_batches():
    """bench.py compares the reference's HOST result of a SAMPLE spread over the whole shard (chunks taken round-robin from
    every batch) with exact_groups(slices=...): slices that tile the batches must give the whole shard's groups."""
    batches = workload.c3_shard(30000, 8000, seed=5, device=torch.device("cpu"), null_fraction=0.02)
    whole = check.exact_groups(batches)
    slices = []
    for bi, b in enumerate(batches):
        n = next(iter(b.values())).length
        slices += [(bi, 0, 3000), (bi, 3000, n - 3000)]
    tiled = check.exact_groups(batches, slices=slices)
    assert np.array_equal(whole[0], tiled[0]) and np.array_equal(whole[1], tiled[1]) and np.array_equal(whole[3], tiled[3])
    part = check.exact_groups(batches, slices=[(1, 100, 500), (3, 0, 200)])
    assert int(part[3].sum()) == int(check.exact_groups(batches[1:2], limit_first_batch=600)[3].sum()) - \
        int(check.exact_groups(batches[1:2], limit_first_batch=100)[3].sum()) + int(check.exact_groups(batches[3:4], limit_first_batch=200)[3].sum())
