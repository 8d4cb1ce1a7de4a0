"""Differential pinning of the C oracle against the reference's own HOST build.

oracle/_ref/libalgorithm.so is compiled from the UNMODIFIED reference sources by
oracle/Makefile.ref; on hundreds of seeded cases the plain-C restatement must produce
bit-identical buffers.  Skipped when oracle/_ref has not been built.
"""
import numpy as np
import pytest

import cases
import harness as H

pytestmark = pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")


@pytest.mark.parametrize("arity", [1, 2])
@pytest.mark.parametrize("as_filter", [False, True])
@pytest.mark.parametrize("block", range(8))
def test_transform_and_filter(arity, as_filter, block):
    o, r = H.oracle_backend(), H.ref_backend()
    for seed in range(block * 40, block * 40 + 40):
        c = cases.TransformCase(seed * 4 + arity * 2 + int(as_filter), arity, as_filter)
        cases.assert_same(c.run(o), c.run(r), repr(c))


@pytest.mark.parametrize("seed", range(40))
def test_hash_lookup(seed):
    c = cases.HashLookupCase(seed)
    cases.assert_same(c.run(H.oracle_backend()), c.run(H.ref_backend()), f"HashLookupCase({seed})")


@pytest.mark.parametrize("seed", range(60))
def test_sort_reduce(seed):
    c = cases.GroupByCase(seed)
    cases.assert_same(c.run_sort_reduce(H.oracle_backend()), c.run_sort_reduce(H.ref_backend()),
                      f"GroupByCase({seed}) sort/reduce")


@pytest.mark.parametrize("seed", range(60))
def test_hash_reduce(seed):
    c = cases.GroupByCase(1000 + seed)
    cases.assert_same(c.run_hash_reduce(H.oracle_backend()), c.run_hash_reduce(H.ref_backend()),
                      f"GroupByCase({1000 + seed}) hash reduce")


@pytest.mark.parametrize("seed", range(40))
def test_hyperloglog(seed):
    c = cases.HllCase(seed)
    cases.assert_same(c.run(H.oracle_backend()), c.run(H.ref_backend()), repr(c))


def test_hyperloglog_dense_groups():
    c = cases.HllCase(500, batches=3, batch_rows=30000, groups=40, registers=1 << 14)
    r = c.run(H.oracle_backend())
    assert (r["reg_counts"] >= 4096).any() and (r["reg_counts"] < 4096).any()
    cases.assert_same(r, c.run(H.ref_backend()), repr(c))


@pytest.mark.parametrize("seed", range(60))
def test_geo_intersects(seed):
    c = cases.GeoCase(seed)
    r = c.run(H.oracle_backend())
    cases.assert_same(r, c.run(H.ref_backend()), repr(c))


def test_murmur_known_answers():
    """Row hashes recorded from the reference build: row = {value, validity=1}."""
    import ctypes as C
    lib = C.CDLL(H.ORACLE_SO)
    lib.oracle_murmur3_128.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 2)()
    exp = {2: 0x60e187b4814392c4, 0: 0x7cb3f5c58dab264c, 3: 0xb73e42bb654cee53, 1: 0xca410abc0a9d4c6b}
    for v, h in exp.items():
        lib.oracle_murmur3_128(bytes([v, 1]), 2, 0, out)
        assert out[0] == h


@pytest.mark.parametrize("as_filter", [False, True], ids=["transform", "filter"])
@pytest.mark.parametrize("seed", range(42))
def test_fast_path_shapes_oracle_vs_reference(seed, as_filter):
    """The directed hot-shape cases of tests/test_hip_parity.py::test_fast_path_shapes, pinned on
    the reference's own build first (small sizes only: the HOST build is single-threaded)."""
    rows = [1, 2, 3, 5, 63, 1000, 4095, 4096, 4099, 8191, 8192, 8197, 2000, 3001][seed % 14]
    style = ["identity", "subset", "perm"][(seed // 2) % 3]
    c = cases.fast_path_case(seed, as_filter, rows, style)
    cases.assert_same(c.run(H.oracle_backend()), c.run(H.ref_backend()), repr(c))
