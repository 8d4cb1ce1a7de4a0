"""HyperLogLog end to end at the ABI level: GetHLLValue measure transform -> HyperLogLog over
several batches -> decoded registers estimate the distinct counts (query/hll.cu, query/common/hll.go)."""
import cases


def test_estimate_of_known_distinct_counts(be):
    cases.hll_estimate_check(be, 60000, [300, 5000, 40000])
