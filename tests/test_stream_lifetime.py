"""Nothing of a stream outlives it inside the two libraries (round-4 finding, profiles/r4_race_hunt.md): an event that
was recorded on a stream and is queried, waited for or re-recorded after the stream's destruction makes the ROCm 7.x HIP
runtime reach into the destroyed stream object — std::bad_variant_access out of hipEventQuery, segmentation faults,
hangs, and single words of unrelated heap memory changed by one (which is what the "off by one filter count" of the
round-3 sequence fuzzer was: a corrupted INPUT element).  libmem.so's fence events, libalgorithm.so's error-word watches
and profiler events are therefore retired inside DestroyCudaStream, while the stream still exists."""
import ctypes as C

import numpy as np
import pytest

import harness as H
from aresdb_amd import abi

pytestmark = pytest.mark.gpu


def _events(be, stream):
    be._mem.AresMemStreamEvents.argtypes, be._mem.AresMemStreamEvents.restype = [C.c_int, C.c_void_p], C.c_size_t
    be._algo.AresStreamEvents.argtypes, be._algo.AresStreamEvents.restype = [C.c_int, C.c_void_p], C.c_size_t
    return be._mem.AresMemStreamEvents(0, stream), be._algo.AresStreamEvents(0, stream)


def _lazy_filter_then_copy(be, stream, n=20000):
    """a fast filter (lazy compaction) whose index vector is then copied: the compaction runs late and leaves an error-word
    watch (event + pinned word) behind on `stream`"""
    vals = np.arange(n, dtype=np.uint32) % 97
    col = H.Column(be, abi.Uint32, vals)
    idx, pred = H.Buf(be, nbytes=4 * n), H.Buf(be, nbytes=n)
    be.call("InitIndexVector", idx.ptr, 0, n, stream, 0)
    kept = be.call("BinaryFilter", col.input(), H.const_int(50), idx.ptr, pred.ptr, n, None, 0, None, 0, abi.LessThan, stream, 0)
    got = H.download(be, idx.ptr, 4 * kept, stream).view(np.uint32)
    assert np.array_equal(got, np.flatnonzero(vals < 50).astype(np.uint32))
    for x in (col, idx, pred):
        x.free()  # fence events on every live stream


def test_no_event_survives_its_stream():
    be = H.hip_backend()
    keep = be.call("CreateCudaStream", 0)  # another live stream: its events must stay
    s = be.call("CreateCudaStream", 0)
    _lazy_filter_then_copy(be, s)
    mem_before, algo_before = _events(be, s)
    assert mem_before > 0, "the frees above fence against every live stream"
    assert algo_before > 0, "the late compaction leaves an error-word watch on its stream"
    kept_before = _events(be, keep)[0]
    assert kept_before > 0
    be.call("DestroyCudaStream", s, 0)
    assert _events(be, s) == (0, 0)
    assert _events(be, keep)[0] == kept_before  # only the destroyed stream's events went
    be.call("DestroyCudaStream", keep, 0)
    assert _events(be, keep) == (0, 0)


def test_profiler_events_are_resolved_when_their_stream_goes():
    be = H.hip_backend()
    s = be.call("CreateCudaStream", 0)
    be.profiler_enable(True)
    _lazy_filter_then_copy(be, s)
    assert _events(be, s)[1] > 0
    be.call("DestroyCudaStream", s, 0)
    assert _events(be, s) == (0, 0)
    kernels = be.profiler_report()  # durations measured before the events went
    be.profiler_enable(False)
    assert any(k.startswith("filter_pred_kernel") for k in kernels), kernels
    assert all(ms >= 0 for _, ms in kernels.values())


def test_stream_churn_with_allocation_churn():
    """the pattern that used to abort about once per 3 000 short queries: streams created and destroyed per query while
    blocks freed under them wait in the cache behind events of those streams"""
    be = H.hip_backend()
    rng = np.random.default_rng(5)
    for _ in range(300):
        streams = [be.call("CreateCudaStream", 0) for _ in range(2)]
        bufs = [H.Buf(be, rng.integers(0, 255, int(rng.integers(100, 60000))).astype(np.uint8)) for _ in range(6)]
        for b in bufs[:3]:
            b.free()
        _lazy_filter_then_copy(be, streams[0], n=int(rng.integers(100, 30000)))
        for b in bufs[3:]:
            b.free()
        for s in streams:
            be.call("DestroyCudaStream", s, 0)
            assert _events(be, s) == (0, 0)
