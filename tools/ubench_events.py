"""What one DeviceFree's fence costs the host, call by call (HIP runtime through ctypes, ~1 us of ctypes overhead per call):
hipEventRecord on the null stream, on an idle non-blocking stream, hipEventQuery of a completed event, hipStreamWaitEvent,
and a record on the null stream while other streams exist (the runtime looks at every stream for the null stream's
implicit ordering).  usage: python tools/ubench_events.py [reps]"""
import ctypes as C
import sys
import time

hip = C.CDLL("libamdhip64.so")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000


def chk(rc):
    assert rc == 0, rc


chk(hip.hipSetDevice(0))
chk(hip.hipFree(None))
ev = [C.c_void_p() for _ in range(4)]
for e in ev:
    chk(hip.hipEventCreateWithFlags(C.byref(e), 2))  # hipEventDisableTiming


def timed(label, fn):
    for _ in range(200):
        fn()
    chk(hip.hipDeviceSynchronize())
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t0) / reps * 1e6
    chk(hip.hipDeviceSynchronize())
    print(f"{label}: {dt:.2f} us")


timed("hipEventRecord(null stream), no other stream", lambda: hip.hipEventRecord(ev[0], None))
streams = [C.c_void_p() for _ in range(2)]
for s in streams:
    chk(hip.hipStreamCreateWithFlags(C.byref(s), 1))  # hipStreamNonBlocking
timed("hipEventRecord(null stream), two non-blocking streams exist", lambda: hip.hipEventRecord(ev[0], None))
timed("hipEventRecord(idle non-blocking stream)", lambda: hip.hipEventRecord(ev[1], streams[0]))
chk(hip.hipDeviceSynchronize())
timed("hipEventQuery(completed event)", lambda: hip.hipEventQuery(ev[1]))
timed("hipStreamWaitEvent(stream 1, completed event of stream 0)", lambda: hip.hipStreamWaitEvent(streams[1], ev[1], 0))


def fence():
    hip.hipEventRecord(ev[0], None)
    hip.hipEventRecord(ev[1], streams[0])
    hip.hipEventRecord(ev[2], streams[1])


timed("one free's fence: three records", fence)


def fence_and_query():
    fence()
    hip.hipEventQuery(ev[0]); hip.hipEventQuery(ev[1]); hip.hipEventQuery(ev[2])


timed("three records + three queries (a free and the allocation that reuses the block)", fence_and_query)
timed("hipStreamSynchronize(idle stream)", lambda: hip.hipStreamSynchronize(streams[0]))
