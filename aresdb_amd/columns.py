"""Host -> device transfer of one column batch, as the reference host does it
(query/aql_processor.go:1345-1431 transferLiveBatch + :1415-1429 layout, and
query/time_series_aggregate.go:166-206 makeVectorPartySlice): one device allocation per column laid
out [counts u32 x(len+1)][validity bitmap][values], every section starting on a 64-byte boundary."""
import ctypes as C

import numpy as np

from . import abi

_NP_OF = {abi.Int8: np.int8, abi.Uint8: np.uint8, abi.Int16: np.int16, abi.Uint16: np.uint16,
          abi.Int32: np.int32, abi.Uint32: np.uint32, abi.Float32: np.float32, abi.Int64: np.int64}


def _align64(n):
    return (n + 63) // 64 * 64


class DeviceColumn:
    def __init__(self, be: abi.Backend, data_type, values, valid=None, counts=None, device=0, stream=None):
        self.be, self.device = be, device
        values = np.asarray(values)
        if data_type == abi.Bool:
            vbytes = np.packbits(values.astype(bool), bitorder="little")
        elif data_type == abi.GeoPoint:  # (n, 2) float32 {lat, long}
            vbytes = np.ascontiguousarray(values.astype(np.float32)).reshape(-1).view(np.uint8)
        else:
            vbytes = np.ascontiguousarray(values.astype(_NP_OF[data_type])).view(np.uint8)
        cbytes = np.zeros(0, np.uint8) if counts is None else np.asarray(counts, np.uint32).view(np.uint8)
        nbytes_ = np.zeros(0, np.uint8) if valid is None else np.packbits(np.asarray(valid, bool), bitorder="little")
        no = _align64(len(cbytes))
        vo = no + _align64(len(nbytes_))
        total = vo + _align64(len(vbytes))
        blob = np.zeros(total, np.uint8)
        blob[:len(cbytes)] = cbytes
        blob[no:no + len(nbytes_)] = nbytes_
        blob[vo:vo + len(vbytes)] = vbytes
        self.nbytes = total
        self.ptr = be.device_alloc(total, device)
        be.h2d(self.ptr, blob.ctypes.data_as(C.c_void_p), total, stream, device)
        be.wait(stream, device)
        vp = abi.VectorPartySlice()
        vp.DataType = data_type
        vp.StartingIndex = 0
        vp.Length = len(values) if counts is None else len(counts) - 1
        if counts is not None:
            vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = self.ptr, no, vo
        elif valid is not None:
            vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = self.ptr + no, 0, vo - no
        else:
            vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = self.ptr + vo, 0, 0
        self.vp = vp

    def free(self):
        if self.ptr:
            self.be.device_free(self.ptr, self.device)
            self.ptr = 0


def slice_from_pointer(ptr, data_type, length):
    """Mode-1 slice over values that already live in device memory (device-resident column cache)."""
    vp = abi.VectorPartySlice()
    vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = ptr, 0, 0
    vp.DataType, vp.Length, vp.StartingIndex = data_type, length, 0
    return vp
