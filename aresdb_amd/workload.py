"""Synthetic fact-table shards for the BASELINE configurations, generated on the
device with torch's counter-based generator and laid out exactly like the batches the Go host
uploads (query/aql_processor.go:1388-1431): one device allocation per column per batch,
[validity bitmap, 64-byte padded][values], handed to the ABI as mode-2 VectorPartySlices
(mode 1 = values only when the shard is generated without nulls).

torch is used for device memory and random numbers only; nothing here computes query results.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from . import abi

C3_COLUMNS = (("ts", abi.Uint32), ("d1", abi.Uint32), ("d2", abi.Uint32), ("d3", abi.Uint32),
              ("m", abi.Float32))


def _align64(n):
    return (n + 63) // 64 * 64


@dataclass
class ResidentColumn:
    """One column of one batch, resident in HBM."""
    blob: torch.Tensor        # uint8: [bitmap (optional)][values]
    values_off: int
    data_type: int
    length: int
    has_nulls: bool

    @property
    def vp(self) -> abi.VectorPartySlice:
        vp = abi.VectorPartySlice()
        base = self.blob.data_ptr()
        if self.has_nulls:
            vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = base, 0, self.values_off
        else:
            vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = base + self.values_off, 0, 0
        vp.DataType, vp.Length, vp.StartingIndex = self.data_type, self.length, 0
        return vp

    def values(self, np_dtype=None) -> torch.Tensor:
        t = self.blob[self.values_off:self.values_off + 4 * self.length]
        return t.view(torch.float32 if self.data_type == abi.Float32 else torch.int32)

    def valid(self) -> Optional[torch.Tensor]:
        if not self.has_nulls:
            return None
        bits = self.blob[:(self.length + 7) // 8]
        shifts = torch.arange(8, device=bits.device, dtype=torch.uint8)
        return ((bits[:, None] >> shifts) & 1).reshape(-1)[:self.length].bool()


def _pack_column(values: torch.Tensor, valid: Optional[torch.Tensor], data_type: int) -> ResidentColumn:
    n = values.numel()
    raw = values.contiguous().view(torch.uint8)
    if valid is None:
        blob = torch.empty(_align64(raw.numel()), dtype=torch.uint8, device=values.device)
        blob[:raw.numel()] = raw
        return ResidentColumn(blob, 0, data_type, n, False)
    nb = (n + 7) // 8
    pad = nb * 8 - n
    v = valid.to(torch.uint8)
    if pad:
        v = torch.cat([v, torch.zeros(pad, dtype=torch.uint8, device=v.device)])
    weights = (1 << torch.arange(8, device=v.device, dtype=torch.int32)).to(torch.uint8)
    bitmap = (v.view(nb, 8) * weights).sum(dim=1, dtype=torch.int32).to(torch.uint8)
    off = _align64(nb)
    blob = torch.zeros(off + _align64(raw.numel()), dtype=torch.uint8, device=values.device)
    blob[:nb] = bitmap
    blob[off:off + raw.numel()] = raw
    return ResidentColumn(blob, off, data_type, n, True)


def _zipf_cdf(alpha, k, device):
    w = 1.0 / torch.arange(1, k + 1, dtype=torch.float64, device=device) ** alpha
    return (torch.cumsum(w, 0) / w.sum()).to(torch.float32)


def c3_batch(n, gen: torch.Generator, device, null_fraction=0.01) -> Dict[str, ResidentColumn]:
    """One batch of BASELINE config C3: ts uniform over 7 days, d1 uniform [0,100),
    d2 Zipf(1.1) over [0,50), d3 uniform [0,2), m float32 uniform [0,100) (quarter steps so float64
    sums are exact in any order)."""
    def ri(hi):
        return torch.randint(0, hi, (n,), dtype=torch.int32, device=device, generator=gen)
    cols = {"ts": ri(86400 * 7), "d1": ri(100)}
    u = torch.rand((n,), dtype=torch.float32, device=device, generator=gen)
    cols["d2"] = torch.searchsorted(_zipf_cdf(1.1, 50, device), u).clamp_(max=49).to(torch.int32)
    del u
    cols["d3"] = ri(2)
    cols["m"] = ri(400).to(torch.float32) * 0.25
    out = {}
    for name, dt in C3_COLUMNS:
        valid = None
        if null_fraction > 0:
            valid = torch.rand((n,), dtype=torch.float32, device=device, generator=gen) >= null_fraction
            # a null slot holds the zero the memstore initialised it with (HostAlloc zero-fills)
            cols[name] = torch.where(valid, cols[name], torch.zeros((), dtype=cols[name].dtype, device=device))
        out[name] = _pack_column(cols[name], valid, dt)
    return out


def c3_shard(rows, batch_rows, seed, device, null_fraction=0.01) -> List[Dict[str, ResidentColumn]]:
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    batches = []
    done = 0
    while done < rows:
        n = min(batch_rows, rows - done)
        batches.append(c3_batch(n, gen, device, null_fraction))
        done += n
    return batches


def batch_to_host(batch: Dict[str, ResidentColumn], limit=None, lo=0):
    """numpy copies of rows [lo, lo + limit) of a batch (values, validity) — what the CPU baseline leg uploads."""
    cols, valid = {}, {}
    for name, rc in batch.items():
        n = rc.length if limit is None else min(lo + limit, rc.length)
        v = rc.values()[lo:n].cpu().numpy()
        cols[name] = (rc.data_type, v.view(np.uint32) if rc.data_type != abi.Float32 else v)
        m = rc.valid()
        valid[name] = None if m is None else m[lo:n].cpu().numpy()
    return cols, valid
