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


@dataclass
class RunLengthColumn(ResidentColumn):
    """A sort column of an ARCHIVE batch, resident in HBM as the Go host uploads it (query/aql_processor.go:1415-1429, mode 3 of
    query/iterator.hpp:117-126): [counts u32 x (runs + 1)][validity bit per run][value per run].  `length` stays the batch's
    ROW count (what the verification and the drivers size the batch by); the per-row values and validity are kept beside the
    encoded blob for the independent group-by of aresdb_amd/check.py only — the libraries get the encoded column."""
    runs: int = 0
    nulls_off: int = 0
    row_values: Optional[torch.Tensor] = None
    row_valid: Optional[torch.Tensor] = None

    @property
    def vp(self) -> abi.VectorPartySlice:
        vp = abi.VectorPartySlice()
        vp.BasePtr, vp.NullsOffset, vp.ValuesOffset = self.blob.data_ptr(), self.nulls_off, self.values_off
        vp.DataType, vp.Length, vp.StartingIndex = self.data_type, self.runs, 0
        return vp

    def values(self, np_dtype=None) -> torch.Tensor:
        return self.row_values

    def valid(self) -> Optional[torch.Tensor]:
        return self.row_valid


def _pack_run_length(values: torch.Tensor, valid: Optional[torch.Tensor], data_type: int) -> RunLengthColumn:
    n = values.numel()
    ok = torch.ones(n, dtype=torch.bool, device=values.device) if valid is None else valid
    head = torch.ones(n, dtype=torch.bool, device=values.device)
    head[1:] = (values[1:] != values[:-1]) | (ok[1:] != ok[:-1])
    starts = torch.nonzero(head).reshape(-1).to(torch.int32)
    runs = starts.numel()
    counts = torch.cat([starts, torch.tensor([n], dtype=torch.int32, device=values.device)])
    run_ok = ok[starts.long()].to(torch.uint8)
    nb = (runs + 7) // 8
    pad = nb * 8 - runs
    if pad:
        run_ok = torch.cat([run_ok, torch.zeros(pad, dtype=torch.uint8, device=values.device)])
    weights = (1 << torch.arange(8, device=values.device, dtype=torch.int32)).to(torch.uint8)
    bitmap = (run_ok.view(nb, 8) * weights).sum(dim=1, dtype=torch.int32).to(torch.uint8)
    nulls_off = _align64(4 * (runs + 1))
    values_off = nulls_off + _align64(nb)
    blob = torch.zeros(values_off + _align64(4 * runs), dtype=torch.uint8, device=values.device)
    blob[:4 * (runs + 1)] = counts.view(torch.uint8)
    blob[nulls_off:nulls_off + nb] = bitmap
    blob[values_off:values_off + 4 * runs] = values[starts.long()].contiguous().view(torch.uint8)
    return RunLengthColumn(blob, values_off, data_type, n, True, runs, nulls_off, values, valid)


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


def c3_archive_batch(n, gen: torch.Generator, device, null_fraction=0.01) -> Dict[str, ResidentColumn]:
    """The same batch as an ARCHIVE batch: rows sorted by (ts, d3) — the table's archiving sort order — with both sort
    columns run-length encoded (mode 3); nulls of a sort column sort first and form runs of their own.  The other columns
    stay as they are (mode 2).  Column order: an uncompressed column first (the batch's rows are its rows: firstColumn of
    query/aql_processor.go:571-625 carries no count vector)."""
    plain = c3_batch(n, gen, device, null_fraction)
    def col(name):
        rc = plain[name]
        return rc.values().clone(), (None if rc.valid() is None else rc.valid().clone())
    ts, tsv = col("ts")
    d3, d3v = col("d3")
    key = ts.to(torch.int64) * 8 + d3.to(torch.int64) * 2
    if tsv is not None:
        key = torch.where(tsv, key + (1 << 40), key)   # null ts first
    if d3v is not None:
        key = key + d3v.to(torch.int64)                # null d3 first inside a ts run
    order = torch.argsort(key, stable=True)
    out = {}
    for name, dt in (("d1", abi.Uint32), ("d2", abi.Uint32), ("m", abi.Float32), ("ts", abi.Uint32), ("d3", abi.Uint32)):
        v, ok = col(name)
        v = v[order].contiguous()
        ok = None if ok is None else ok[order].contiguous()
        out[name] = _pack_run_length(v, ok, dt) if name in ("ts", "d3") else _pack_column(v, ok, dt)
    return out


def c3_shard(rows, batch_rows, seed, device, null_fraction=0.01, archive=False) -> List[Dict[str, ResidentColumn]]:
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    batches = []
    done = 0
    while done < rows:
        n = min(batch_rows, rows - done)
        batches.append((c3_archive_batch if archive else c3_batch)(n, gen, device, null_fraction))
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
