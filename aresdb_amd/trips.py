"""The reference's example table at bench scale (examples/1k_trips/schema/trips.json): request_at Uint32, city_id Uint16,
status SmallEnum (one byte), fare Float32 — synthetic shard, the two example queries with city_id as a second dimension
(queries/total_fare.aql: SUM(fare) through HashReduce; queries/total_trips.aql: COUNT(*) through Sort + Reduce — the Go
compiler sends every aggregate but SUM_SIGNED / SUM_FLOAT down the sort path, query/aql_context.go:426-434), and their
key-level verification.  Verification only computes on torch / numpy; nothing here is on the product path."""
from typing import Dict, List

import numpy as np
import torch

from . import abi
from .executor import Binary, Col, Const, DimensionSpec, QueryPlan
from .workload import ResidentColumn, _align64

COLUMNS = (("request_at", abi.Uint32), ("city_id", abi.Uint16), ("status", abi.Uint8), ("fare", abi.Float32))
DAYS, CITIES, COMPLETED = 8, 400, 2
_HOURS = DAYS * 24
_R_HOUR, _R_CITY = _HOURS + 1, CITIES + 1  # + 1 slot for "null"
_TORCH = {abi.Uint32: torch.int32, abi.Uint16: torch.int16, abi.Uint8: torch.uint8, abi.Float32: torch.float32}
_BYTES = {abi.Uint32: 4, abi.Uint16: 2, abi.Uint8: 1, abi.Float32: 4}


def _pack(values: torch.Tensor, valid, data_type) -> ResidentColumn:
    n = values.numel()
    raw = values.contiguous().view(torch.uint8)
    if valid is None:
        blob = torch.zeros(_align64(raw.numel()) + 64, dtype=torch.uint8, device=values.device)
        blob[:raw.numel()] = raw
        return ResidentColumn(blob, 0, data_type, n, False)
    nb = (n + 7) // 8
    v = valid.to(torch.uint8)
    if nb * 8 != n:
        v = torch.cat([v, torch.zeros(nb * 8 - n, dtype=torch.uint8, device=v.device)])
    weights = (1 << torch.arange(8, device=v.device, dtype=torch.int32)).to(torch.uint8)
    bitmap = (v.view(nb, 8) * weights).sum(dim=1, dtype=torch.int32).to(torch.uint8)
    off = _align64(nb)
    blob = torch.zeros(off + _align64(raw.numel()) + 64, dtype=torch.uint8, device=values.device)
    blob[:nb] = bitmap
    blob[off:off + raw.numel()] = raw
    return ResidentColumn(blob, off, data_type, n, True)


def column_values(rc: ResidentColumn) -> torch.Tensor:
    t = rc.blob[rc.values_off:rc.values_off + _BYTES[rc.data_type] * rc.length]
    return t.view(_TORCH[rc.data_type])


def trips_batch(n, gen, device, null_fraction=0.01) -> Dict[str, ResidentColumn]:
    """request_at uniform over DAYS days, city_id Zipf-free uniform over CITIES, status: 70 % `completed` (= 2), the rest
    spread over 0, 1, 3, 4; fare in quarter steps (float64 sums exact in any order)."""
    def ri(hi):
        return torch.randint(0, hi, (n,), dtype=torch.int32, device=device, generator=gen)
    u = torch.rand((n,), dtype=torch.float32, device=device, generator=gen)
    other = ri(4)
    other = other + (other >= COMPLETED).to(torch.int32)
    cols = {"request_at": ri(86400 * DAYS), "city_id": ri(CITIES).to(torch.int16),
            "status": torch.where(u < 0.7, torch.full_like(other, COMPLETED), other).to(torch.uint8),
            "fare": ri(400).to(torch.float32) * 0.25}
    del u, other
    out = {}
    for name, dt in COLUMNS:
        valid = None
        if null_fraction > 0:
            valid = torch.rand((n,), dtype=torch.float32, device=device, generator=gen) >= null_fraction
            cols[name] = torch.where(valid, cols[name], torch.zeros((), dtype=cols[name].dtype, device=device))
        out[name] = _pack(cols[name], valid, dt)
    return out


def trips_shard(rows, batch_rows, seed, device, null_fraction=0.01) -> List[Dict[str, ResidentColumn]]:
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out, done = [], 0
    while done < rows:
        n = min(batch_rows, rows - done)
        out.append(trips_batch(n, gen, device, null_fraction))
        done += n
    return out


TIME_RANGE = (86400 // 2, 86400 * DAYS - 86400 // 2)  # "from" and "to" of the query's time filter: 7 of the 8 days


def trips_plan(count=False, time_range=TIME_RANGE):
    filters = [Binary(abi.GreaterThanOrEqual, Col("request_at"), Const(int(time_range[0]))),
               Binary(abi.LessThan, Col("request_at"), Const(int(time_range[1]))),
               Binary(abi.Equal, Col("status"), Const(COMPLETED))]
    dims = [DimensionSpec(Binary(abi.Floor, Col("request_at"), Const(3600)), abi.Uint32), DimensionSpec(Col("city_id"), abi.Uint16)]
    if count:
        return QueryPlan(filters=filters, dimensions=dims, measure=Const(1), agg=abi.AGGR_SUM_UNSIGNED, measure_type=abi.Uint32,
                         use_hash_reduction=False)
    return QueryPlan(filters=filters, dimensions=dims, measure=Col("fare"), agg=abi.AGGR_SUM_FLOAT, measure_type=abi.Float64,
                     use_hash_reduction=True)


def _valid(rc):
    return rc.valid()


def exact_groups(batches, time_range=TIME_RANGE):
    """Exact group-by (torch, dense over hour x city codes): numpy (code, sum(fare), count, first row) of the groups."""
    dev = batches[0]["fare"].blob.device
    space = _R_HOUR * _R_CITY
    salt = 64  # spreads the atomics of a small key space
    acc = torch.zeros(space * salt, dtype=torch.float64, device=dev)
    cnt = torch.zeros(space * salt, dtype=torch.int64, device=dev)
    first = torch.full((space * salt,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    offset = 0
    for b in batches:
        ts, city, st, fare = (column_values(b[k]) for k in ("request_at", "city_id", "status", "fare"))
        ok = {k: _valid(b[k]) for k in b}
        keep = (ts >= time_range[0]) & (ts < time_range[1]) & (st == COMPLETED)
        for k in ("request_at", "status"):
            if ok[k] is not None:
                keep &= ok[k]
        hour = torch.div(ts, 3600, rounding_mode="floor").to(torch.int64)  # (request_at is valid wherever keep is)
        c = city.to(torch.int64) & 0xFFFF
        if ok["city_id"] is not None:
            c = torch.where(ok["city_id"], c, torch.full_like(c, CITIES))
        code = hour * _R_CITY + c
        mm = fare.to(torch.float64)
        if ok["fare"] is not None:
            mm = torch.where(ok["fare"], mm, torch.zeros_like(mm))
        rows = torch.arange(offset, offset + code.numel(), dtype=torch.int64, device=dev)[keep]
        idx = code[keep] * salt + rows % salt
        acc.index_add_(0, idx, mm[keep])
        cnt.index_add_(0, idx, torch.ones_like(idx))
        first.scatter_reduce_(0, idx, rows, reduce="amin", include_self=True)
        offset += code.numel()
        del ts, city, st, fare, keep, hour, c, code, mm, rows, idx
    acc, cnt, first = acc.view(space, salt).sum(1), cnt.view(space, salt).sum(1), first.view(space, salt).amin(1)
    live = torch.nonzero(cnt > 0).reshape(-1)
    return live.cpu().numpy(), acc[live].cpu().numpy(), cnt[live].cpu().numpy(), first[live].cpu().numpy()


def _rotl(x, r):
    return ((x << np.uint32(r)) | (x >> np.uint32(32 - r))).astype(np.uint32)


def murmur3_32_packed(hour_value, city_value, city_ok):
    """murmur3_x86_32 (seed 0) of the packed dimension row [hour bucket u32][city u16][validity bytes 1, city_ok]: eight
    bytes, two blocks, no tail (query/utils.cu:113-155 over the row query/hash_reduction.cu:216-243 packs)."""
    c1, c2 = np.uint32(0xcc9e2d51), np.uint32(0x1b873593)
    w0 = hour_value.astype(np.uint32)
    w1 = (city_value.astype(np.uint32) & np.uint32(0xFFFF)) | np.uint32(1 << 16) | (city_ok.astype(np.uint32) << np.uint32(24))
    h = np.zeros(len(w0), np.uint32)
    with np.errstate(over="ignore"):
        for k in (w0, w1):
            k = (k * c1).astype(np.uint32)
            k = (_rotl(k, 15) * c2).astype(np.uint32)
            h ^= k
            h = (_rotl(h, 13) * np.uint32(5) + np.uint32(0xe6546b64)).astype(np.uint32)
        h ^= np.uint32(8)
        h ^= h >> np.uint32(16)
        h = (h * np.uint32(0x85ebca6b)).astype(np.uint32)
        h ^= h >> np.uint32(13)
        h = (h * np.uint32(0xc2b2ae35)).astype(np.uint32)
        h ^= h >> np.uint32(16)
    return h


def _decode(code):
    hour, c = code // _R_CITY, code % _R_CITY
    ok = c != CITIES
    return (hour * 3600).astype(np.uint32), np.where(ok, c, 0).astype(np.uint32), ok.astype(np.uint8)


def compare(fetched, expected, count=False):
    """fetched = NativeQuery.fetch() (dimension vector order: [hour u32][city u16]); expected = exact_groups(...).  SUM(fare)
    comes from HashReduce: groups are 32-bit hashes, distinct rows with equal hashes merge under the first row; COUNT(*)
    from Sort + Reduce on the 64-bit hash (exact at this cardinality)."""
    dims, valids, meas = fetched
    code, sums, counts, first = expected
    g_hour = np.frombuffer(dims[0], np.uint32).astype(np.int64)
    g_city = np.frombuffer(dims[1], np.uint16).astype(np.int64)
    g_ok = np.frombuffer(valids[1], np.uint8).astype(bool)
    if not np.frombuffer(valids[0], np.uint8).all():
        return {"status": "MISMATCH: a null hour bucket survived the time filter"}
    got_code = (g_hour // 3600) * _R_CITY + np.where(g_ok, g_city, CITIES)
    got_val = np.frombuffer(meas, np.uint32).astype(np.float64) if count else np.frombuffer(meas, np.float64)
    want_val = counts.astype(np.float64) if count else sums
    want_code, merged = code, 0
    if not count:
        hv, cv, cok = _decode(code)
        h = murmur3_32_packed(hv, cv, cok)
        order = np.lexsort((first, h))
        hs = h[order]
        head = np.ones(len(hs), bool)
        head[1:] = hs[1:] != hs[:-1]
        seg = np.cumsum(head) - 1
        ms = np.zeros(int(seg[-1]) + 1 if len(seg) else 0, np.float64)
        np.add.at(ms, seg, want_val[order])
        want_code, want_val, merged = code[order][head], ms, int(len(code) - head.sum())
    why = None
    if len(got_code) != len(want_code):
        why = f"group count {len(got_code)} != expected {len(want_code)}"
    else:
        go, wo = np.argsort(got_code, kind="stable"), np.argsort(want_code, kind="stable")
        if (got_code[go] != want_code[wo]).any():
            why = f"{int((got_code[go] != want_code[wo]).sum())} dimension rows differ"
        elif (got_val[go] != want_val[wo]).any():
            bad = np.nonzero(got_val[go] != want_val[wo])[0]
            why = f"{len(bad)} values differ (first: code {got_code[go][bad[0]]} got {got_val[go][bad[0]]!r} expected {want_val[wo][bad[0]]!r})"
    return {"status": "ok" if why is None else "MISMATCH: " + why, "groups": int(len(got_code)), "expected_groups": int(len(want_code)),
            "merged_by_32bit_hash": merged}
