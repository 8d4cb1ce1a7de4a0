"""Verification helpers for BASELINE config C3 — used by bench.py (outside the timed region) and by
the GPU tests, never by the product path.

`exact_groups` is an independent exact group-by of the synthetic shard (torch, dense accumulation
over the small key space of C3); `predict_hash_merges` applies the reference's group identity —
HashReduce identifies a group by the 32-bit murmur3 of its packed dimension row
(query/hash_reduction.cu:216-243, query/utils.cu:113-155), so distinct rows with equal hashes form
ONE group whose dimensions are those of its first row — and `compare_result` checks a fetched
result against that, key by key: group count, every (dimension row -> sum) and every representative.
"""
import numpy as np
import torch

# radices of the dense key code: value range + 1 slot for "null"
_R_TS, _R_D1, _R_D2, _R_D3 = 169, 101, 51, 3
KEY_SPACE = _R_TS * _R_D1 * _R_D2 * _R_D3
ALL_DIMS = ("ts", "d1", "d2", "d3")
_RADIX = {"ts": _R_TS, "d1": _R_D1, "d2": _R_D2, "d3": _R_D3}
_SCALE = {"ts": 3600, "d1": 1, "d2": 1, "d3": 1}  # stored dimension value = code x scale (the hourly bucket)


def key_space(dims=ALL_DIMS):
    n = 1
    for d in dims:
        n *= _RADIX[d]
    return n


def _rotl(x, r):
    return ((x << np.uint32(r)) | (x >> np.uint32(32 - r))).astype(np.uint32)


def murmur3_32_rows(values, valids):
    """murmur3_x86_32 (seed 0) of the packed dimension rows [v0..v(nd-1) little-endian uint32][nd validity
    bytes] — 5 nd bytes (query/utils.cu:113-155, query/hash_reduction.cu:216-243): the validity bytes are a
    fifth block when nd = 4, otherwise murmur's tail."""
    nd = len(values)
    assert 1 <= nd <= 4 and len(valids) == nd
    c1, c2 = np.uint32(0xcc9e2d51), np.uint32(0x1b873593)
    h = np.zeros(len(values[0]), np.uint32)
    tail = np.zeros(len(values[0]), np.uint32)
    for d, v in enumerate(valids):
        tail |= (v.astype(np.uint32) & np.uint32(0xFF)) << np.uint32(8 * d)
    with np.errstate(over="ignore"):
        for k in [v.astype(np.uint32) for v in values] + ([tail] if nd == 4 else []):
            k = (k * c1).astype(np.uint32)
            k = (_rotl(k, 15) * c2).astype(np.uint32)
            h ^= k
            h = (_rotl(h, 13) * np.uint32(5) + np.uint32(0xe6546b64)).astype(np.uint32)
        if nd < 4:
            k = (tail * c1).astype(np.uint32)
            k = (_rotl(k, 15) * c2).astype(np.uint32)
            h ^= k
        h ^= np.uint32(5 * nd)
        h ^= h >> np.uint32(16)
        h = (h * np.uint32(0x85ebca6b)).astype(np.uint32)
        h ^= h >> np.uint32(13)
        h = (h * np.uint32(0xc2b2ae35)).astype(np.uint32)
        h ^= h >> np.uint32(16)
    return h


def murmur3_128_lo64_rows(rows):
    """lo64(murmur3_x64_128) (seed 0) of every row of a uint8 matrix [n, row bytes] — the key Sort orders by
    (query/utils.cu:157-241, query/iterator.hpp:934-1025: the packed row is [dimension values, widest first][one validity
    byte per dimension])."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, nbytes = rows.shape
    c1, c2 = np.uint64(0x87c37b91114253d5), np.uint64(0x4cf5ad432745937f)
    padded = np.zeros((n, (nbytes + 15) // 16 * 16), np.uint8)
    padded[:, :nbytes] = rows
    lanes = padded.view("<u8")  # [n, 2 * blocks]

    def rotl(x, r):
        return (x << np.uint64(r)) | (x >> np.uint64(64 - r))

    def fmix(k):
        k ^= k >> np.uint64(33)
        k = k * np.uint64(0xff51afd7ed558ccd)
        k ^= k >> np.uint64(33)
        k = k * np.uint64(0xc4ceb9fe1a85ec53)
        k ^= k >> np.uint64(33)
        return k

    h1 = np.zeros(n, np.uint64)
    h2 = np.zeros(n, np.uint64)
    with np.errstate(over="ignore"):
        for b in range(nbytes // 16):
            k1, k2 = lanes[:, 2 * b].copy(), lanes[:, 2 * b + 1].copy()
            k1 = rotl(k1 * c1, 31) * c2
            h1 ^= k1
            h1 = (rotl(h1, 27) + h2) * np.uint64(5) + np.uint64(0x52dce729)
            k2 = rotl(k2 * c2, 33) * c1
            h2 ^= k2
            h2 = (rotl(h2, 31) + h1) * np.uint64(5) + np.uint64(0x38495ab5)
        tail = nbytes % 16
        if tail:
            b = nbytes // 16
            if tail > 8:
                h2 ^= rotl(lanes[:, 2 * b + 1] * c2, 33) * c1
            h1 ^= rotl(lanes[:, 2 * b] * c1, 31) * c2
        h1 ^= np.uint64(nbytes)
        h2 ^= np.uint64(nbytes)
        h1 = h1 + h2
        h2 = h2 + h1
        h1, h2 = fmix(h1), fmix(h2)
        return h1 + h2


def row_hashes_of_fetched(dims_, valids):
    """64-bit row hashes of a fetched result (NativeQuery.fetch: per-dimension value bytes in vector order, validity bytes)."""
    n = len(valids[0]) if valids else 0
    parts = [np.frombuffer(d, np.uint8).reshape(n, -1) for d in dims_] + [np.frombuffer(v, np.uint8).reshape(n, 1) for v in valids]
    return murmur3_128_lo64_rows(np.concatenate(parts, axis=1)) if n else np.zeros(0, np.uint64)


def murmur3_32_bytes(rows):
    """murmur3_x86_32 (seed 0) of every row of a uint8 matrix [n, row bytes] (query/utils.cu:113-155): any row width."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, nbytes = rows.shape
    padded = np.zeros((n, (nbytes + 3) // 4 * 4), np.uint8)
    padded[:, :nbytes] = rows
    words = padded.view("<u4")
    c1, c2 = np.uint32(0xcc9e2d51), np.uint32(0x1b873593)
    h = np.zeros(n, np.uint32)
    with np.errstate(over="ignore"):
        for b in range(nbytes // 4):
            k = (words[:, b] * c1).astype(np.uint32)
            k = (_rotl(k, 15) * c2).astype(np.uint32)
            h ^= k
            h = (_rotl(h, 13) * np.uint32(5) + np.uint32(0xe6546b64)).astype(np.uint32)
        if nbytes % 4:
            k = (words[:, nbytes // 4] * c1).astype(np.uint32)
            k = (_rotl(k, 15) * c2).astype(np.uint32)
            h ^= k
        h ^= np.uint32(nbytes)
        h ^= h >> np.uint32(16)
        h = (h * np.uint32(0x85ebca6b)).astype(np.uint32)
        h ^= h >> np.uint32(13)
        h = (h * np.uint32(0xc2b2ae35)).astype(np.uint32)
        h ^= h >> np.uint32(16)
    return h


def derive_eight(values, valids):
    """The four derived dimensions of queries.c3_plan(eight_dims=True) from the four base ones (ts bucket, d1, d2, d3): d1 + 5,
    d2 * 3, d3 + 1, d2 mod 7 — a null operand of a binary functor yields value 0, validity 0 (query/functor.hpp:660-697)."""
    _, d1, d2, d3 = [np.asarray(v).astype(np.uint32) for v in values]
    _, o1, o2, o3 = [np.asarray(v).astype(np.uint8) for v in valids]
    with np.errstate(over="ignore"):
        extra = [np.where(o1 != 0, d1 + np.uint32(5), 0), np.where(o2 != 0, d2 * np.uint32(3), 0), np.where(o3 != 0, d3 + np.uint32(1), 0),
                 np.where(o2 != 0, d2 % np.uint32(7), 0)]
    return [e.astype(np.uint32) for e in extra], [o1, o2, o3, o2]


def _codes_of_batch(b, limit=None, dims=ALL_DIMS, d1_below=90, ts_range=None, measure="m", lo=0):
    """dense key code, keep mask and float64 measure of every row of one C3 batch (torch, on the
    batch's device) for the group-by dimensions `dims` (a subset of ts-bucket, d1, d2, d3; the filter d1 < 90 and the
    measure stay); a null dimension is its own key slot, a null measure contributes 0."""
    def col(name):
        rc = b[name]
        n = rc.length if limit is None else min(lo + limit, rc.length)
        v = rc.values()[lo:n]
        ok = rc.valid()
        return v, (None if ok is None else ok[lo:n])
    d1, d1v = col("d1")
    m, mv = col("m")
    keep = d1 < d1_below
    if d1v is not None:
        keep &= d1v
    if ts_range is not None:  # the Go host's time filters: ts >= from, ts < to (a null ts fails both)
        ts, tsv = col("ts")
        keep &= (ts >= int(ts_range[0])) & (ts < int(ts_range[1]))
        if tsv is not None:
            keep &= tsv
    def code(v, ok, null_code):
        v = v.to(torch.int64)
        return v if ok is None else torch.where(ok, v, torch.full_like(v, null_code))
    c = None
    for name in dims:
        v, ok = col(name)
        if name == "ts":
            v = torch.div(v, 3600, rounding_mode="floor")
        k = code(v, ok, _RADIX[name] - 1)
        c = k if c is None else c * _RADIX[name] + k
    if measure != "m":  # an integer column summed (null -> 0): "count" takes the row counts instead
        m, mv = col(measure if measure != "count" else "d1")
    mm = m.to(torch.float64)
    if mv is not None:
        mm = torch.where(mv, mm, torch.zeros_like(mm))
    return c, keep, mm


def exact_groups(batches, limit_first_batch=None, dims=ALL_DIMS, d1_below=90, ts_range=None, measure="m", slices=None):
    """Exact group-by of the C3 query (group-by dimensions `dims`) over `batches` (all rows, or the first
    `limit_first_batch` rows of the first batch only).  Returns numpy arrays (code, sum, first_row, rows) of the groups.
    measure: "m" (SUM of the float measure), another column's name (SUM of that integer column, nulls as 0) or "count"
    (COUNT(*): the sums are the row counts).  slices: instead of whole batches, the row ranges [(batch index, first row,
    rows), ...] in that order (a sample spread over the shard)."""
    dev = batches[0]["m"].blob.device
    space = key_space(dims)
    # A small key space is spread over `salt` sub-slots per key (row mod salt) and folded at the end: index_add_ /
    # scatter_reduce_ are atomics, and 10^9 of them on ~100 addresses take minutes (the 153-group leg of bench.py
    # spent 200 s here)
    salt = 1 if space >= (1 << 20) else max(1, min(8192, (1 << 24) // space))
    acc = torch.zeros(space * salt, dtype=torch.float64, device=dev)
    cnt = torch.zeros(space * salt, dtype=torch.int64, device=dev)
    first = torch.full((space * salt,), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
    offset = 0
    if slices is not None:
        work = [(batches[bi], lo, n) for bi, lo, n in slices]
    else:
        work = [(b, 0, limit_first_batch) for b in (batches[:1] if limit_first_batch is not None else batches)]
    for b, lo, limit in work:
        c, keep, mm = _codes_of_batch(b, limit, dims, d1_below, ts_range, measure, lo)
        rows = torch.arange(offset, offset + c.numel(), dtype=torch.int64, device=dev)[keep]
        idx = c[keep]
        if salt > 1:
            idx = idx * salt + rows % salt
        acc.index_add_(0, idx, mm[keep])
        cnt.index_add_(0, idx, torch.ones_like(idx))
        first.scatter_reduce_(0, idx, rows, reduce="amin", include_self=True)
        offset += c.numel()
        del c, keep, mm, idx, rows
    if salt > 1:
        acc = acc.view(space, salt).sum(dim=1)
        cnt = cnt.view(space, salt).sum(dim=1)
        first = first.view(space, salt).amin(dim=1)
    live = torch.nonzero(cnt > 0).reshape(-1)
    if measure == "count":
        acc = cnt.to(torch.float64)
    return (live.cpu().numpy(), acc[live].cpu().numpy(), first[live].cpu().numpy(), cnt[live].cpu().numpy())


def decode_codes(code, dims=ALL_DIMS):
    """(values[nd], valids[nd]) of dense key codes, as the dimension vector stores them: a null
    dimension holds value 0 with validity 0 (the synthetic columns keep zeros under nulls)."""
    c = code.astype(np.int64)
    parts = []
    for name in reversed(dims):
        parts.append((name, c % _RADIX[name]))
        c = c // _RADIX[name]
    out_v, out_ok = [], []
    for name, v in reversed(parts):
        ok = v != _RADIX[name] - 1
        out_v.append(np.where(ok, v * _SCALE[name], 0).astype(np.uint32))
        out_ok.append(ok.astype(np.uint8))
    return out_v, out_ok


def encode_rows(values, valids, dims=ALL_DIMS):
    """dense key codes of fetched dimension rows (inverse of decode_codes)."""
    c = None
    for name, v, ok in zip(dims, values, valids):
        v = np.asarray(v).view(np.uint32).astype(np.int64)
        k = np.where(np.asarray(ok).astype(bool), v // _SCALE[name], _RADIX[name] - 1)
        c = k if c is None else c * _RADIX[name] + k
    return c


def predict_hash_merges(code, sums, first_row, dims=ALL_DIMS, eight=False):
    """Groups as HashReduce forms them: one per distinct 32-bit hash; representative = the member
    whose first row comes first; value = sum over the members.  eight: the rows carry derive_eight's four dimensions too."""
    values, valids = decode_codes(code, dims)
    if eight:
        ev, eo = derive_eight(values, valids)
        values, valids = values + ev, valids + eo
        h = murmur3_32_bytes(np.concatenate([v.astype("<u4").view(np.uint8).reshape(-1, 4) for v in values] +
                                            [o.astype(np.uint8).reshape(-1, 1) for o in valids], axis=1))
    else:
        h = murmur3_32_rows(values, valids)
    order = np.lexsort((first_row, h))
    hs = h[order]
    head = np.ones(len(hs), bool)
    head[1:] = hs[1:] != hs[:-1]
    seg = np.cumsum(head) - 1
    merged_sum = np.zeros(int(seg[-1]) + 1 if len(seg) else 0, np.float64)
    np.add.at(merged_sum, seg, sums[order])
    rep_code = code[order][head]
    return rep_code, merged_sum, int(len(code) - len(rep_code))


def compare_tables(got_code, got_sum, want_code, want_sum, rel=0.0):
    """None when the two group tables hold the same keys with the same sums, else a description."""
    if len(got_code) != len(want_code):
        return f"group count {len(got_code)} != expected {len(want_code)}"
    go, wo = np.argsort(got_code, kind="stable"), np.argsort(want_code, kind="stable")
    gc, wc = got_code[go], want_code[wo]
    if len(gc) > 1 and (gc[1:] == gc[:-1]).any():
        return "duplicate dimension rows in the result"
    bad = np.nonzero(gc != wc)[0]
    if len(bad):
        return f"{len(bad)} dimension rows differ (first: got code {gc[bad[0]]}, expected {wc[bad[0]]})"
    gs, ws = got_sum[go], want_sum[wo]
    tol = rel * np.maximum(1.0, np.abs(ws))
    bad = np.nonzero(np.abs(gs - ws) > tol)[0]
    if len(bad):
        return f"{len(bad)} sums differ (first: key {gc[bad[0]]} got {gs[bad[0]]!r} expected {ws[bad[0]]!r})"
    return None


def compare_result(fetched, expected, hash_identity=True, rel=0.0, dims=ALL_DIMS, measure_dtype=np.float64, ordered=False, eight=False):
    """fetched = (dims, valids, measures) of NativeQuery.fetch(); expected = exact_groups(...).
    hash_identity: the result comes from HashReduce (groups are hashes); False: Sort+Reduce on the
    64-bit hash (exact groups at these cardinalities).  measure_dtype: how the fetched measure bytes read (float64 sums,
    uint32 counts, int64 integer sums).  ordered: the rows must come in strictly ascending order of their 64-bit row hash
    (what Sort + Reduce leaves: query/sort_reduce.cu:118-249).  Returns a report dict."""
    dims_, valids, meas = fetched
    code, sums, first_row, _ = expected
    got_code = encode_rows([np.frombuffer(d, np.uint32) for d in dims_], [np.frombuffer(v, np.uint8) for v in valids], dims)
    got_sum = np.frombuffer(meas, measure_dtype).astype(np.float64)
    merged = 0
    if eight:  # the four derived dimensions are functions of the first four
        base_v = [np.frombuffer(d, np.uint32) for d in dims_[:4]]
        base_o = [np.frombuffer(v, np.uint8) for v in valids[:4]]
        ev, eo = derive_eight(base_v, base_o)
        for k in range(4):
            if not (np.array_equal(np.frombuffer(dims_[4 + k], np.uint32), ev[k]) and np.array_equal(np.frombuffer(valids[4 + k], np.uint8) != 0, eo[k] != 0)):
                return {"status": f"MISMATCH: derived dimension {4 + k} is not the function of its base dimension", "groups": int(len(got_code)),
                        "expected_groups": int(len(code)), "distinct_dimension_rows": int(len(code)), "merged_by_32bit_hash": 0}
    if hash_identity:
        want_code, want_sum, merged = predict_hash_merges(code, sums, first_row, dims, eight)
    else:
        want_code, want_sum = code, sums
    why = compare_tables(got_code, got_sum, want_code, want_sum, rel)
    if why is None and ordered and len(got_code) > 1:
        h = row_hashes_of_fetched(dims_, valids)
        if not (h[1:] > h[:-1]).all():
            why = f"rows are not in ascending order of their 64-bit hash (first descent at row {int(np.nonzero(h[1:] <= h[:-1])[0][0]) + 1})"
    return {"status": "ok" if why is None else "MISMATCH: " + why, "groups": int(len(got_code)),
            "expected_groups": int(len(want_code)), "distinct_dimension_rows": int(len(code)),
            "merged_by_32bit_hash": merged}


def fetched_from_columnar(dims, measures, size, capacity, nd=4):
    """(dims, valids, measures) in NativeQuery.fetch() form from a device-layout dimension vector of
    nd 4-byte dimensions (values per dimension with stride `capacity`, then validity bytes) and a
    float64 measure vector."""
    dims = np.asarray(dims, np.uint8)
    values = [dims[4 * capacity * d: 4 * capacity * d + 4 * size].copy() for d in range(nd)]
    valids = [dims[4 * capacity * nd + capacity * d: 4 * capacity * nd + capacity * d + size].copy() for d in range(nd)]
    return values, valids, np.asarray(measures, np.uint8)[:8 * size].copy()


def hash_groups_of_rows(values, valids, measures):
    """Groups as HashReduce forms them over explicit rows (4 uint32 dimension columns, their validity
    bytes, float64 measures; rows in input order): {representative dimension row -> sum}.  Distinct
    rows with equal 32-bit hashes merge under the first one."""
    h = murmur3_32_rows(values, valids)
    order = np.argsort(h, kind="stable")  # stable: input order inside a hash
    hs = h[order]
    head = np.ones(len(hs), bool)
    head[1:] = hs[1:] != hs[:-1]
    seg = np.cumsum(head) - 1
    sums = np.zeros(int(seg[-1]) + 1 if len(seg) else 0, np.float64)
    np.add.at(sums, seg, np.asarray(measures, np.float64)[order])
    rep = order[head]
    out = {}
    for i, r in enumerate(rep):
        out[tuple((int(values[d][r]), int(valids[d][r])) for d in range(4))] = sums[i]
    return out
