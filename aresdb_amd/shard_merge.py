"""Cross-device merge of per-shard group-by partials.

The reference never merges on the device: shards of one host are processed one after another on a
single GPU and their results are re-reduced batch by batch ("append previous results + re-reduce",
query/aql_batchexecutor.go:236-251); across hosts the broker merges JSON
(broker/result_merge.go:42-141: SUM/COUNT add, MIN/MAX compare).  Here every shard runs on its own
GPU (one process per device) and the partial tables meet in ONE exchange step:

  1. all_gather of the partial sizes, then all_gather of the padded columnar partials
     (values per dimension, validity bytes per dimension, measures) — RCCL over xGMI on the GPU
     box, gloo in the CPU tests;
  2. every rank appends the partials into one DimensionVector and runs the library's own
     HashReduce (or Sort+Reduce) over them — the same re-reduce contract the Go host relies on, so
     the combine rules are the aggregate's own.

No element-wise all-reduce exists for this data: key sets differ per shard.
"""
from typing import Optional

import torch
import torch.distributed as dist

from . import abi
from .executor import DIM_WIDTHS, dimension_start_offsets


class MergedResult:
    def __init__(self, dims, measures, size, capacity):
        self.dims, self.measures, self.size, self.capacity = dims, measures, size, capacity


def _d2d(ctx, dst, src, nbytes):
    if nbytes:
        ctx.be.call("AsyncCopyDeviceToDevice", dst, src, nbytes, ctx.stream, ctx.device)


def _torch_ready(t):
    """torch fills its tensors on its own stream; the ABI copies run on the query's (non-blocking)
    stream — the fill must have finished before a copy may land in the tensor."""
    if t.is_cuda:
        torch.cuda.synchronize(t.device)


def merge_shard_results(ctx, tensor_device, group: Optional[dist.ProcessGroup] = None) -> MergedResult:
    """Merges the result of `ctx` (dim_vec[0] / measure_vec[0] / result_size) across all ranks of
    `group`; every rank returns the full merged table.  `tensor_device` is where the staging
    tensors live: the rank's GPU for the HIP backend, "cpu" for a host-memory backend."""
    plan = ctx.plan
    world = dist.get_world_size(group)
    widths = [w for w, c in zip(DIM_WIDTHS, ctx.ndw) for _ in range(c)]
    nd, mb = len(widths), plan.measure_bytes
    row_bytes = sum(widths) + nd + mb

    sizes = torch.zeros(world, dtype=torch.int64, device=tensor_device)
    mine = torch.tensor([ctx.result_size], dtype=torch.int64, device=tensor_device)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = [int(s) for s in sizes.cpu()]
    gmax, total = max(max(sizes), 1), sum(sizes)

    # columnar, padded to gmax rows: [dim values...][dim validity...][measures]
    packed = torch.zeros(gmax * row_bytes, dtype=torch.uint8, device=tensor_device)
    sect, off = [], 0
    for w in widths:
        sect.append((off, w)); off += gmax * w
    for _ in range(nd):
        sect.append((off, 1)); off += gmax
    sect.append((off, mb))
    _torch_ready(packed)
    g = ctx.result_size
    for d, w in enumerate(widths):
        vo, no = dimension_start_offsets(ctx.ndw, d, ctx.result_capacity)
        _d2d(ctx, packed.data_ptr() + sect[d][0], ctx.dim_vec[0] + vo, g * w)
        _d2d(ctx, packed.data_ptr() + sect[nd + d][0], ctx.dim_vec[0] + no, g)
    _d2d(ctx, packed.data_ptr() + sect[2 * nd][0], ctx.measure_vec[0], g * mb)
    ctx.be.wait(ctx.stream, ctx.device)

    gathered = torch.empty(world * packed.numel(), dtype=torch.uint8, device=tensor_device)
    dist.all_gather_into_tensor(gathered, packed, group=group)
    if gathered.is_cuda:
        torch.cuda.synchronize(gathered.device)

    cap = max(total, 1)
    in_dims = torch.zeros(cap * (sum(widths) + nd), dtype=torch.uint8, device=tensor_device)
    in_meas = torch.zeros(cap * mb, dtype=torch.uint8, device=tensor_device)
    out_dims = torch.zeros_like(in_dims)
    out_meas = torch.zeros_like(in_meas)
    _torch_ready(out_meas)
    base = 0
    for r in range(world):
        src = gathered.data_ptr() + r * packed.numel()
        for d, w in enumerate(widths):
            vo, no = dimension_start_offsets(ctx.ndw, d, cap)
            _d2d(ctx, in_dims.data_ptr() + vo + base * w, src + sect[d][0], sizes[r] * w)
            _d2d(ctx, in_dims.data_ptr() + no + base, src + sect[nd + d][0], sizes[r])
        _d2d(ctx, in_meas.data_ptr() + base * mb, src + sect[2 * nd][0], sizes[r] * mb)
        base += sizes[r]

    def dvec(values, hashes=None, index=None):
        dv = abi.DimensionVector()
        dv.DimValues, dv.VectorCapacity = values.data_ptr(), cap
        dv.HashValues = hashes.data_ptr() if hashes is not None else None
        dv.IndexVector = index.data_ptr() if index is not None else None
        for i, c in enumerate(ctx.ndw):
            dv.NumDimsPerDimWidth[i] = c
        return dv

    if total == 0:
        return MergedResult(out_dims, out_meas, 0, cap)
    if plan.use_hash_reduction:
        n = ctx.call("HashReduce", dvec(in_dims), in_meas.data_ptr(), dvec(out_dims), out_meas.data_ptr(), mb, total,
                     plan.agg, ctx.stream, ctx.device)
    else:
        h0 = torch.zeros(cap, dtype=torch.int64, device=tensor_device)
        h1 = torch.zeros(cap, dtype=torch.int64, device=tensor_device)
        i0 = torch.zeros(cap, dtype=torch.int32, device=tensor_device)
        i1 = torch.zeros(cap, dtype=torch.int32, device=tensor_device)
        _torch_ready(i1)
        ctx.call("InitIndexVector", i0.data_ptr(), 0, total, ctx.stream, ctx.device)
        ctx.call("Sort", dvec(in_dims, h0, i0), total, ctx.stream, ctx.device)
        n = ctx.call("Reduce", dvec(in_dims, h0, i0), in_meas.data_ptr(), dvec(out_dims, h1, i1), out_meas.data_ptr(),
                     mb, total, plan.agg, ctx.stream, ctx.device)
    ctx.be.wait(ctx.stream, ctx.device)
    return MergedResult(out_dims, out_meas, n, cap)


def merged_to_dict(ctx, res: MergedResult):
    """{((value bytes, validity), ...) -> measure} of a merged result, on the host (tests)."""
    import numpy as np
    widths = [w for w, c in zip(DIM_WIDTHS, ctx.ndw) for _ in range(c)]
    dims = res.dims.cpu().numpy()
    meas = res.measures.cpu().numpy()
    mtype = np.float64 if ctx.plan.measure_type == abi.Float64 else \
        np.uint32 if ctx.plan.measure_bytes == 4 else np.int64
    m = meas.view(mtype)
    out = {}
    order = ctx.dim_index
    for r in range(res.size):
        key = []
        for q in range(len(ctx.plan.dimensions)):
            d = order[q]
            vo, no = dimension_start_offsets(ctx.ndw, d, res.capacity)
            w = widths[d]
            key.append((bytes(dims[vo + r * w: vo + (r + 1) * w]), int(dims[no + r])))
        out[tuple(key)] = m[r]
    return out
