"""Doc-id-range sharding over the GPUs of one node: one process per GPU, one index
shard per process, RCCL all-gather of the packed per-shard partials, merge on the device.

New surface — the reference has no sharding.  The collective is the only exchange step of the path:
per batch each rank contributes `PartialBatch.nbytes` bytes (top-(top+skip) keys, hit counts and facet
histograms of every query), identical in size on every rank.
"""
import numpy as np
import torch


def shard_range(num_docs, rank, world_size):
    """Contiguous doc-id range of `rank`: [lo, hi)."""
    lo = num_docs * rank // world_size
    hi = num_docs * (rank + 1) // world_size
    return lo, hi


class _RawDevice:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}


def device_view(ptr, nbytes, device=None):
    """Zero-copy uint8 torch view of `nbytes` of device memory at `ptr` (plumbing for RCCL)."""
    return torch.as_tensor(_RawDevice(ptr, nbytes), device=device if device is not None else "cuda")


def all_reduce_global_lens(data, group=None):
    """Posting-list lengths of the unsharded index = sum over the shards' local lengths
    (needed for the AND summation order, set_op.rs:388-393).  Collective; works on any backend."""
    import torch.distributed as dist
    for path, (offsets, anchors, scores, _) in list(data.token_to_anchor_score.items()):
        lens = np.diff(offsets.astype(np.int64))
        t = torch.from_numpy(lens.copy())
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        data.token_to_anchor_score[path] = (offsets, anchors, scores, t.cpu().numpy().astype(np.uint64))
    return data


def gather_partials(local, group=None):
    """All-gather equal-sized packed partial buffers (uint8 tensors) into one shard-major tensor."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if local.is_cuda and dist.get_backend(group) != "nccl":
        # rehearsal on a backend without device collectives (gloo): stage through the host
        host = gather_partials(local.cpu(), group)
        return host.to(local.device)
    out = torch.empty(world * local.numel(), dtype=torch.uint8, device=local.device)
    if local.is_cuda:
        dist.all_gather_into_tensor(out, local, group=group)
    else:
        chunks = list(out.chunk(world))
        dist.all_gather(chunks, local, group=group)
    return out


class ShardedSearcher:
    """search_batch over an index sharded by doc-id range across the ranks of `group`."""

    def __init__(self, index, group=None):
        import torch.distributed as dist
        self.index = index
        self.group = group
        self.world = dist.get_world_size(group)
        if self.world > 1 and dist.get_backend(group) == "nccl":
            # scans, the RCCL all-gather and the merge are ordered on ONE stream (torch's current one): no host
            # synchronisation between the shard scan and the collective
            index.set_stream(torch.cuda.current_stream().cuda_stream)

    def _gather(self, pb):
        import torch.distributed as dist
        local = device_view(pb.device_ptr, pb.nbytes)
        gathered = gather_partials(local, self.group)
        if dist.get_backend(self.group) != "nccl":
            torch.cuda.current_stream().synchronize()
        return gathered

    def search_batch(self, requests):
        from .search import PartialBatch
        pb = PartialBatch(self.index, requests)
        if self.world == 1:
            return pb.merge(None, 1)
        gathered = self._gather(pb)
        return pb.merge(gathered.data_ptr(), self.world)

    def search_batch_flat(self, requests, stride=10):
        """Flat-output variant (see veloci_amd.search_batch_flat): no per-result Python objects."""
        from .search import PartialBatch
        pb = PartialBatch(self.index, requests)
        if self.world == 1:
            return pb.merge_flat(None, 1, stride)
        gathered = self._gather(pb)
        out = pb.merge_flat(gathered.data_ptr(), self.world, stride)
        pb.close()
        return out
