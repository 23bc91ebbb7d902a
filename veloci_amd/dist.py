"""Doc-id-range sharding over the GPUs of one node: one process per GPU, one index
shard per process, RCCL all-gather of the packed per-shard partials, merge on the device.

New surface — the reference has no sharding.  The collective is the only exchange step of the path:
per batch each rank contributes `PartialBatch.nbytes` bytes to one all-gather (top-(top+skip) keys and hit counts
of every query, identical in size on every rank) and sums the batch's facet histograms with one all-reduce.
"""
import ctypes as C
import os

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


def reduce_histograms(hist, group=None):
    """Sum a batch's facet histograms (uint8 view of u32 counts) over the shards in place: one all-reduce instead of
    gathering P copies (facet counts are additive).  int32 addition == u32 addition bit for bit."""
    import torch.distributed as dist
    if hist.numel() == 0:
        return hist
    t = hist.view(torch.int32)
    if t.is_cuda and dist.get_backend(group) != "nccl":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return hist


def exchange_local(pbs):
    """Rehearsal of the exchange step with every shard in ONE process (tests: several doc-range shards on one GPU): returns the
    shard-major gathered buffer; the summed histograms end up in pbs[0]'s own histogram area, which is what its merge reads."""
    g = torch.cat([device_view(pb.device_ptr, pb.nbytes).clone() for pb in pbs])
    if pbs[0].hist_nbytes:
        hs = [device_view(pb.hist_device_ptr, pb.hist_nbytes).view(torch.int32) for pb in pbs]
        total = hs[0].clone()
        for h in hs[1:]:
            total += h
        hs[0].copy_(total)
    torch.cuda.synchronize()
    return g


class ShardedSearcher:
    """search_batch over an index sharded by doc-id range across the ranks of `group`."""

    def __init__(self, index, group=None, always_collective=False):
        """always_collective: take the collective path even with one rank (tests: RCCL all-gather / all-reduce on a 1-GPU box)."""
        import torch.distributed as dist
        self.index = index
        self.group = group
        self.world = dist.get_world_size(group)
        self.collective = self.world > 1 or always_collective
        self.stream = None
        self._views = {}
        self._garbage = []  # finished partials: freed while the NEXT step's scan runs (destroying 1024 compiled queries takes 0.14 ms)
        from . import _lib
        self.slots = int(_lib.lib().vq_partial_slots())
        # The step itself runs inside the library when the ranks talk RCCL: rank 0 takes a communicator id, the process group carries it to
        # the others, every rank joins (vq_comm_init) — from then on a step is one or two C calls (VQ_PY_COLLECTIVE=1: the older path, where
        # this module drives partial -> torch.distributed all-gather -> merge).
        self.native = bool(self.collective and dist.get_backend(group) == "nccl" and not os.environ.get("VQ_PY_COLLECTIVE"))
        if self.native:
            # every rank first finds out whether IT can load RCCL (taking an id does: cheap, nothing is joined yet) and the ranks agree — a rank that
            # could not would never enter ncclCommInitRank, and the others would wait in it for ever
            probe = (C.c_uint8 * _lib.COMM_ID_BYTES)()
            can = torch.tensor([1 if _lib.lib().vq_comm_unique_id(probe) == 0 else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(can, op=dist.ReduceOp.MIN, group=group)
            if int(can.item()) == 0:
                print(f"[veloci_amd.dist] rank {dist.get_rank(group)}: RCCL cannot be loaded on every rank; all ranks use the module's own path", flush=True)
                self.native = False
        if self.native:
            L = _lib.lib()
            rank = dist.get_rank(group)
            failed = 0
            ident = torch.zeros(_lib.COMM_ID_BYTES + 1, dtype=torch.uint8)  # the id + one byte: rank 0 has no id to give (nobody may wait for it in the init)
            try:
                if rank == 0:
                    buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
                    _lib.check(L.vq_comm_unique_id(buf))
                    ident[:_lib.COMM_ID_BYTES] = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
            except Exception as e:  # noqa: BLE001 — e.g. no RCCL library to load: every rank must learn of it (below)
                failed, err = 1, e
                ident[_lib.COMM_ID_BYTES] = 1
            ident = ident.cuda()
            dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = ident.cpu()
            if not failed and int(ident[_lib.COMM_ID_BYTES]) != 0:
                failed, err = 1, RuntimeError("rank 0 could not create a communicator id")
            if not failed:
                try:
                    _lib.check(L.vq_comm_init(index.h, self.world, rank, bytes(ident[:_lib.COMM_ID_BYTES].numpy().tobytes())))
                except Exception as e:  # noqa: BLE001
                    failed, err = 1, e
            # the ranks agree on the path: one rank on the library's exchange and another on this module's would never meet in a collective
            flag = torch.tensor([failed], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            if int(flag.item()) == 0:
                return
            if failed:
                print(f"[veloci_amd.dist] rank {rank}: the in-library exchange is not available ({err}); all ranks use the module's own path", flush=True)
            else:
                L.vq_comm_destroy(index.h)
            self.native = False
        if self.collective and dist.get_backend(group) == "nccl":
            # scans on one side stream, RCCL all-gather + merge on another that waits for the batch's scan through an event:
            # no host synchronisation between the shard scan and the collective.  (Not torch's default stream: its handle
            # is 0, which the C ABI reads as "use the index's own streams".)
            self.stream = torch.cuda.Stream()
            self.fin_stream = torch.cuda.Stream()  # all-gather + merge of batch c overlap the scan of batch c+1
            self._events = [torch.cuda.Event() for _ in range(4)]
            self._ev = 0
            index.set_streams(self.stream.cuda_stream, self.fin_stream.cuda_stream)
        if self.collective:
            index.set_allreduce(self._sum_over_ranks)

    def _sum_over_ranks(self, values):
        """vq_index_set_allreduce hook: result sizes / merged list lengths summed over the shards (a few u64 per request that needs
        them; every rank calls it with the same shape in the same order because the requests are the same)."""
        import torch.distributed as dist
        t = torch.from_numpy(values.astype(np.int64))
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        values[:] = t.cpu().numpy().astype(np.uint64)

    def _free_finished(self):
        for pb in self._garbage:
            pb.close()
        self._garbage.clear()

    def _partial(self, requests):
        from .search import PartialBatch
        if any(g.named for g in self._garbage):  # finished chunks of a one-collective step still hold their workspaces by name: hand them back first
            self._free_finished()
        pb = PartialBatch(self.index, requests)
        self._free_finished()  # (this batch's scan is queued: the GPU is busy while the host tidies up)
        if self.stream is not None:  # the collective (finish stream) must wait for this batch's scan, not for the next one's
            self._ev = (self._ev + 1) % len(self._events)  # (at most two partials are in flight: four events never collide)
            pb.scanned = self._events[self._ev]
            pb.scanned.record(self.stream)
        return pb

    def _gather(self, pb):
        if self.stream is not None:
            import torch.distributed as dist
            self.fin_stream.wait_event(pb.scanned)
            with torch.cuda.stream(self.fin_stream):
                # the partial lives in one of the index's two batch workspaces: its view and the gathered buffer are reused as long
                # as the workspace keeps its address and size (building them costs more host time than the collective itself)
                key = (pb.device_ptr, pb.nbytes)
                ent = self._views.get(key)
                if ent is None:
                    if len(self._views) > 8:
                        self._views.clear()
                    local = device_view(pb.device_ptr, pb.nbytes)
                    ent = self._views[key] = (local, torch.empty(self.world * pb.nbytes, dtype=torch.uint8, device=local.device))
                dist.all_gather_into_tensor(ent[1], ent[0], group=self.group)
                hb = pb.hist_nbytes
                if hb:  # facet histograms: summed in place, read by this rank's merge from its own partial
                    hkey = ("h", pb.hist_device_ptr, hb)
                    hv = self._views.get(hkey)
                    if hv is None:
                        hv = self._views[hkey] = device_view(pb.hist_device_ptr, hb).view(torch.int32)
                    dist.all_reduce(hv, op=dist.ReduceOp.SUM, group=self.group)
                return ent[1]
        # rehearsal backends: the partial is complete (vq_search_batch_partial synchronised the index's own stream);
        # make sure the gathered copy is, too, before the merge kernels (own stream) read it
        local = device_view(pb.device_ptr, pb.nbytes)
        gathered = gather_partials(local, self.group)
        if pb.hist_nbytes:
            reduce_histograms(device_view(pb.hist_device_ptr, pb.hist_nbytes), self.group)
        torch.cuda.synchronize()
        return gathered

    def _one_collective(self, subs, stride, out):
        """A step as a pipeline of chunks with ONE exchange: every chunk scans into its own workspace with its partial placed in the index's
        arena, back to back; when the last scan is queued the arena's used prefix is all-gathered once and every chunk merges out of the
        gathered copy.  (The host compiles chunk c+1 while the GPU scans chunk c; only chunks without facet histograms — those are summed by
        an all-reduce per chunk.)  -> False when the step has to take the per-chunk path (nothing was written to `out`)."""
        import torch.distributed as dist
        from .search import PartialBatch
        from . import _lib
        # Which path a step takes is decided BEFORE anything is scanned and from rank-invariant facts only (the requests are the same on every
        # rank): steps with facet histograms take the per-chunk path (their histograms are summed by an all-reduce per chunk).  A partial that
        # does not fit the arena — or any other failure while the chunks are queued — is an error of the step on every rank alike, never a
        # silent switch of path on one of them (the ranks' collectives would no longer match).
        if any(sb.has_facets for sb in subs):
            return False
        pbs, arena_off = [], 0
        if any(g.named for g in self._garbage):  # the previous step's chunks hold the very workspaces this step names
            self._free_finished()
        try:
            for c, sb in enumerate(subs):
                pb = PartialBatch(self.index, sb, slot=c, arena_offset=arena_off)
                pbs.append(pb)
                if c == 0:
                    self._free_finished()
                arena_off += (pb.total_nbytes + 255) // 256 * 256
            if any(pb.hist_nbytes for pb in pbs):  # (has_facets comes from the parsed requests: this cannot happen — and must not pass silently)
                raise RuntimeError("a chunk of a one-collective step carries facet histograms")
        except Exception as ex:
            # nothing of the step may stay behind: every queued chunk holds a workspace (and scans in flight) until it is closed
            for pb in pbs:
                pb.close()
            from .search import VelociError
            if isinstance(ex, VelociError) and "partial arena" in str(ex):
                # the arena is too small for this step's partials: a fact of the layout, identical on every rank (the requests and the
                # partial layout are) — every rank takes the per-chunk path, nothing was exchanged yet
                return False
            raise
        self._ev = (self._ev + 1) % len(self._events)
        scanned = self._events[self._ev]
        scanned.record(self.stream)
        self.fin_stream.wait_event(scanned)
        with torch.cuda.stream(self.fin_stream):
            key = ("arena", arena_off)
            ent = self._views.get(key)
            if ent is None:
                if len(self._views) > 8:
                    self._views.clear()
                local = device_view(self.index.partial_arena_ptr, arena_off)
                ent = self._views[key] = (local, torch.empty(self.world * arena_off, dtype=torch.uint8, device=local.device))
            dist.all_gather_into_tensor(ent[1], ent[0], group=self.group)
        base, offset = ent[1].data_ptr(), 0
        for pb, sb in zip(pbs, subs):
            pb.merge_flat(base + pb.arena_offset, self.world, stride, out, offset, shard_stride=arena_off)
            self._garbage.append(pb)
            offset += sb.n
        return True

    def search_batch(self, requests):
        """Result objects of a batch over all shards.  A request whose top + skip reaches beyond one scan's ranking (1024 hits) is paged:
        its merged page 0 is the same on every rank, so every rank sends the same continuation requests through further rounds."""
        from .search import complete_deep_pages, _as_request

        def one_round(reqs):
            pb = self._partial(reqs)
            if not self.collective:
                return pb.merge(None, 1, raise_on_error=False)
            gathered = self._gather(pb)
            return pb.merge(gathered.data_ptr(), self.world, raise_on_error=False)

        requests = [_as_request(r) for r in requests]
        pb = self._partial(requests)
        results = pb.merge(None, 1) if not self.collective else pb.merge(self._gather(pb).data_ptr(), self.world)
        if any(getattr(r, "is_page", False) for r in results):
            results = complete_deep_pages(requests, results, one_round)
            for r in results:
                if isinstance(r, Exception):
                    raise r
        return results


    def step_begin(self, requests):
        """Queue a whole step (native path); at most two steps may be in flight: begin(i + 1) before end(i)."""
        from .search import RequestBatch
        batch = requests if isinstance(requests, RequestBatch) else RequestBatch(requests)
        return shard_step_begin(self.index, batch)

    def step_end(self, step, stride=10):
        return shard_step_end(step, stride)

    def search_batch_flat(self, requests, stride=10, chunks=None):
        """Flat-output variant (see veloci_amd.search_batch_flat): no per-result Python objects.  A large batch runs as
        a pipeline of chunks over the index's two workspaces: the host compiles chunk c+1 while the GPU scans chunk c;
        every chunk has its own all-gather (equal sizes on all ranks by construction)."""
        from .search import PartialBatch, RequestBatch
        batch = requests if isinstance(requests, RequestBatch) else RequestBatch(requests)
        if self.native:
            return self.step_end(self.step_begin(batch), stride)
        if chunks is None:
            # chunks per step, from the GLOBAL doc count and the world size (identical on every rank).  Pipelining pays while a chunk's scan is long
            # next to the fixed cost of a chunk (compile hand-over, launch gaps; 256-query launches also run at a lower rate than 1024-query ones
            # on small shards).  Measured through the collective path (tools/shard_sweep.sh, ms per 1024-query step with 1 / 2 / 4 chunks):
            # 12.5 M docs per shard 1.91 / 2.09 / 2.4, 25 M 3.09 / 3.15 / 3.55, 50 M 5.30 / 5.19 / 5.45, 100 M: 4 chunks.
            per_shard = self.index.num_anchors // max(self.world, 1)
            chunks = 1 if (batch.n < 512 or (self.collective and per_shard < 40_000_000)) else 2 if (self.collective and per_shard < 80_000_000) else 4
            if os.environ.get("VQ_SHARD_CHUNKS"):
                chunks = int(os.environ["VQ_SHARD_CHUNKS"])
        subs = batch.split(chunks)

        n = batch.n  # every chunk writes its rows of one set of output arrays
        out = (np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, stride), np.uint32), np.zeros((n, stride), np.float32), np.zeros(n, np.int32))

        def finish(pb, offset):
            if not self.collective:
                pb.merge_flat(None, 1, stride, out, offset)
            else:
                gathered = self._gather(pb)
                pb.merge_flat(gathered.data_ptr(), self.world, stride, out, offset)
            self._garbage.append(pb)

        if self.collective and self.stream is not None and 1 < len(subs) <= self.slots and not os.environ.get("VQ_PER_CHUNK_COLLECTIVE"):
            done = self._one_collective(subs, stride, out)
            if done:
                return out
        inflight, offset = [], 0
        for sb in subs:
            if len(inflight) >= 2:
                finish(*inflight.pop(0))
            inflight.append((self._partial(sb), offset))
            offset += sb.n
        while inflight:
            finish(*inflight.pop(0))
        return out


class LocalExchange:
    """The exchange of `vq_comm_init_custom` for several doc-range shards living in ONE process, one thread per shard (tests, rehearsals on a
    single GPU): an all-gather and a u32 sum over device buffers, done with copies between barriers."""

    def __init__(self, indexes):
        import threading
        from . import _lib
        self.n = len(indexes)
        self.barrier = threading.Barrier(self.n)
        self.slots = [None] * self.n
        self._keep = []
        L = _lib.lib()
        for rank, index in enumerate(indexes):
            ag = _lib.ALLGATHER_FN(lambda ctx, local, gathered, nbytes, stream, rank=rank: self._allgather(rank, local, gathered, nbytes))
            ar = _lib.ALLREDUCE_U32_FN(lambda ctx, inout, count, stream, rank=rank: self._allreduce(rank, inout, count))
            self._keep += [ag, ar]
            _lib.check(L.vq_comm_init_custom(index.h, self.n, rank, ag, ar, None))

    def _allgather(self, rank, local, gathered, nbytes):
        try:
            torch.cuda.synchronize()
            self.slots[rank] = (local, nbytes)
            self.barrier.wait()
            out = device_view(gathered, self.n * nbytes)
            for r, (ptr, nb) in enumerate(self.slots):
                assert nb == nbytes, "the shards' partials differ in size"
                out[r * nbytes:(r + 1) * nbytes].copy_(device_view(ptr, nbytes))
            torch.cuda.synchronize()
            self.barrier.wait()
            return 0
        except Exception:  # noqa: BLE001 — reported through the C ABI's error code
            self.barrier.abort()
            return -1

    def _allreduce(self, rank, inout, count):
        try:
            torch.cuda.synchronize()
            self.slots[rank] = (inout, count)
            self.barrier.wait()
            total = torch.zeros(count, dtype=torch.int32, device="cuda")
            for ptr, cnt in self.slots:
                total += device_view(ptr, cnt * 4).view(torch.int32)
            torch.cuda.synchronize()
            self.barrier.wait()
            device_view(inout, count * 4).view(torch.int32).copy_(total)
            torch.cuda.synchronize()
            self.barrier.wait()
            return 0
        except Exception:  # noqa: BLE001
            self.barrier.abort()
            return -1


def shard_step_begin(index, batch):
    """`vq_shard_step_begin`: the step's compile, scans and exchange are queued; -> handle for shard_step_end."""
    from . import _lib
    h = C.c_void_p()
    _lib.check(_lib.lib().vq_shard_step_begin(index.h, batch.arr, batch.n, C.byref(h)))
    return (h, batch.n)


def shard_step_end(step, stride=10):
    """`vq_shard_step_end`: (num_hits u64[n], counts u32[n], ids u32[n, stride], scores f32[n, stride], status i32[n])."""
    from . import _lib
    h, n = step
    out = (np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, stride), np.uint32), np.zeros((n, stride), np.float32), np.zeros(n, np.int32))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _lib.check(_lib.lib().vq_shard_step_end(h, stride, *[p(a) for a in out]))
    return out


def search_shards_local(shards, requests):
    """Every doc-range shard in THIS process (tests, single-GPU rehearsals): partial per shard -> exchange_local -> merge, deep requests paged
    like ShardedSearcher.search_batch pages them."""
    from .search import PartialBatch, complete_deep_pages, _as_request

    def one_round(reqs, raise_on_error=False):
        pbs = [PartialBatch(s, reqs) for s in shards]
        g = exchange_local(pbs)
        res = pbs[0].merge(g.data_ptr(), len(shards), raise_on_error=raise_on_error)
        for pb in pbs[1:]:
            pb.merge(None, 1, raise_on_error=False)  # releases the shard's workspace
        return res

    requests = [_as_request(r) for r in requests]
    results = one_round(requests, raise_on_error=True)
    if any(getattr(r, "is_page", False) for r in results):
        results = complete_deep_pages(requests, results, one_round)
        for r in results:
            if isinstance(r, Exception):
                raise r
    return results
