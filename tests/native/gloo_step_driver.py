"""One rank of a world-size-2 run of the in-library sharded step on the CPU (tests/test_dist_gloo.py): vq_shard_step_begin / _end with two steps
in flight over `vq_comm_init_custom`, the exchange being torch.distributed's gloo backend.  The library is the host-stub build
(`make -C veloci_amd/csrc hoststub`: device memory is host memory, launches do nothing — VQ_STUB_NOOP_LAUNCH=1), so the results are garbage;
what runs for real is everything around the kernels on TWO ranks: compile, pack, the order and sizes of the collectives of pipelined steps,
merge bookkeeping, copy-out — and what happens when one rank's step fails between its collectives.
usage: RANK=r WORLD_SIZE=2 MASTER_PORT=p python gloo_step_driver.py <ok|fail>"""
import ctypes as C
import datetime
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import veloci_amd  # noqa: E402
from veloci_amd import _lib, synth  # noqa: E402
from veloci_amd import dist as vdist  # noqa: E402

scenario = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert "host_stub" in _lib.lib_path() and os.environ.get("VQ_STUB_NOOP_LAUNCH") == "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=20))

N = 200_000
lo, hi = vdist.shard_range(N, rank, world)
spec = synth.SynthSpec(num_docs=N, num_terms=3000, triples=2, extra_probe_dfs=(500, 20_000), background_terms=20)
data, meta = synth.generate(spec, doc_lo=lo, doc_hi=hi, device="cpu")
vdist.all_reduce_global_lens(data)
index = veloci_amd.Index(data, device=0, doc_lo=lo, doc_hi=hi)
L = _lib.lib()
calls = []  # (kind, bytes): must be the same sequence on both ranks


def host_view(ptr, nbytes):
    return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr))


def allgather(ctx, local, gathered, nbytes, stream):
    try:
        calls.append(("gather", int(nbytes)))
        mine = torch.from_numpy(host_view(local, nbytes).copy())
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, mine)
        out = host_view(gathered, nbytes * world)
        for r in range(world):
            out[r * nbytes:(r + 1) * nbytes] = parts[r].numpy()
        return 0
    except Exception as ex:  # noqa: BLE001 — reported through the C ABI's error code
        print(f"[rank {rank}] all-gather failed: {type(ex).__name__}", file=sys.stderr)
        return -1


def allreduce(ctx, inout, count, stream):
    try:
        calls.append(("sum", int(count)))
        t = torch.from_numpy(host_view(inout, count * 4).view(np.int32).copy())
        dist.all_reduce(t)
        host_view(inout, count * 4).view(np.int32)[:] = t.numpy()
        return 0
    except Exception as ex:  # noqa: BLE001
        print(f"[rank {rank}] all-reduce failed: {type(ex).__name__}", file=sys.stderr)
        return -1


ag, ar = _lib.ALLGATHER_FN(allgather), _lib.ALLREDUCE_U32_FN(allreduce)
_lib.check(L.vq_comm_init_custom(index.h, world, rank, ag, ar, None))

a, b = meta.triples
reqs = []
for i in range(600):
    t = (a, b)[i & 1]
    reqs.append((synth.req_and(list(t)), synth.req_or(list(t), top=20), synth.req_single(meta.extra_probes[i % 2]),
                 dict(synth.req_single(meta.background[i % 20]), facets=[{"field": "cat", "top": 5}]))[i % 4])
batch = veloci_amd.RequestBatch([veloci_amd.Request(r) for r in reqs])
plain = veloci_amd.RequestBatch([veloci_amd.Request(r) for r in reqs if "facets" not in r])

steps = 0
prev = None
for s in range(5):  # two steps in flight, with and without facet histograms (an all-reduce beside the all-gather)
    h = vdist.shard_step_begin(index, batch if s % 2 == 0 else plain)
    if prev is not None:
        out = vdist.shard_step_end(prev, 20)
        assert not out[4].any(), out[4][out[4] != 0][:4]
        steps += 1
    prev = h
out = vdist.shard_step_end(prev, 20)
steps += 1
seqs = [None] * world
dist.all_gather_object(seqs, calls)
assert all(sq == seqs[0] for sq in seqs), "the ranks' collectives differ"
assert sum(1 for k, _ in calls if k == "gather") >= steps and any(k == "sum" for k, _ in calls)

if scenario == "fail":
    # rank 1's next step fails in the middle (a launch throws): it must come back with an error — and so must rank 0, whose exchange of that
    # step will never be joined; afterwards the communicator is down on both, not half-alive
    dist.barrier()
    if rank == 1:
        L.vq_stub_fail_launches_after.argtypes = [C.c_long]
        L.vq_stub_fail_launches_after(1)
    err = None
    try:
        vdist.shard_step_end(vdist.shard_step_begin(index, plain), 20)
    except veloci_amd.VelociError as ex:
        err = str(ex)
    assert err is not None, "a step one rank could not run came back without an error"
    if rank == 1:
        assert "injected failure" in err, err
        L.vq_stub_fail_launches_after(-1)
        # (this rank leaves: what rank 0 is waiting for will not come — its gloo call ends with an error when the peer is gone or its time is up)
    else:
        assert "all-gather failed" in err, err
    try:
        vdist.shard_step_begin(index, plain)
        raise AssertionError("a step was accepted on a communicator that is down")
    except veloci_amd.VelociError as ex:
        assert "communicator is down" in str(ex), str(ex)
    _lib.check(L.vq_comm_destroy(index.h))
    # without a communicator the shard answers alone again
    out = vdist.shard_step_end(vdist.shard_step_begin(index, plain), 20)
    assert len(out[0]) == plain.n
    print("GLOO_STEP_DRIVER_OK " + json.dumps({"rank": rank, "steps": steps, "collectives": len(calls), "failed_step_error": err[:120]}))
    sys.stdout.flush()
    os._exit(0)  # (no orderly shutdown of a process group whose peer has failed)

dist.barrier()
print("GLOO_STEP_DRIVER_OK " + json.dumps({"rank": rank, "steps": steps, "collectives": len(calls)}))
dist.destroy_process_group()
