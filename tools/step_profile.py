import os, sys, time, socket
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.distributed as dist
import veloci_amd, bench
from veloci_amd import dist as vdist, synth
from veloci_amd.search import PartialBatch, RequestBatch
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
spec = bench.spec_for("and", 12_500_000, 100_000, 256)
data, meta = synth.generate(spec, device="cuda:0")
idx = veloci_amd.Index(data, device=0)
S = vdist.ShardedSearcher(idx, always_collective=True)
reqs = [veloci_amd.Request(r) for r in bench.make_requests("and", meta, 1024, 1024)]
batch = RequestBatch(reqs)
for _ in range(5): S.search_batch_flat(batch, stride=10)
T = {k: 0.0 for k in ("alloc", "partial", "event", "gather", "merge", "close", "total")}
N = 50
for _ in range(N):
    t0 = time.perf_counter()
    out = (np.zeros(1024, np.uint64), np.zeros(1024, np.uint32), np.zeros((1024, 10), np.uint32), np.zeros((1024, 10), np.float32), np.zeros(1024, np.int32))
    t1 = time.perf_counter()
    pb = PartialBatch(idx, batch)
    t2 = time.perf_counter()
    S._ev = (S._ev + 1) % len(S._events); pb.scanned = S._events[S._ev]; pb.scanned.record(S.stream)
    t3 = time.perf_counter()
    g = S._gather(pb)
    t4 = time.perf_counter()
    pb.merge_flat(g.data_ptr(), 1, 10, out, 0)
    t5 = time.perf_counter()
    pb.close()
    t6 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0)): T[k] += v
print({k: round(v / N * 1e3, 3) for k, v in T.items()})
dist.destroy_process_group()
