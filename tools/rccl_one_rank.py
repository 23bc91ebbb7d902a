"""Timing of the collective path with one RCCL rank (1-GPU box): search_batch, search_batch_flat, the all-reduce hook, teardown."""
import os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, torch.distributed as dist
import veloci_amd
from veloci_amd import synth
from veloci_amd.dist import ShardedSearcher
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
torch.cuda.set_device(0)
T = time.perf_counter
t0 = T(); dist.init_process_group("nccl", rank=0, world_size=1); print("init", T() - t0, flush=True)
spec = synth.SynthSpec(num_docs=300_000, num_terms=5000, triples=2, extra_probe_dfs=(1000, 30_000, 300_000), background_terms=40)
data, meta = synth.generate(spec)
idx = veloci_amd.Index(data, device=0)
s = ShardedSearcher(idx, always_collective=True)
t = [list(x) for x in meta.triples]
reqs = [synth.req_and(t[i % 2]) for i in range(576)]
for k in range(3):
    t0 = T(); s.search_batch(reqs[:96]); print("search_batch", T() - t0, flush=True)
for k in range(3):
    t0 = T(); s.search_batch_flat(reqs); print("search_batch_flat", T() - t0, flush=True)
v = np.array([1, 2, 3], dtype=np.uint64)
t0 = T(); s._sum_over_ranks(v); print("allreduce", T() - t0, flush=True)
t0 = T(); dist.destroy_process_group(); print("destroy", T() - t0, flush=True)
# raw cost of a small all-gather on a side stream: host time per call, and time per call including completion
dist.init_process_group("nccl", rank=0, world_size=1)
st = torch.cuda.Stream()
a = torch.zeros(25000, dtype=torch.uint8, device="cuda"); o = torch.empty_like(a)
with torch.cuda.stream(st):
    for _ in range(5): dist.all_gather_into_tensor(o, a)
    torch.cuda.synchronize()
    t0 = T()
    for _ in range(200): dist.all_gather_into_tensor(o, a)
    t1 = T(); torch.cuda.synchronize(); t2 = T()
    print("all_gather host us/call", (t1 - t0) / 200 * 1e6, "incl completion us/call", (t2 - t0) / 200 * 1e6, flush=True)
    t0 = T()
    for _ in range(200):
        dist.all_gather_into_tensor(o, a); st.synchronize()
    print("all_gather + stream sync us/call", (T() - t0) / 200 * 1e6, flush=True)
dist.destroy_process_group()
