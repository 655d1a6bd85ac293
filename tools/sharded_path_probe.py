#!/usr/bin/env python3
"""Where the time of bench.py's multi-GPU step goes at world size 1 (one MI355X, RCCL with one rank): the step graph with and
without the gather bucket, and the host constructs around it (event record, cross-stream waits, the all-gather in or out of
stream order)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29518")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0); device = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=device)
import bitsandbytes_amd as bnb
from bitsandbytes_amd.parallel import ShardedLinear4bit
import bench
LAYERS = 128; M, N, K = 1, 4096, 4096
layers, x = bench.build_layers(device, LAYERS, N, K, M, 64, "nf4", seed=1)
shards = [ShardedLinear4bit(q, st, out_features=N, group=None) for q, st in layers]
buckets = [torch.empty(LAYERS, M, N, device=device, dtype=torch.bfloat16) for _ in range(2)]
gathered = [torch.empty(LAYERS, M, N, device=device, dtype=torch.bfloat16) for _ in range(2)]

def cap(fn):
    return bench.capture(fn)

def t_replays(g, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3 / LAYERS

def f_plain():
    for q, st in layers: bnb.matmul_4bit(x, q, st)
def f_keep():
    outs = [bnb.matmul_4bit(x, q, st) for q, st in layers]
    return outs
def f_shard_keep():
    return [sh.local_forward(x) for sh in shards]
def f_stack(b=0):
    torch.stack([sh.local_forward(x) for sh in shards], out=buckets[b])
keep = []
def wrap(f):
    def g():
        keep.append(f())
    return g
print("us/layer: plain (outputs freed)     ", round(t_replays(cap(f_plain)), 3))
print("us/layer: outputs kept alive         ", round(t_replays(cap(wrap(f_keep))), 3))
print("us/layer: shards, outputs kept alive ", round(t_replays(cap(wrap(f_shard_keep))), 3))
g0 = cap(lambda: f_stack(0)); g1 = cap(lambda: f_stack(1))
print("us/layer: shards + stack into bucket ", round(t_replays(g0), 3))
def timeit(fn, n=40):
    fn(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n / LAYERS * 1e6, 3)
graphs = [g0, g1]
def v_alt(n):
    for c in range(n): graphs[c & 1].replay()
def v_record(n):
    for c in range(n):
        graphs[c & 1].replay(); e = torch.cuda.Event(); e.record()
comm = torch.cuda.Stream()
def v_side_wait(n):
    for c in range(n):
        graphs[c & 1].replay(); e = torch.cuda.Event(); e.record()
        with torch.cuda.stream(comm):
            comm.wait_event(e)
def v_side_wait_back(n):
    pend = [None, None]
    for c in range(n):
        b = c & 1
        if pend[b] is not None: torch.cuda.current_stream().wait_event(pend[b])
        graphs[b].replay(); e = torch.cuda.Event(); e.record()
        with torch.cuda.stream(comm):
            comm.wait_event(e); ev = torch.cuda.Event(); ev.record()
        pend[b] = ev
def v_gather_same_stream(n):
    for c in range(n):
        b = c & 1
        graphs[b].replay()
        dist.all_gather_into_tensor(gathered[b].view(LAYERS * M, N), buckets[b].view(LAYERS * M, N))
for name, fn in (("alternate two graphs", v_alt), ("+ event record", v_record), ("+ side stream waits", v_side_wait),
                 ("+ main stream waits back", v_side_wait_back), ("gather on the same stream", v_gather_same_stream)):
    print(f"us/layer: {name:28s}", timeit(fn))
dist.destroy_process_group()
