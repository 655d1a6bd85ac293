#!/usr/bin/env python3
"""Round-6 experiment (review item 3): consecutive DEPENDENT layers without a kernel boundary between them. The data dependency of a
decode chain already travels as tagged granules (peer.PeerChain: the consumer launch fetches its x from the exchange buffer and
re-fetches until the tags are there), so stream order between layer i and layer i + 1 is redundant. Round 4 put the layers on two
alternating streams and dead-locked: each launch could fill the device, and nothing orders the dispatch of two queues. Here every
launch takes at most HALF the workgroup slots (wg_limit = 128 of 256 CUs): at most one launch per stream is resident (stream order),
so whatever the dispatch order the producer of the exchange a resident launch waits for can always be placed. Every wait is bounded
(BNB_MI355X_PEER_WAIT_POLLS): a protocol error shows as a status word / NaN, not as a hang.
    python tools/chain_overlap.py            (one process, a gloo group of one; 128 layers of 4096 x 4096 NF4 bf16)
Forms timed (hipGraph replay, us per layer): plain matmul_4bit launches | chain on ONE stream, 256 workgroups | chain on one stream,
128 workgroups | chain on TWO alternating streams, 128 workgroups. The chain forms must give the same bits."""
import os
import socket
import statistics
import sys

os.environ.setdefault("BNB_MI355X_PEER_WAIT_POLLS", "300000")  # ~0.3 s per fetch: a dead-lock ends quickly, as an error
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.peer import PeerChain  # noqa: E402

N = K = 4096
L = 128
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
torch.cuda.set_device(0)
print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode())
g = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(L):
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, quant_type="nf4"))
    del W
x = torch.randn(K, device="cuda", generator=g).bfloat16()
chain = PeerChain(max_values=K)
out = torch.empty(N, dtype=torch.bfloat16, device="cuda")
side = [torch.cuda.Stream(), torch.cuda.Stream()]


def plain():
    y = x.view(1, -1)
    for q, st in layers:
        y = bnb.matmul_4bit(y, q, st)
    out.copy_(y.view(-1))


def chained(wg_limit, two_streams):
    chain.wg_limit = wg_limit
    cur = torch.cuda.current_stream()
    if two_streams:
        for s_ in side:
            s_.wait_stream(cur)
    for i, (q, st) in enumerate(layers):
        if two_streams:
            with torch.cuda.stream(side[i & 1]):
                ok = chain.gemv(x if i == 0 else None, q, st, consume=i > 0, produce=True, dtype=torch.bfloat16)
        else:
            ok = chain.gemv(x if i == 0 else None, q, st, consume=i > 0, produce=True, dtype=torch.bfloat16)
        assert ok, i
    if two_streams:
        for s_ in side:
            cur.wait_stream(s_)
    chain.read(N, torch.bfloat16, out=out)


def capture(fn):
    fn()
    torch.cuda.synchronize()
    chain.check()
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        fn()
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    gr.replay()
    torch.cuda.synchronize()
    chain.check()
    return gr


def timed(gr, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * L) * 1e3


forms = [("plain launches", plain), ("chain, 1 stream, 256 wg", lambda: chained(0, False)), ("chain, 1 stream, 128 wg", lambda: chained(128, False)),
         ("chain, 2 streams, 128 wg", lambda: chained(128, True)), ("chain, 2 streams, 96 wg", lambda: chained(96, True))]
graphs, results = [], []
for name, fn in forms:
    gr = capture(fn)
    graphs.append(gr)
    results.append(out.clone())
    print(f"captured: {name}; status word {chain.status()}", flush=True)
same = [bool(torch.equal(results[1], r)) for r in results[1:]]
print("chain forms bit-identical to the one-stream chain:", same, "| finite:", bool(torch.isfinite(results[1].float()).all()))
print("plain launches vs chain (the plain form rounds every layer's y to bf16 too - same values expected):", bool(torch.equal(results[0], results[1])))
samples = [[] for _ in forms]
for r in range(5):
    order = list(range(len(forms)))
    if r % 2:
        order.reverse()
    for i in order:
        samples[i].append(timed(graphs[i], 40))
chain.check()
for (name, _), smp in zip(forms, samples):
    print(f"{name:28s} {statistics.median(smp):6.2f} us per layer   (min {min(smp):.2f}, max {max(smp):.2f})")
# the replays must still give the captured result
graphs[3].replay()
torch.cuda.synchronize()
print("two-stream replay reproduces the one-stream chain:", bool(torch.equal(out, results[1])), "| status word", chain.status())
chain.close()
dist.destroy_process_group()
