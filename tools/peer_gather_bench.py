#!/usr/bin/env python3
"""Latency of the one-shot peer all-gather (bitsandbytes_amd/peer.py) per collective: WORLD processes sharing cuda:0 (what a 1-GPU
box can run: same kernel, flags and ordering as across xGMI, without the link), hipGraph of 200 dependent collectives, next to
torch.distributed's all_gather_into_tensor on an RCCL group of ONE rank (the protocol's fixed cost on this stack).
    python tools/peer_gather_bench.py [world ...]          (spawns its own ranks)
    python tools/peer_gather_bench.py chain [world ...]    the FUSED form (peer.PeerChain: the gather inside the gemv launches) against
                                                           kernel + separate gather and against the kernels alone, per layer of an
                                                           up / down chain whose per-rank shard is 4096^2 weights at every world size
    python tools/peer_gather_bench.py ffn [world ...]      one gated FFN block per token on the chain (parallel.ShardedFFN4bit) at the
                                                           per-rank work of an 8-way sharded Llama FFN, against the member-by-member form"""
import os
import socket
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def graph_us(fn, n=200, reps=5):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3


def rank_main():
    import torch.distributed as dist

    from bitsandbytes_amd.peer import PeerAllGather

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    peer = PeerAllGather(max_bytes=64 * 1024)
    try:
        for i, nbytes in enumerate((2752, 2752, 8192, 65536)):
            y = torch.zeros(1, nbytes // 2, device="cuda", dtype=torch.bfloat16)
            out = torch.empty(world, nbytes // 2, device="cuda", dtype=torch.bfloat16)
            dist.barrier()
            t = graph_us(lambda: peer.all_gather(y, out))
            peer.check()
            if rank == 0 and i > 0:  # (the first pass pays the processes' start-up skew)
                print(f"peer all-gather, {world} process(es) on one GPU, {nbytes:6d} B per rank: {t:6.2f} us per collective", flush=True)
    finally:
        peer.close()
        dist.destroy_process_group()


def chain_main():
    """Per layer of a decode chain up (K = 4096 -> world x 4096 features, 4096 per rank) / down (K = world x 4096 -> 4096 features,
    4096 / world per rank): every shard is 16.8 M weights = the headline layer's bytes, whatever the world size."""
    import torch.distributed as dist

    import bitsandbytes_amd as bnb
    import bitsandbytes_amd.nn as bnn
    from bitsandbytes_amd.peer import PeerAllGather, PeerChain

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, Fd = 4096, 4096 * world
    pairs = 12
    peer = PeerAllGather(max_bytes=64 * 1024)
    chain = PeerChain(max_values=Fd)
    try:
        torch.manual_seed(11)
        shards = []
        for i in range(pairs):
            for (k, n) in ((H, Fd), (Fd, H)):
                layer = bnn.Linear4bit(k, n, bias=False, compute_dtype=torch.bfloat16, quant_type="nf4", compress_statistics=False).to(dev)
                shards.append((bnb.shard_linear4bit(layer, rank, world), bnb.shard_linear4bit(layer, rank, world, peer=peer)))
                del layer
        L = len(shards)
        x = torch.randn(1, H, device=dev, dtype=torch.bfloat16)
        xs = {H: x, Fd: torch.randn(1, Fd, device=dev, dtype=torch.bfloat16)}
        fused = bnb.ShardedLinear4bitChain([a for a, _ in shards], chain)
        assert fused.fused(x)

        def run_fused():
            fused(x)

        def run_separate():
            y = x
            for _, b in shards:
                y = b(y)

        def run_alone():  # the shard kernels only, each on a resident input of its K (no exchange at all)
            for a, _ in shards:
                a.local_forward(xs[int(a.quant_state.shape[1])])

        y1 = fused(x)
        y2 = x
        for _, b in shards:
            y2 = b(y2)
        torch.cuda.synchronize()
        same = bool(torch.equal(y1, y2))
        res = {}
        variants = [("kernels alone", run_alone), ("kernel + separate peer gather", run_separate), ("fused chain", run_fused)]
        if world == 1:
            # what each half of the protocol costs (one process: every layer is 4096 x 4096, any exchange fits any layer)
            out_l = torch.empty(H, device=dev, dtype=torch.bfloat16)

            def run_produce_only():
                for a, _ in shards:
                    chain.gemv(x.view(-1), a.weight, a.quant_state, consume=False, produce=True)
                chain.read(H, torch.bfloat16)

            def run_consume_only():
                a0 = shards[0][0]
                chain.gemv(x.view(-1), a0.weight, a0.quant_state, consume=False, produce=True)
                chain.gemv(x.view(-1), a0.weight, a0.quant_state, consume=False, produce=True)   # (a chain holds >= 2 exchanges)
                for a, _ in shards:
                    chain.gemv(None, a.weight, a.quant_state, out_local=out_l, consume=True, produce=False, dtype=torch.bfloat16)
                chain.read(H, torch.bfloat16)

            variants += [("produce only", run_produce_only), ("consume only (+2 launches)", run_consume_only)]
        for name, fn in variants:
            dist.barrier()
            res[name] = graph_us(fn, n=1, reps=20) / L
        chain.check()
        peer.check()
        if rank == 0:
            print(f"{world} process(es) on one GPU, {L} layers (up {H} -> {Fd}, down {Fd} -> {H}; 16.8 M weights per rank and layer"
                  f"{', workgroups capped at ' + str(chain.wg_limit) + ' per rank for the fused form' if chain.wg_limit else ''}), us per layer: "
                  + " | ".join(f"{k} {v:6.2f}" for k, v in res.items()) + f" | fused == separate bit for bit: {same}", flush=True)
    finally:
        chain.close()
        peer.close()
        dist.destroy_process_group()


def ffn_main():
    """One gated FFN block per token (BASELINE.json configs[3]). The per-RANK work of an 8-way shard of a Llama FFN is a [2 x F/8, H]
    gate / up launch and an [H/8, F] down launch whatever the number of ranks present: with WORLD processes on this one GPU the block
    is built at F_total = WORLD x F/8, H_out = WORLD x H/8 so that every rank does exactly that work (weak scaling, as bench.py's chain).
    us per block: the fused form (two launches + read-out on the peer chain), the member-by-member form (grouped gate / up launch,
    one gather, torch's silu * mul, the down shard, one gather), and - world 1 only - the three UNSHARDED matrices of the 8B model."""
    import torch.distributed as dist

    import bitsandbytes_amd as bnb
    import bitsandbytes_amd.nn as bnn
    from bitsandbytes_amd.peer import PeerAllGather, PeerChain

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    peer = PeerAllGather(max_bytes=64 * 1024)
    chain = PeerChain(max_values=32768)
    try:
        for name, H, F8, H8 in (("Llama-3-8B FFN / 8 (H 4096, F 14336)", 4096, 14336 // 8, 512), ("Llama-2-7B FFN / 8 (H 4096, F 11008)", 4096, 11008 // 8, 512)):
            Fd, Hout = world * F8, world * H8
            if Fd % 64:
                continue  # (1 x 1376: the down projection's K must be whole quantization blocks)
            torch.manual_seed(3)
            gate, up = [bnn.Linear4bit(H, Fd, bias=False, compute_dtype=torch.bfloat16, quant_type="nf4").to(dev) for _ in range(2)]
            down = bnn.Linear4bit(Fd, Hout, bias=False, compute_dtype=torch.bfloat16, quant_type="nf4").to(dev)
            fused = bnb.shard_ffn4bit(gate, up, down, rank, world, chain=chain)
            plain = bnb.shard_ffn4bit(gate, up, down, rank, world, chain=None, peer=peer)
            x = torch.randn(1, H, device=dev, dtype=torch.bfloat16)
            assert fused.fused(x) and torch.equal(fused(x), plain(x))
            dist.barrier()
            with torch.no_grad():
                t_f = min(graph_us(lambda: fused(x), n=100) for _ in range(3))
                dist.barrier()
                t_p = min(graph_us(lambda: plain(x), n=100) for _ in range(3))
            chain.check()
            line = f"{name}, {world} process(es) on one GPU: fused chain {t_f:6.2f} us per block | member by member {t_p:6.2f}"
            if world == 1 and rank == 0:
                full = [bnn.Linear4bit(k, n, bias=False, compute_dtype=torch.bfloat16, quant_type="nf4").to(dev) for k, n in ((H, 8 * F8), (H, 8 * F8), (8 * F8, H))]
                with torch.no_grad():
                    t_u = min(graph_us(lambda: full[2](torch.nn.functional.silu(full[0](x)) * full[1](x)), n=50) for _ in range(3))
                line += f" | the UNSHARDED 8B block on this one GPU {t_u:6.2f}"
            if rank == 0:
                print(line, flush=True)
    finally:
        chain.close()
        peer.close()
        dist.destroy_process_group()


def rccl_world_one():
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        for nbytes in (2752, 65536):
            y = torch.zeros(1, nbytes // 2, device="cuda", dtype=torch.bfloat16)
            out = torch.empty(1, nbytes // 2, device="cuda", dtype=torch.bfloat16)
            try:
                t = graph_us(lambda: dist.all_gather_into_tensor(out, y))
                how = "hipGraph"
            except Exception:  # capture of the collective refused: eager enqueue
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(1000):
                    dist.all_gather_into_tensor(out, y)
                e1.record()
                torch.cuda.synchronize()
                t, how = e0.elapsed_time(e1), "eager"
            print(f"RCCL all_gather_into_tensor, group of ONE rank, {nbytes:6d} B: {t:6.2f} us per collective ({how})", flush=True)
    finally:
        dist.destroy_process_group()


def main():
    if os.environ.get("PEER_BENCH_RANK") == "1":
        return rank_main()
    if os.environ.get("PEER_BENCH_RANK") == "rccl":
        return rccl_world_one()
    if os.environ.get("PEER_BENCH_RANK") == "chain":
        return chain_main()
    if os.environ.get("PEER_BENCH_RANK") == "ffn":
        return ffn_main()
    mode = "1"
    if len(sys.argv) > 1 and sys.argv[1] in ("chain", "ffn"):
        mode = sys.argv[1]
        del sys.argv[1]
    # (more than ~4 processes on ONE device are time-sliced by the driver - 10 ms per collective at 8: not a property of the kernel)
    worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
    for world in worlds:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)],
                                  env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world),
                                           PEER_BENCH_RANK=mode, HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(world)]
        for p in procs:
            p.wait(timeout=300)
    if mode in ("chain", "ffn"):
        return
    subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, PEER_BENCH_RANK="rccl"), timeout=300)


if __name__ == "__main__":
    main()
