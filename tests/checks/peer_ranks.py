"""Per-rank body of tests/test_gpu_peer.py: WORLD processes that share ONE GPU (cuda:0, rendezvous over gloo) - or, with
PEER_DEVICE_PER_RANK=1, one GPU per rank (backend nccl = RCCL; needs WORLD GPUs) - gather buffers mapped into each other by hipIpc. Exercises bitsandbytes_amd.peer.PeerAllGather, ShardedLinear4bit(peer=), ShardedLinear4bitGroup and
GraphedBlock with the peer kernel inside the graph. Prints "PEER_OK <rank>" on success."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.nn as bnn  # noqa: E402
from bitsandbytes_amd.peer import PeerAllGather, PeerChain  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    per_rank = os.environ.get("PEER_DEVICE_PER_RANK") == "1"
    torch.cuda.set_device(rank if per_rank else 0)
    dev = torch.device("cuda", rank if per_rank else 0)
    if per_rank:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    peer = PeerAllGather(max_bytes=64 * 1024)
    try:
        # ---- the collective itself: several sizes, aligned and not, many rounds (the double buffer turns over)
        for it, (m, ns, dt) in enumerate([(1, 1376, torch.bfloat16), (1, 1376, torch.bfloat16), (4, 1376, torch.float16), (1, 7, torch.bfloat16),
                                          (3, 129, torch.float32), (1, 16384, torch.float32)] * 3):
            y = (torch.arange(m * ns, device=dev, dtype=torch.float32) % 251 + 1000 * rank + it).to(dt).view(m, ns)
            got = peer.all_gather(y)
            want = torch.cat([(torch.arange(m * ns, device=dev, dtype=torch.float32) % 251 + 1000 * r + it).to(dt).view(m, ns)
                              for r in range(world)])
            torch.cuda.synchronize()
            assert torch.equal(got, want), (it, m, ns, dt)
        peer.check()

        # ---- the sharded layer through the peer kernel == the full layer, bit for bit
        torch.manual_seed(3)  # same weights on every rank
        for (N, K, M, dq) in ((1024, 2048, 1, False), (1024, 2048, 3, True), (11008, 4096, 1, False)):
            layer = bnn.Linear4bit(K, N, bias=True, compute_dtype=torch.bfloat16, quant_type="nf4", compress_statistics=dq).to(dev)
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            sharded = bnb.shard_linear4bit(layer, rank, world, peer=peer)
            y = sharded(x)
            torch.cuda.synchronize()
            y_full = layer(x)
            assert y.shape == (M, N) and torch.equal(y, y_full), (N, K, M, dq)

        # ---- a group (Q/K/V-like): one grouped launch + ONE gather == the members one by one
        torch.manual_seed(5)
        K = 2048
        layers = [bnn.Linear4bit(K, n, bias=b, compute_dtype=torch.bfloat16, quant_type="nf4").to(dev) for n, b in ((2048, True), (512, False), (512, True))]
        shards = [bnb.shard_linear4bit(layer, rank, world, peer=peer) for layer in layers]
        group = bnb.ShardedLinear4bitGroup(shards)
        for M in (1, 2, 5):
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            ys = group(x)
            torch.cuda.synchronize()
            for y, layer in zip(ys, layers):
                assert torch.equal(y, layer(x)), M

        # ---- the same group + one more sharded layer as ONE hipGraph per rank, replayed with fresh inputs
        down = bnn.Linear4bit(512 * 1, 1024, bias=False, compute_dtype=torch.bfloat16, quant_type="nf4").to(dev)
        down_s = bnb.shard_linear4bit(down, rank, world, peer=peer)

        def block(x):
            q, k, v = group(x)
            return down_s(k * v) + q[..., :1024]

        x0 = torch.randn(1, K, device=dev, dtype=torch.bfloat16)
        graphed = bnb.GraphedBlock(block, x0)
        for seed in range(4):
            torch.manual_seed(100 + seed)  # same input on every rank
            x = torch.randn(1, K, device=dev, dtype=torch.bfloat16)
            y = graphed(x).clone()
            torch.cuda.synchronize()
            q, k, v = [layer(x) for layer in layers]
            want = down(k * v) + q[..., :1024]
            assert torch.equal(y, want), seed
        peer.check()

        # ---- the peer chain: every all-gather fused into the gemv launches (M = 1) == the layers one by one through the
        # separate gather, bit for bit - an up / down / up stretch, plain and nested statistics, NF4 and FP4, bf16 and fp16,
        # eagerly (the double buffer turns over many times) and as one hipGraph per rank
        chain = PeerChain(max_values=16384)
        try:
            for (H, F, dt, qt, dq, bias) in ((2048, 8192, torch.bfloat16, "nf4", False, True), (4096, 11008 - 11008 % (2 * world * 32), torch.bfloat16, "nf4", True, False),
                                            (1024, 4096, torch.float16, "fp4", True, True)):
                torch.manual_seed(7)  # same weights on every rank
                dims = [(H, F), (F, H), (H, F), (F, H)]
                layers = [bnn.Linear4bit(k, n, bias=bias, compute_dtype=dt, quant_type=qt, compress_statistics=dq).to(dev) for k, n in dims]
                fused = bnb.ShardedLinear4bitChain([bnb.shard_linear4bit(layer, rank, world) for layer in layers], chain)
                plain = [bnb.shard_linear4bit(layer, rank, world, peer=peer) for layer in layers]
                x = torch.randn(1, H, device=dev, dtype=dt)
                assert fused.fused(x), (H, F, dt)
                for it in range(6):
                    torch.manual_seed(200 + it)
                    x = torch.randn(1, H, device=dev, dtype=dt) * 4
                    y = fused(x)
                    torch.cuda.synchronize()
                    want = x
                    for s in plain:
                        want = s(want)
                    assert y.shape == want.shape and torch.equal(y, want), (H, F, dt, qt, dq, it)
                graphed = bnb.GraphedBlock(fused, x, peers=[chain])
                for it in range(4):
                    torch.manual_seed(300 + it)
                    x = torch.randn(1, H, device=dev, dtype=dt) * 4
                    y = graphed(x).clone()
                    torch.cuda.synchronize()
                    want = x
                    for s in plain:
                        want = s(want)
                    assert torch.equal(y, want), ("graph", H, F, dt, it)
                # a batch of two rows is outside the fused form: the chain runs its members one by one, same values
                x2 = torch.randn(2, H, device=dev, dtype=dt)
                want = x2
                for s in plain:
                    want = s(want)
                assert not fused.fused(x2) and torch.equal(fused(x2), want)
            chain.check()
            # which memory the buffers live in follows the topology: cacheable only where every rank sits on ONE device
            assert chain.memory == ("fine" if per_rank and world > 1 else "coarse"), (chain.memory, chain.sharing)
            # serves() IS the launcher's check (geometry included): a first layer whose rows are longer than one workgroup's
            # segment columns (K > 32768: several phases) is outside the form although its shape passes the simple rules - the
            # chain must fall back to its members, not raise "refused a launch its own serves() accepted" 
            assert not chain.serves(64, 32768 + 2048, 64, consume=False) and chain.serves(64, 32768, 64, consume=False)
            assert not chain.serves(64, 16384 + 32, 64, consume=True) and not chain.serves(63, 4096, 64, consume=False)
        finally:
            chain.close()

        # ---- the other memory kind and an odd max_values, on a short stretch: the FINE-grained allocation (what ranks on different
        # devices get) has then run on hardware at least inside one device; max_values % 4 == 2 used to leave the regions 8-byte
        # aligned under 16-byte stores. Construction runs the collective self-test (PeerChain._self_test) each time.
        for memory, max_values in (("fine", 4096), ("coarse", 4098)):
            chain = PeerChain(max_values=max_values, memory=memory)
            try:
                assert chain.memory == memory and chain.max_values % 4 == 0 and chain.max_values >= max_values
                torch.manual_seed(11)
                H, F = 1024, 4096
                layers = [bnn.Linear4bit(k, n, bias=True, compute_dtype=torch.bfloat16, quant_type="nf4").to(dev) for k, n in ((H, F), (F, H), (H, F))]
                fused = bnb.ShardedLinear4bitChain([bnb.shard_linear4bit(layer, rank, world) for layer in layers], chain)
                for it in range(3):
                    torch.manual_seed(400 + it)
                    x = torch.randn(1, H, device=dev, dtype=torch.bfloat16)
                    assert fused.fused(x)
                    y = fused(x)
                    torch.cuda.synchronize()
                    want = x
                    for layer in layers:
                        want = layer(want)
                    assert torch.equal(y, want), (memory, it)
                # a call that wants gradients takes the differentiable member-by-member path, on every rank alike
                xg = torch.randn(1, H, device=dev, dtype=torch.bfloat16, requires_grad=True)
                assert not fused.fused(xg)
                with torch.no_grad():
                    assert fused.fused(xg)  # (nothing would be recorded anyway)
                if world == 1:
                    # (only in a group of one: across ranks the member path crosses the group's all-gather, which is not
                    # differentiable - and gloo's, used by this shared-GPU set-up, refuses to run under grad mode at all)
                    yg = fused(xg)
                    want = xg.detach()
                    for layer in layers:
                        want = layer(want)
                    assert torch.equal(yg.detach(), want)
                    yg.float().sum().backward()
                    assert xg.grad is not None and bool(torch.isfinite(xg.grad).all())
                chain.check()
            finally:
                chain.close()
        # ---- one gated FFN block on the chain (BASELINE.json configs[3]: the Llama FFN matrices sharded over the ranks):
        # [gate; up] (rows interleaved) as ONE produce launch whose epilogue computes silu(gate) * up, the down shard's launch
        # consuming that exchange, the read-out - against the UNSHARDED block down(F.silu(gate(x)) * up(x)) of the three Linear4bit layers, bit for bit.
        import torch.nn.functional as TF

        chain = PeerChain(max_values=32768)
        try:
            for (H, Fd, dt, qt, dq, bias) in ((4096, 11008, torch.bfloat16, "nf4", False, False), (4096, 14336, torch.bfloat16, "nf4", True, False),
                                              (2048, 4096, torch.float16, "fp4", True, True)):
                torch.manual_seed(21)  # same weights on every rank
                gate, up, down = [bnn.Linear4bit(k, n, bias=bias, compute_dtype=dt, quant_type=qt, compress_statistics=dq).to(dev)
                                  for k, n in ((H, Fd), (H, Fd), (Fd, H))]
                ffn = bnb.shard_ffn4bit(gate, up, down, rank, world, chain=chain, peer=peer)

                def block(x):
                    return down(TF.silu(gate(x)) * up(x))

                x = torch.randn(1, H, device=dev, dtype=dt)
                assert ffn.fused(x), (H, Fd, dt)
                for it in range(5):
                    torch.manual_seed(500 + it)
                    x = torch.randn(1, H, device=dev, dtype=dt) * (1 + it)
                    y = ffn(x)
                    torch.cuda.synchronize()
                    assert torch.equal(y, block(x)), ("ffn", H, Fd, dt, qt, dq, it)
                graphed = bnb.GraphedBlock(ffn, x, peers=[chain])
                for it in range(3):
                    torch.manual_seed(600 + it)
                    x = torch.randn(1, H, device=dev, dtype=dt)
                    y = graphed(x).clone()
                    torch.cuda.synchronize()
                    assert torch.equal(y, block(x)), ("ffn graph", H, Fd, dt, it)
                # two rows are outside the fused form: grouped launch, torch's activation, two gathers. (Two, not three: from three rows
                # on the router picks the kernel FAMILY by matrix size - a shard and the full matrix can then run different arithmetic.)
                x2 = torch.randn(2, H, device=dev, dtype=dt)
                # (round 6: two rows on an unsharded projection of >= 3072 rows run the streaming MFMA kernel, its narrower shards the
                # streaming kernel: the matmul tolerance through the block, not the bits - one row, above: bits)
                y2, want2 = ffn(x2), block(x2)
                assert not ffn.fused(x2) and y2.shape == want2.shape
                assert float((y2.float() - want2.float()).norm() / want2.float().norm()) < 2e-2, ("ffn 2 rows", H, Fd, dt)
            chain.check()
            if world == 1:
                # The activation in the producer's epilogue, EXHAUSTIVELY: gate = identity, up = a permutation, down = identity - NF4 holds
                # 1.0 and 0.0 exactly, so gate(x) = x, up(x) = x[perm] and down(a) = a bit for bit, and the block's output IS
                # T(T(silu(g)) * u). Every finite 16-bit pattern is a g once (the upper half of x holds multipliers in [-1, 1]: no
                # overflow, an inf * 0 in the down layer's sum would turn the whole row into NaN); the unsharded block computes the
                # same through torch's own silu and multiply.
                Hh = 4096
                eye = torch.eye(Hh, device=dev)
                perm = torch.cat([2048 + (torch.arange(2048, device=dev) * 7 + 3) % 2048, torch.arange(2048, Hh, device=dev)])
                for dt, ibits in ((torch.bfloat16, torch.int16), (torch.float16, torch.int16)):
                    def make(W):
                        layer = bnn.Linear4bit(Hh, Hh, bias=False, compute_dtype=dt, quant_type="nf4")
                        layer.weight = bnn.Params4bit(W.to(dt), requires_grad=False, quant_type="nf4", module=layer)
                        return layer.to(dev)

                    gate, up, down = make(eye), make(eye[perm]), make(eye)
                    ffn = bnb.shard_ffn4bit(gate, up, down, rank, world, chain=chain)
                    pats = torch.arange(65536, device=dev, dtype=torch.int32).to(torch.int16).view(dt)
                    pats = pats[torch.isfinite(pats)]
                    torch.manual_seed(9)
                    seen = 0
                    for i in range(0, pats.numel(), 2048):
                        gpart = pats[i:i + 2048]
                        x = torch.cat([gpart, gpart.new_zeros(2048 - gpart.numel()), (torch.rand(2048, device=dev) * 2 - 1).to(dt)]).view(1, Hh)
                        assert ffn.fused(x)
                        y = ffn(x)
                        want = down(TF.silu(gate(x)) * up(x))
                        torch.cuda.synchronize()
                        assert bool(torch.isfinite(want).all()) and torch.equal(y, want), ("silu sweep", dt, i, int((y != want).sum()))
                        # (and the block really is the activation: the identity layers add nothing)
                        assert torch.equal(want[0, :gpart.numel()], TF.silu(gpart) * x[0, perm[:gpart.numel()]])
                        seen += gpart.numel()
                    assert seen == pats.numel() >= 63488, seen
                chain.check()
        finally:
            chain.close()
        print(f"PEER_OK {rank}", flush=True)
    finally:
        peer.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
