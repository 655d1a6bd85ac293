"""CPU, world_size 2 over gloo: the N-sharded Linear4bit (bitsandbytes_amd/parallel.py) — shard
slicing of packed weight / absmax (plain and nested), per-rank matmul, ONE all-gather, and equality
with the unsharded layer. Arithmetic is the oracle (test-only CPU kernels)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle_cpu_backend

    _oracle_cpu_backend.register()
    import bitsandbytes_amd as bnb
    from bitsandbytes_amd.nn import Linear4bit

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for (N, K, dq, qt, bs, bias) in [(64, 256, False, "nf4", 64, True), (64, 512, True, "nf4", 64, False),
                                         (48, 256, True, "fp4", 128, True), (1024, 128, True, "nf4", 64, False)]:
            torch.manual_seed(7)  # same weights on every rank
            layer = Linear4bit(K, N, bias=bias, quant_type=qt, compress_statistics=dq)
            layer.weight.blocksize = bs
            layer = layer.to("cpu")
            sharded = bnb.shard_linear4bit(layer, rank, world)
            assert sharded.weight.numel() == N * K // 2 // world
            for M in (1, 5):
                x = torch.randn(M, K)
                y_full = layer(x)
                y = sharded(x)
                ok &= y.shape == (M, N) and bool(torch.equal(y, y_full))
                # local part is this rank's column block
                y_loc = sharded.local_forward(x)
                ns = N // world
                ok &= bool(torch.equal(y_loc, y_full[:, rank * ns:(rank + 1) * ns]))
        # a group of layers that share x: one buffer, one gather, the members' values
        torch.manual_seed(11)
        K = 256
        layers = [Linear4bit(K, n, bias=b, quant_type="nf4", compress_statistics=dq).to("cpu") for n, b, dq in
                  ((64, True, False), (32, False, False), (96, True, False))]
        group = bnb.ShardedLinear4bitGroup([bnb.shard_linear4bit(layer, rank, world) for layer in layers])
        for M in (1, 3):
            x = torch.randn(M, K)
            ys = group(x)
            for y, layer in zip(ys, layers):
                ok &= y.shape == (M, layer.out_features) and bool(torch.equal(y, layer(x)))
            # leading batch dimensions survive the gather
            x3 = torch.randn(2, M, K)
            for y, layer in zip(group(x3), layers):
                ok &= y.shape == (2, M, layer.out_features) and bool(torch.equal(y, layer(x3)))
        # double-quantised members (the Linear4bit default)
        torch.manual_seed(12)
        layers_dq = [Linear4bit(512, n, bias=b, quant_type="nf4", compress_statistics=True).to("cpu") for n, b in ((64, False), (128, True))]
        group_dq = bnb.ShardedLinear4bitGroup([bnb.shard_linear4bit(layer, rank, world) for layer in layers_dq])
        x = torch.randn(2, 512)
        for y, layer in zip(group_dq(x), layers_dq):
            ok &= bool(torch.equal(y, layer(x)))
        # a chain of layers (each one's gathered y is the next one's x) without a PeerChain - what every rank runs when the fused
        # form is not available (CPU, shapes outside it): the members one by one, the same values; shape validation at build time
        torch.manual_seed(13)
        dims = [(128, 256), (256, 128), (128, 256)]
        chain_layers = [Linear4bit(k, n, bias=(i == 1), quant_type="nf4", compress_statistics=(i == 2)).to("cpu") for i, (k, n) in enumerate(dims)]
        chain = bnb.ShardedLinear4bitChain([bnb.shard_linear4bit(layer, rank, world) for layer in chain_layers], None)
        for M in (1, 3):
            x = torch.randn(M, 128)
            want = x
            for layer in chain_layers:
                want = layer(want)
            ok &= not chain.fused(x) and bool(torch.equal(chain(x), want))
        try:
            bnb.ShardedLinear4bitChain([bnb.shard_linear4bit(chain_layers[0], rank, world), bnb.shard_linear4bit(chain_layers[0], rank, world)], None)
            ok = False
        except ValueError:
            pass
        # one gated FFN block (round 5; BASELINE.json configs[3]) without a PeerChain - the member-by-member path every rank runs off
        # the fused form: grouped gate / up launch + ONE gather, torch's silu * mul, the down shard + one gather. Equal to the
        # unsharded block; and the row-interleaved [gate; up] matrix the fused form would launch over computes, on this rank's
        # rows, exactly the members' outputs (g0, u0, g1, u1, ...), nested statistics carried un-nested.
        for dq, bias in ((False, False), (True, True)):
            torch.manual_seed(14 + dq)
            H, Fd = 128, 512
            gate, up, down = [Linear4bit(k, n, bias=bias, quant_type="nf4", compress_statistics=dq).to("cpu") for k, n in ((H, Fd), (H, Fd), (Fd, H))]
            ffn = bnb.shard_ffn4bit(gate, up, down, rank, world)
            for M in (1, 2):
                x = torch.randn(M, H)
                want = down(torch.nn.functional.silu(gate(x)) * up(x))
                ok &= not ffn.fused(x) and bool(torch.equal(ffn(x), want))
                ns = Fd // world
                stacked = bnb.matmul_4bit(x, ffn.gu_weight, ffn.gu_state, bias=ffn._stacked_bias(x.dtype))
                mine = slice(rank * ns, (rank + 1) * ns)
                ok &= bool(torch.equal(stacked, torch.stack([gate(x)[:, mine], up(x)[:, mine]], dim=-1).reshape(M, -1)))
        try:
            bnb.ShardedFFN4bit(bnb.shard_linear4bit(gate, rank, world), bnb.shard_linear4bit(up, rank, world), bnb.shard_linear4bit(gate, rank, world))
            ok = False  # (down must take what gate / up produce)
        except ValueError:
            pass
        results[rank] = ok
    finally:
        dist.destroy_process_group()


def test_sharded_linear4bit_two_ranks():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
        assert dict(results) == {0: True, 1: True}


def _worker_c4(rank, world, port, results):
    """BASELINE.json configs[3] with its REAL shard geometry: the Llama FFN projections 11008 x 4096 (gate, up) and 4096 x 11008
    (down) N-sharded over `world` ranks - 8 ranks: 1376 / 512 rows per rank, 4 ranks: 2752 / 1024 - i.e. the shapes bench.py --gpus 8
    and a real tensor-parallel decode run, end to end over gloo."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    import _oracle_cpu_backend

    _oracle_cpu_backend.register()
    import bitsandbytes_amd as bnb
    from bitsandbytes_amd.nn import Linear4bit

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        H, Fd = 4096, 11008
        torch.manual_seed(21)  # the same block on every rank
        gate, up, down = [Linear4bit(k, n, bias=b, quant_type="nf4", compress_statistics=True, compute_dtype=torch.bfloat16).to("cpu")
                          for k, n, b in ((H, Fd, False), (H, Fd, False), (Fd, H, True))]
        ffn = bnb.shard_ffn4bit(gate, up, down, rank, world)
        ok &= tuple(ffn.gate.quant_state.shape) == (Fd // world, H) and tuple(ffn.down.quant_state.shape) == (H // world, Fd)
        chain = bnb.ShardedLinear4bitChain([bnb.shard_linear4bit(up, rank, world), bnb.shard_linear4bit(down, rank, world)], None)
        for M in (1, 2):
            x = torch.randn(M, H, generator=torch.Generator().manual_seed(5 + M)).bfloat16()
            want = down(torch.nn.functional.silu(gate(x)) * up(x))
            got = ffn(x)
            ok &= got.shape == (M, H) and bool(torch.equal(got, want))
            ok &= bool(torch.equal(chain(x), down(up(x))))
            # this rank's part of the gathered activation is its row block of the unsharded projections
            ns = Fd // world
            ok &= bool(torch.equal(ffn.up.local_forward(x), up(x)[:, rank * ns:(rank + 1) * ns]))
        results[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_sharded_ffn_block_with_config4_shard_geometry(world):
    """Round-5 review item 8: the sharded FFN block and the sharded chain at world 4 and 8 with BASELINE.json configs[3]'s real
    shapes (1376 / 512 and 2752 / 1024 rows per rank) - what `bench.py --gpus 8` and the peer chain's member-by-member path
    compute - equal the unsharded block bit for bit on every rank (CPU, gloo, the oracle's arithmetic)."""
    port = 29500 + ((os.getpid() + 7 * world) % 2000)
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker_c4, args=(world, port, results), nprocs=world, join=True)
        assert dict(results) == {r: True for r in range(world)}


def test_shard_validation():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle_cpu_backend

    _oracle_cpu_backend.register()
    import bitsandbytes_amd.functional as F
    from bitsandbytes_amd.parallel import shard_quant_state

    W = torch.randn(30, 128)
    q, st = F.quantize_4bit(W, quant_type="nf4")
    with pytest.raises(ValueError, match="divisible"):
        shard_quant_state(q, st, 0, 4)
    # nested state whose shard boundary is not on a second-level block boundary -> un-nested fp32 absmax
    W = torch.randn(64, 128).bfloat16()
    q, st = F.quantize_4bit(W, quant_type="nf4", compress_statistics=True)
    qs, sts = shard_quant_state(q, st, 1, 2)
    assert not sts.nested and sts.absmax.dtype == torch.float32 and sts.shape == (32, 128)
    full = F.dequantize_4bit(q, st)
    assert torch.equal(F.dequantize_4bit(qs, sts), full[32:])


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("N,K,bs", [(11008 // 16, 4096 // 8, 64), (256, 1376, 32), (64, 2048, 128), (8, 4096, 64)])
@pytest.mark.parametrize("dq", [False, True], ids=["fp32-absmax", "nested"])
@pytest.mark.parametrize("quant_storage", [torch.uint8, torch.bfloat16], ids=["u8", "bf16-storage"])
def test_every_shard_reproduces_its_rows(world, N, K, bs, dq, quant_storage):
    """shard_quant_state for 2 / 4 / 8 ranks (BASELINE config 4 shards the FFN matrices over 1-8 GPUs; shapes here
    are those scaled down): every rank's packed shard + state dequantizes to exactly its rows of the unsharded
    weight, the per-rank matmul is exactly its column block, and the rank-major concatenation that the all-gather
    produces is the full output - for plain and nested absmax (both the sliced and the un-nested fallback branch)
    and for non-uint8 quant_storage views."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle_cpu_backend

    _oracle_cpu_backend.register()
    import bitsandbytes_amd as bnb
    import bitsandbytes_amd.functional as F
    from bitsandbytes_amd.parallel import shard_quant_state

    if N % world:
        pytest.skip("rows not divisible by the world size")
    torch.manual_seed(N + K + world)
    W = (torch.randn(N, K) / K**0.5).bfloat16()
    q, st = F.quantize_4bit(W, blocksize=bs, quant_type="nf4", compress_statistics=dq, quant_storage=quant_storage)
    full = F.dequantize_4bit(q, st)
    x = torch.randn(3, K).bfloat16()
    y_full = bnb.matmul_4bit(x, q, st)
    ns = N // world
    parts = []
    for r in range(world):
        qs, sts = shard_quant_state(q, st, r, world)
        assert qs.dtype == torch.uint8 and qs.numel() == ns * K // 2 and tuple(sts.shape) == (ns, K)
        blocks = ns * K // bs
        assert sts.nested == (dq and blocks % 256 == 0)
        assert torch.equal(F.dequantize_4bit(qs, sts), full[r * ns:(r + 1) * ns]), f"rank {r}"
        y_r = bnb.matmul_4bit(x, qs, sts)
        assert torch.equal(y_r, y_full[:, r * ns:(r + 1) * ns]), f"rank {r}"
        parts.append(y_r)
    # what ShardedLinear4bit.gather does with the rank-major all-gather buffer
    buf = torch.stack(parts)  # [G, M, ns]
    assert torch.equal(buf.permute(1, 0, 2).reshape(3, N), y_full)
