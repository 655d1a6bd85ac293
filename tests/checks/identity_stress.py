#!/usr/bin/env python3
"""Stress of the path's bit-for-bit identities on one GPU, many random inputs each (a claim checked on five inputs can hold by luck: the
one-fma nested scale did for two rounds - DESIGN 6a). Single process; shards are cut with shard_linear4bit(layer, r, w) and their LOCAL
outputs compared with the slices of the unsharded layer's output.
    python tests/checks/identity_stress.py [--iters 60]
 1. every row shard (world 2, 4, 8) of a layer == the same rows of the full layer, M = 1 and 2, plain and double-quantised statistics,
    shard boundaries on and off second-level block boundaries;
 2. grouped launch (Q/K/V-like) == the members one by one;
 3. the sharded FFN block (world 1: both launches + read-out on the peer chain) == down(silu(gate(x)) * up(x));
 4. nested statistics == the reconstructed fp32 absmax in the routed kernel at M = 1 ... 64;
 5. dequantize_4bit(nested) == the three-operator sequence.
Prints one line per claim: comparisons made, mismatches (must be 0)."""
import argparse
import os
import sys

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
import bitsandbytes_amd.nn as bnn  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402

DEV = "cuda"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=60)
    a = ap.parse_args()
    bad_total = 0

    def report(name, n, bad):
        nonlocal bad_total
        bad_total += bad
        print(f"{name:78s} {n:6d} comparisons, {bad} mismatches" + ("   <-- FAIL" if bad else ""), flush=True)

    # 1. row shards
    n = bad = same_family = 0
    for (N, K, dt, qt, dq) in ((14336, 4096, torch.bfloat16, "nf4", True), (11008, 4096, torch.bfloat16, "nf4", False),
                               (4096, 11008, torch.float16, "fp4", True), (1000, 2048, torch.bfloat16, "nf4", True)):
        torch.manual_seed(N)
        layer = bnn.Linear4bit(K, N, bias=True, compute_dtype=dt, quant_type=qt, compress_statistics=dq).to(DEV)
        for world in (2, 4, 8):
            if N % world:
                continue
            shards = [bnb.shard_linear4bit(layer, r, world, gather_output=False) for r in range(world)]
            for it in range(max(4, a.iters // 6)):
                for M in (1, 2, 4, 8):
                    x = (torch.randn(M, K, device=DEV) * (1 + it % 3)).to(dt)
                    y = layer(x)
                    fam_full = bnb.lib.bnb_mi355x_last_gemm_kernel()
                    for r, sh in enumerate(shards):
                        ns = N // world
                        n += 1
                        ys = sh.local_forward(x)
                        if bnb.lib.bnb_mi355x_last_gemm_kernel() == fam_full and fam_full in (1, 7):
                            # (the streaming kernel and - round 6 - the streaming MFMA kernel sum a row in an order that does not
                            # depend on the launch geometry)
                            same_family += 1
                            bad += 0 if torch.equal(ys, y[:, r * ns:(r + 1) * ns]) else 1
                        else:
                            # (the shard and the layer ran different kernel families - e.g. two rows on long rows of a narrow shard: the
                            # streaming kernel - another summation order and 16-bit code values: the matmul tolerance, not the bits)
                            assert M >= 2, (M, fam_full)
                            ref = y[:, r * ns:(r + 1) * ns].float()
                            bad += 0 if float((ys.detach().float() - ref).norm() / ref.norm()) < 1e-2 else 1
    report(f"row shards (world 2/4/8) == rows of the unsharded layer, M = 1, 2, 4, 8: bits ({same_family} of them: same kernel family) or 1e-2", n, bad)

    # 2. grouped launch
    n = bad = 0
    torch.manual_seed(5)
    K = 4096
    for dq in (False, True):
        layers = [bnn.Linear4bit(K, nn_, bias=b, compute_dtype=torch.bfloat16, quant_type="nf4", compress_statistics=dq).to(DEV)
                  for nn_, b in ((4096, True), (1024, False), (1024, True))]
        group = bnb.ShardedLinear4bitGroup([bnb.shard_linear4bit(layer, 0, 1) for layer in layers])
        for it in range(a.iters):
            for M in (1, 2, 4, 8, 16):
                x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
                for y, layer in zip(group(x), layers):
                    n += 1
                    bad += 0 if torch.equal(y, layer(x)) else 1
    report("grouped launch == the members one by one, M = 1, 2, 4, 8, 16", n, bad)

    # 3. the FFN block on the peer chain, world 1
    import torch.distributed as dist
    from bitsandbytes_amd.peer import PeerChain

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        import socket

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
    dist.init_process_group("gloo", rank=0, world_size=1)
    n = bad = 0
    chain = PeerChain(max_values=32768)
    try:
        for (H, Fd, dt, qt, dq, bias) in ((4096, 14336, torch.bfloat16, "nf4", True, False), (4096, 11008, torch.bfloat16, "nf4", False, False),
                                          (2048, 4096, torch.float16, "fp4", True, True)):
            torch.manual_seed(21)
            gate, up, down = [bnn.Linear4bit(k, n_, bias=bias, compute_dtype=dt, quant_type=qt, compress_statistics=dq).to(DEV)
                              for k, n_ in ((H, Fd), (H, Fd), (Fd, H))]
            ffn = bnb.shard_ffn4bit(gate, up, down, 0, 1, chain=chain)
            for it in range(a.iters):
                x = torch.randn(1, H, device=DEV, dtype=dt) * (1 + it % 4)
                assert ffn.fused(x)
                y = ffn(x)
                torch.cuda.synchronize()
                n += 1
                bad += 0 if torch.equal(y, down(TF.silu(gate(x)) * up(x))) else 1
        chain.check()
    finally:
        chain.close()
        dist.destroy_process_group()
    report("sharded FFN block on the peer chain (world 1) == the unsharded block", n, bad)

    # 4. nested statistics == reconstructed fp32 absmax, routed kernel
    n = bad = 0
    for (N, K, dt, qt, bs) in ((4096, 4096, torch.bfloat16, "nf4", 64), (11008, 4096, torch.float16, "fp4", 128), (8192, 8192, torch.bfloat16, "nf4", 64)):
        torch.manual_seed(N + bs)
        W = (torch.randn(N, K, device=DEV) / K**0.5).to(dt)
        q, st = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=True)
        am = torch.ops.bitsandbytes.dequantize_blockwise.default(st.absmax, st.state2.absmax, st.state2.code, 256, torch.float32) + st.offset
        for it in range(max(4, a.iters // 6)):
            for M in (1, 2, 4, 8, 16, 32, 64):
                x = torch.randn(M, K, device=DEV).to(dt)
                ya = hip._gemm_4bit_fused(x, q, st.shape, st.state2.absmax, bs, qt, None, st.absmax, st.state2.code, st.offset)
                fa = bnb.lib.bnb_mi355x_last_gemm_kernel()
                yb = hip._gemm_4bit_fused(x, q, st.shape, am, bs, qt, None, None, None, None)
                if fa != bnb.lib.bnb_mi355x_last_gemm_kernel():
                    continue  # (another kernel family for the other kind of statistics: another summation order)
                n += 1
                bad += 0 if torch.equal(ya, yb) else 1
    report("nested statistics == reconstructed fp32 absmax (routed kernel, M = 1 ... 64)", n, bad)

    # 5. dequantize
    n = bad = 0
    for (shape, dt, qt, bs) in (((4096, 4096), torch.float16, "nf4", 64), ((11008, 4096), torch.bfloat16, "fp4", 128), ((1000, 96), torch.float32, "nf4", 32)):
        for it in range(max(4, a.iters // 6)):
            W = (torch.randn(*shape, device=DEV) * 0.05 * (1 + it)).to(dt)
            q, st = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=True)
            am = torch.ops.bitsandbytes.dequantize_blockwise.default(st.absmax, st.state2.absmax, st.state2.code, 256, torch.float32) + st.offset
            want = torch.ops.bitsandbytes.dequantize_4bit.default(q, am, bs, qt, list(shape), dt)
            n += 1
            bad += 0 if torch.equal(F.dequantize_4bit(q, st), want) else 1
    report("dequantize_4bit(nested state) == the three-operator sequence", n, bad)
    print("IDENTITY_STRESS " + ("OK" if bad_total == 0 else f"FAILED ({bad_total})"))
    sys.exit(0 if bad_total == 0 else 1)


if __name__ == "__main__":
    main()
