#!/usr/bin/env python3
"""What the library GEMM itself reaches on the prefill shapes (torch F.linear = hipBLASLt on bf16 [M, K] x [N, K]^T, no decode at all),
rotating over > 256 MiB of weights: the yardstick for the fused tall batches and for the unfused route above fused_max_m().
    python tools/blas_probe.py            PYTORCH_TUNABLEOP_ENABLED=1 python tools/blas_probe.py"""
import os, sys, time, torch
dev="cuda"
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
print("tunable", os.environ.get("PYTORCH_TUNABLEOP_ENABLED"))
for (N,K) in ((4096,4096),(8192,8192),(11008,4096),(4096,11008)):
    Ws=[torch.randn(N,K,device=dev,dtype=torch.bfloat16) for _ in range(max(2, int(600e6/(N*K*2))))]
    for M in (512,1024,2048,4096):
        x=torch.randn(M,K,device=dev,dtype=torch.bfloat16)
        i=[0]
        def fn():
            i[0]=(i[0]+1)%len(Ws)
            return torch.nn.functional.linear(x,Ws[i[0]])
        us=t(fn)
        print(f"{N}x{K} M={M}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s ({2*M*N*K/us/1e6/2500*100:.1f} %)", flush=True)
