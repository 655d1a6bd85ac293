#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X 4-bit path (contract in the task statement).

Workload at N = 1 (BASELINE.json configs[1], the configuration the headline metric is quoted on):
    gemv_4bit / Linear4bit forward, NF4, bf16, M = 1, N = K = 4096, blocksize 64, fp32 absmax.
One "step" = one pass of ONE activation row through a stack of LAYERS = 128 distinct such layers (a decode step
through 128 Linear4bit layers; 128 x 9.45 MB = 1.21 GB, far beyond the 256 MiB Infinity Cache, so every launch
streams its weights from HBM). Inputs are resident in HBM before the timed region. Every launch is its own kernel
in a dependent stream - nothing is grouped or overlapped across layers in the headline number.

    value = algorithmic bytes per step x steps / wall time            [GB/s, whole job, all ranks]
    algorithmic bytes per layer = N*K/2 + 4*N*K/bs + 2*M*K + 2*M*N = 9 453 568   

Timed region: each step is one replay of a hipGraph holding the step's 128 launches through the public op
(bitsandbytes_amd.matmul_4bit) - for ANY --steps / --warmup value, so a short driver run measures the same thing as
a long one. Bracketed by barrier + synchronize; MAX over ranks.

--gpus N > 1 (weak scaling, bitsandbytes_amd.parallel semantics: rows sharded, x replicated, no reduction): a decode step
through an N-sharded MLP stack - up (H -> F) / down (F -> H) layers, each sharded by output features, each layer's
gathered y the next layer's x; H x F = N x 4096^2, so every rank's shard of every layer holds 4096^2 weights, the
headline layer's bytes, at every N (1: 4096 / 4096, 2: 4096 / 8192, 4: 8192 / 8192, 8: 8192 / 16384). A step is what a
tensor-parallel decode does: 128 x (shard kernel + the all-gather of THAT layer's outputs), one hipGraph per step.
`value` takes the fused form (bitsandbytes_amd.peer.PeerChain: the gather inside the gemv launches, one launch per
layer) when it reproduces the separate form bit for bit at start-up on every rank, else kernel + a collective per layer
(the one-shot peer kernel when IT reproduces RCCL, else RCCL's all_gather_into_tensor); the other forms and the shard
kernels alone are timed beside it ("sharded_chain"). N > 1 was never run on real links by the builder.

Extra objects on the JSON line:
  "roofline"      dominant kernel; achieved = algorithmic bytes / kernel_us, frac = achieved / 8 TB/s. FOUR clocks are
                  taken and all stay on the line:
                    frac         (the figure of record) kernel_us = HIP events - recorded on the stream the kernels are launched
                                 on - around the timed region / launches: the product binary, kernel + launch boundary;
                    frac_wall    the same region on the host's wall clock (perf_counter around synchronize: what `value` and
                                 ms_per_step use and the driver can check from outside);
                    frac_span    the kernel's own span, first wavefront in to last wavefront out, from in-kernel
                                 s_memrealtime stamps over hipGraph replays of the same step (child process, MEASUREMENT
                                 build of the library - a different binary, hence not the figure of record): the only
                                 per-kernel clock whose 128-fold sum fits inside the driver-timed step;
                    frac_rocprof average duration from `rocprofv3 --kernel-trace --stats` over the same workload (the
                                 profiler serialises dispatches: 128 x this figure exceeds the step - kept as the
                                 cross-check the rules name; the CSV is copied to --profile-out when given);
                  "method" says which clock each key uses; the HIP-event launch-to-launch time of a separate 10-replay region is
                  kept as a secondary key; "traffic" from two rocprofv3 PMC passes.
  "cpu_baseline"  (N = 1, rank 0) the reference's own AVX512-BF16 fused CPU gemv from oracle/_ref when the host
                  supports it, else the scalar port from oracle/.
  "headline_sweep_N4096_K4096"  M = 1..64 (the other half of BASELINE.json's metric; --no-sweep skips it): every entry carries the
                  kernel family that ran, GB/s, TFLOP/s and both roofline fractions.
  "config3"       BASELINE.json configs[2] (gemm_4bit M = 64, N = K = 8192) through the public op: kernel, us, frac_hbm, frac_mfma and the
                  HBM traffic per call from the PMC counters (--no-config3 skips it).
  "grouped"       the same 128 layers launched as 32 groups of 4 through matmul_4bit_grouped (one launch per group):
                  what the boundary costs, reported beside the headline, never instead of it; "sweep": the same at M = 1 ... 32.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
LAYERS = 128
KERNEL_SUBSTR = "gemv4_stream_kernel"  # the headline (M = 1) kernel; other --m: dominant_kernel() asks the library which family ran
# bnb_mi355x_last_gemm_kernel() (csrc/bnb_common.h, GemmKernelId) -> the kernel's name as the profiler prints it
KERNEL_FAMILIES = {1: "gemv4_stream_kernel", 2: "gemv4_generic_kernel", 3: "gemm4_mfma_rt_kernel", 4: "gemm4_mfma_pc_kernel", 6: "gemm4_mfma_kq_kernel",
                   7: "gemm4_mfma_sm_kernel"}
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak (MI355X_MICROARCH.md)


def dominant_kernel(M, N, K, bs, qt):
    """(profiler substring, label) of the kernel `matmul_4bit` launches for this shape - asked of the library after one real
    call (thread-local record of the family that ran), not assumed from M."""
    import bitsandbytes_amd as bnb
    import bitsandbytes_amd.functional as F

    W = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16)
    q, st = F.quantize_4bit(W, blocksize=bs, quant_type=qt)
    bnb.matmul_4bit(torch.zeros(M, K, device="cuda", dtype=torch.bfloat16), q, st)
    torch.cuda.synchronize()
    fam = KERNEL_FAMILIES.get(int(bnb.lib.bnb_mi355x_last_gemm_kernel()), KERNEL_SUBSTR)
    rows = f"{M} row" + ("" if M == 1 else "s")
    return fam, f"{fam}<bf16, {rows}>" + (" (+ its split-K finalize launch where the plan has K slices)" if fam in ("gemm4_mfma_pc_kernel", "gemm4_mfma_kq_kernel") else "")


def algorithmic_bytes(M, N, K, bs, elt=2):
    return N * K // 2 + 4 * (N * K // bs) + elt * M * K + elt * M * N


def sharded_chain_dims(world, N, K):
    """(H, F, [(rows of a rank's shard, K) of an up layer, ... of a down layer]) of the N-sharded MLP chain bench.py --gpus
    `world` runs: up H -> F, down F -> H, H x F = world x N x K, so every rank's shard of every layer holds N x K weights at
    every world size (weak scaling). None when `world` does not divide the model evenly (not a power of two)."""
    lg = max(0, world.bit_length() - 1)
    H = N << (lg // 2)
    Fd = (N * K * world) // H
    if (H * Fd != N * K * world) or Fd % world or H % world:
        return None
    return H, Fd, [(Fd // world, H), (H // world, Fd)]


def build_layers(device, n_layers, N, K, M, blocksize, quant_type, seed):
    import bitsandbytes_amd.functional as F

    g = torch.Generator(device=device).manual_seed(seed)
    layers = []
    for _ in range(n_layers):
        W = (torch.randn(N, K, device=device, generator=g) / K**0.5).to(torch.bfloat16)
        q, st = F.quantize_4bit(W, blocksize=blocksize, quant_type=quant_type)
        layers.append((q, st))
        del W
    x = torch.randn(M, K, device=device, generator=g).to(torch.bfloat16)
    return layers, x


def capture(fn):
    """Warm `fn` on a side stream, capture it into a hipGraph, replay once."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    return g


def cpu_baseline(M, N, K, blocksize, quant_type):
    """Reference CPU path timed on this host (rank 0, N = 1 only). Checker infrastructure from oracle/."""
    from oracle import oracle as O

    torch.manual_seed(0)
    W = (torch.randn(N, K) / K**0.5).bfloat16()
    x = torch.randn(M, K).bfloat16()
    q, am = O.quantize_4bit(W, blocksize, quant_type)
    nbytes = algorithmic_bytes(M, N, K, blocksize)
    if O.ref_has_avx512bf16():
        wp, amt = O.ref_pack_for_cpu_gemv(q, am, N, K, blocksize)
        fn = lambda: O.ref_fused_gemv(x, wp, amt, N, K, blocksize, quant_type)  # noqa: E731
        kind, cores = "reference", len(os.sched_getaffinity(0))
        what = "gemv_4bit_inference_cpu_nf4_bf16 (csrc/cpu_ops.cpp:865-915, OpenMP over all host cores)"
    else:
        fn = lambda: O.gemv_4bit_f32acc(x, q, (N, K), am, blocksize, quant_type)  # noqa: E731
        kind, cores = "port", 1
        what = "oracle scalar dequant+dot (host lacks AVX512-BF16 for the reference's fused kernel)"
    for _ in range(3):
        fn()
    iters, t0 = 0, time.perf_counter()
    while True:
        fn()
        iters += 1
        dt = time.perf_counter() - t0
        if (iters >= 20 and dt > 10.0) or dt > 30.0:  # a bounded ~10 s sample
            break
    out = {
        "value": round(nbytes * iters / dt / 1e9, 3),
        "unit": "GB/s",
        "cores": cores,
        "kind": kind,
        "ms_per_layer": round(dt / iters * 1e3, 4),
        "sample": f"{iters} forward passes of one M={M} N=K={N} layer (the unit the GPU step repeats {LAYERS}x) in {dt:.1f} s; {what}",
    }

    # the other two CPU legs SURVEY section 8(d) lists beside the fused gemv, each a bounded ~2 s sample
    def timed(fn, budget=2.0):
        fn()
        n, t0 = 0, time.perf_counter()
        while n < 3 or time.perf_counter() - t0 < budget:
            fn()
            n += 1
        return (time.perf_counter() - t0) / n * 1e3, n

    try:
        if O.ref_lib_usable():
            deq = lambda: torch.nn.functional.linear(  # noqa: E731
                x, O.ref_dequantize_4bit(q, am, blocksize, quant_type, (N, K), torch.bfloat16))
            deq_kind = "reference dequantizeBlockwise4bitCpu (csrc/cpu_ops.cpp:304-434) + torch F.linear"
        else:
            deq = lambda: torch.nn.functional.linear(  # noqa: E731
                x, O.dequantize_4bit(q, am, blocksize, quant_type, (N, K), torch.bfloat16))
            deq_kind = "oracle dequantize_4bit + torch F.linear"
        ms, n = timed(deq)
        out["unfused_dequantize_linear"] = {"ms_per_layer": round(ms, 3), "iters": n, "what": deq_kind,
                                            "torch_threads": torch.get_num_threads()}
        ms, n = timed(lambda: O.quantize_4bit(W, blocksize, quant_type))
        out["quantize_4bit"] = {"ms_per_call": round(ms, 3), "iters": n, "cores": 1,
                                "what": f"oracle scalar port of backends/default/ops.py:225-259 on a {N}x{K} bf16 weight"}
    except Exception as exc:  # informational
        out["other_legs_error"] = f"{type(exc).__name__}: {exc}"[:200]
    return out


def _under_profiler():
    return any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ) or \
        "rocprofiler" in os.environ.get("LD_PRELOAD", "")


def _child_cmd(extra_args):
    return [sys.executable, os.path.abspath(__file__), "--prof-child", "--no-cpu-baseline"] + list(extra_args)


def _child_env():
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def rocprof_kernel_stats(extra_args, profile_out=None, timeout_s=180):
    """Average duration of the dominant kernel from `rocprofv3 --kernel-trace --stats` over the same workload
    (child process: warm-up + a few replays of the step graph). Returns (avg_ns or None, detail)."""
    import csv
    import glob

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, {"error": "rocprofv3 not on PATH"}
    if _under_profiler():
        return None, {"skipped": "bench.py is already running under a profiler"}
    out_dir = tempfile.mkdtemp(prefix="bnb_kt_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", out_dir, "-o", "bench", "--"] + _child_cmd(extra_args)
    detail = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --prof-child " + " ".join(extra_args)}
    try:
        subprocess.run(cmd, cwd="/tmp", env=_child_env(), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=timeout_s, check=True)
        files = glob.glob(os.path.join(out_dir, "**", "*kernel_stats.csv"), recursive=True)
        if not files:
            raise RuntimeError("no kernel_stats.csv produced")
        avg = calls = None
        with open(files[0], newline="") as fh:
            for r in csv.DictReader(fh):
                if KERNEL_SUBSTR in r.get("Name", ""):
                    c = int(float(r["Calls"]))
                    if calls is None or c > calls:  # the instance the workload launches (most calls)
                        calls, avg = c, float(r["AverageNs"])
        if avg is None:
            raise RuntimeError(f"no {KERNEL_SUBSTR} row in kernel_stats.csv")
        detail.update({"calls": calls, "average_ns": round(avg, 1)})
        if profile_out:
            os.makedirs(os.path.dirname(os.path.abspath(profile_out)) or ".", exist_ok=True)
            shutil.copyfile(files[0], profile_out)
            detail["csv"] = profile_out
        return avg, detail
    except Exception as exc:  # informational: never lose the bench line over the profiler
        detail["error"] = f"{type(exc).__name__}: {exc}"[:300]
        return None, detail
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


PROF_LIB = os.path.join(ROOT, "bitsandbytes_amd", "libbitsandbytes_mi355x_prof.so")


def kernel_span_child(M, N, K, bs, qt, layers_n=LAYERS, replays=20):
    """(child process, measurement build of the library) The dominant kernel's own span per launch: every wavefront records
    s_memrealtime (100 MHz constant clock) at its first instruction and after its last store; span of a launch = last end -
    first start over all wavefronts. Same workload as the timed region: hipGraph replays of the dependent one-launch-per-layer
    step; every launch of the graph writes its own stamp region (the buffer address is baked into the node at capture)."""
    import torch

    import bitsandbytes_amd as bnb
    from bitsandbytes_amd.cextension import LIB_PATH

    assert os.path.samefile(str(LIB_PATH), PROF_LIB), LIB_PATH
    device = torch.device("cuda:0")
    layers, x = build_layers(device, layers_n, N, K, M, bs, qt, seed=0)
    region = 256 * 16 * 16                       # u64 words per launch: workgroups x wavefronts x 16 stamps
    buf = torch.zeros(layers_n * region, dtype=torch.int64, device=device)

    def step_fn():
        for i, (q, st) in enumerate(layers):
            bnb.lib.bnb_mi355x_set_stamp_buffer((buf.data_ptr() + 8 * i * region) | 1)   # bit 0: the two span stamps only
            bnb.matmul_4bit(x, q, st)
        bnb.lib.bnb_mi355x_set_stamp_buffer(None)

    for _ in range(2):
        step_fn()
    torch.cuda.synchronize()
    g = capture(step_fn)
    spans = []
    for _ in range(replays):
        buf.zero_()
        g.replay()
        torch.cuda.synchronize()
        t = buf.view(layers_n, 256 * 16, 16)
        start, end = t[:, :, 13], t[:, :, 14]
        used = start > 0
        big = torch.full_like(start, 2**62)
        first = torch.where(used, start, big).min(dim=1).values
        last = torch.where(used, end, torch.zeros_like(end)).max(dim=1).values
        spans.append((last - first).double().cpu() * 0.01)                             # 100 MHz ticks -> us
    sp = torch.stack(spans)[2:]                                                          # drop two warm-up replays
    out = {"kernel_span_us": round(float(sp.mean()), 4), "p50": round(float(sp.median()), 3), "min": round(float(sp.min()), 2),
           "max": round(float(sp.max()), 2), "launches": int(sp.numel()), "wavefronts_per_launch": int(used[0].sum()),
           "clock": "s_memrealtime, 100 MHz (10 ns per tick; the mean over thousands of launches resolves finer)"}
    print("SPAN_JSON " + json.dumps(out), flush=True)


def kernel_span(extra_args, span_out=None, timeout_s=240):
    """Run the span child under the measurement build; None when that library is not there."""
    if not os.path.exists(PROF_LIB):
        return None, {"error": "libbitsandbytes_mi355x_prof.so not built (make -C bitsandbytes_amd/csrc profiling)"}
    env = _child_env()
    env["BNB_MI355X_LIBRARY"] = PROF_LIB
    cmd = [sys.executable, os.path.abspath(__file__), "--span-child", "--no-cpu-baseline", "--no-pmc", "--no-rocprof", "--no-sweep"] + list(extra_args)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return None, {"error": "span child timed out"}
    for ln in r.stdout.splitlines():
        if ln.startswith("SPAN_JSON "):
            d = json.loads(ln[len("SPAN_JSON "):])
            if span_out:
                with open(span_out, "w") as fh:
                    json.dump({"command": " ".join(cmd[1:]), **d}, fh, indent=1)
            return d["kernel_span_us"], d
    return None, {"error": "no span line", "stderr": r.stderr[-400:]}


def pmc_traffic(extra_args, timeout_s=180, substrings=None, calls_per_sweep=None):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC counters, collected the way
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
    passes (they do not fit one TCC pass), no tracing domains besides the kernel trace; both counters are
    reported in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes,
    so it is doubled. WRITE_SIZE is uncalibrated on gfx950 (guide) and is only ~0.1 % of this kernel's
    traffic. Returns (bytes_per_launch or None, detail dict). With `substrings` (several kernels per call of the op: a split-K
    kernel and its finalize launch) the counters of every matching dispatch of the second sweep are summed and divided by
    `calls_per_sweep`: bytes per CALL of the op."""
    import csv
    import glob

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, {"error": "rocprofv3 not on PATH"}
    if _under_profiler():
        return None, {"skipped": "bench.py is already running under a profiler; run it bare for the PMC passes"}
    detail, vals = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix="bnb_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--"] + \
            _child_cmd(list(extra_args) + ["--prof-eager"])
        try:
            subprocess.run(cmd, cwd="/tmp", env=_child_env(), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=timeout_s, check=True)
            rows, by_kernel = [], {}
            for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    recs = list(csv.DictReader(fh))
                recs.sort(key=lambda r: int(float(r.get("Dispatch_Id", 0) or 0)))
                for r in recs:
                    if any(sub in r.get("Kernel_Name", "") for sub in (substrings or (KERNEL_SUBSTR,))) and r.get("Counter_Name") == counter:
                        rows.append(float(r["Counter_Value"]))
                        by_kernel.setdefault(r["Kernel_Name"].split("<")[0].split("(")[0], []).append(float(r["Counter_Value"]))
            if not rows:
                raise RuntimeError("no counter rows for the kernel")
            if calls_per_sweep:
                # several kernels per call of the op: every kernel's last `calls_per_sweep` dispatches = the second sweep over the rotation
                vals[counter] = sum(sum(v[-calls_per_sweep:]) / len(v[-calls_per_sweep:]) for v in by_kernel.values())
                detail[counter + "_kernels"] = {k: round(sum(v[-calls_per_sweep:]) / len(v[-calls_per_sweep:]), 1) for k, v in by_kernel.items()}
            else:
                rows = rows[len(rows) // 2:]  # second sweep over the rotation
                vals[counter] = sum(rows) / len(rows)
            detail[counter + "_KiB_per_launch_raw"] = round(vals[counter], 1)
            detail[counter + "_dispatches"] = len(rows)
        except Exception as exc:
            detail["error"] = f"{counter}: {type(exc).__name__}: {exc}"[:300]
            return None, detail
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    traffic = 2.0 * vals["FETCH_SIZE"] * 1024.0 + vals["WRITE_SIZE"] * 1024.0
    detail["correction"] = "bytes = 2 x FETCH_SIZE[KiB] x 1024 (gfx950 64-B tally of 128-B requests) + WRITE_SIZE[KiB] x 1024"
    return traffic, detail


def config3_leg(device, with_pmc):
    """BASELINE.json configs[2] - gemm_4bit NF4 bf16 M = 64, N = K = 8192, the one compute-side configuration - through the public op:
    a hipGraph over 8 distinct layers (302 MB of packed weights + scales: beyond the 256-MiB Infinity Cache), HIP events around a
    >= 20 ms region; the kernel family the library reports; HBM traffic per call from two PMC child passes (kernel + finalize)."""
    import bitsandbytes_amd as bnb

    M3, N3, K3, L3 = 64, 8192, 8192, 8
    layers3, x3 = build_layers(device, L3, N3, K3, M3, 64, "nf4", seed=77)

    def fn():
        for q, st in layers3:
            bnb.matmul_4bit(x3, q, st)

    g = capture(fn)
    fam = KERNEL_FAMILIES.get(int(bnb.lib.bnb_mi355x_last_gemm_kernel()), "?")
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 150
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (reps * L3) * 1e3
    nbytes, flops = algorithmic_bytes(M3, N3, K3, 64), 2 * M3 * N3 * K3
    out = {"workload": "gemm_4bit NF4 bf16 M=64 N=K=8192 blocksize=64 fp32-absmax (BASELINE.json configs[2]) through bitsandbytes_amd.matmul_4bit; "
                       f"hipGraph of {L3} distinct layers ({L3 * nbytes / 1e6:.0f} MB: HBM-resident), HIP events around {reps} replays",
           "kernel": fam + (" + its split-K finalize launch" if fam in ("gemm4_mfma_pc_kernel", "gemm4_mfma_kq_kernel") else ""),
           "us": round(us, 2), "bytes_per_call": nbytes, "flops_per_call": flops,
           "GBps": round(nbytes / us / 1e3, 1), "TFLOPs": round(flops / us / 1e6, 1),
           "frac_hbm": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4), "frac_mfma": round(flops / us / 1e6 / MFMA_PEAK_TFLOPS, 4),
           "bound": "hbm (arithmetic intensity 216 FLOP/B, below the ~300 FLOP/B machine balance)", "traffic": None}
    del layers3
    if with_pmc:
        traffic, detail = pmc_traffic(["--m", str(M3), "--n", str(N3), "--k", str(K3), "--blocksize", "64", "--quant-type", "nf4", "--layers", str(L3)],
                                      substrings=("gemm4_",), calls_per_sweep=L3)
        out["traffic"] = None if traffic is None else round(traffic)
        out["traffic_detail"] = detail
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--m", type=int, default=1, help="activation rows (headline: 1)")
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--blocksize", type=int, default=64)
    ap.add_argument("--quant-type", default="nf4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep", action="store_true", help="(default at N = 1) also time M = 1..64 at N = K = 4096: the headline sweep of BASELINE.json's metric")
    ap.add_argument("--no-sweep", action="store_true", help="skip the M = 1..64 sweep")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC passes that fill roofline.traffic")
    ap.add_argument("--no-rocprof", action="store_true", help="skip the rocprofv3 kernel-trace pass (roofline then falls back to HIP events)")
    ap.add_argument("--profile-out", default=None, help="copy the rocprofv3 kernel_stats.csv of the roofline pass here (e.g. profiles/r2_bench_kernel_stats.csv)")
    ap.add_argument("--no-peer-gather", action="store_true", help="multi-GPU: RCCL all_gather_into_tensor per layer instead of the one-shot peer kernel")
    ap.add_argument("--sharded-path", action="store_true",
                    help="run the multi-GPU code path (ShardedLinear4bit shards, bucketed RCCL all-gather, per-layer "
                         "gather) even at world size 1: how that path is exercised on a 1-GPU box")
    ap.add_argument("--no-config3", action="store_true", help="skip the BASELINE.json configs[2] leg (M = 64, N = K = 8192)")
    ap.add_argument("--no-span", action="store_true", help="skip the kernel-span leg (roofline then falls back to the rocprofv3 average)")
    ap.add_argument("--span-out", default=None, help="write the kernel-span measurement here (e.g. profiles/r3_bench_kernel_span.json)")
    ap.add_argument("--span-child", action="store_true", help=argparse.SUPPRESS)  # span measurement under the measurement build
    ap.add_argument("--prof-child", action="store_true", help=argparse.SUPPRESS)  # workload run under rocprofv3
    ap.add_argument("--prof-eager", action="store_true", help=argparse.SUPPRESS)  # ... enqueued eagerly (PMC passes serialise dispatches)
    ap.add_argument("--layers", type=int, default=LAYERS, help=argparse.SUPPRESS)  # child passes of the config3 leg: a shorter rotation
    args = ap.parse_args()
    if args.span_child:
        kernel_span_child(args.m, args.n, args.k, args.blocksize, args.quant_type)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import torch.distributed as dist

    multi = world > 1 or args.sharded_path
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # (--sharded-path without a launcher: any free port)
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    import bitsandbytes_amd as bnb
    from bitsandbytes_amd.parallel import ShardedLinear4bit

    assert bnb.lib, "native HIP library missing: run `python -c 'import __graft_entry__ as g; g.build()'`"

    M, N, K, bs, qt = args.m, args.n, args.k, args.blocksize, args.quant_type
    global KERNEL_SUBSTR
    KERNEL_SUBSTR, kernel_label = dominant_kernel(M, N, K, bs, qt)
    peer = chain = None
    chain_info = None
    if not multi:
        layers, x = build_layers(device, args.layers if args.prof_child else LAYERS, N, K, M, bs, qt, seed=1234 + rank)
        nbytes_layer = algorithmic_bytes(M, N, K, bs)
        nbytes_step = LAYERS * nbytes_layer
        flops_step = LAYERS * 2 * M * N * K

        # ---- one step = the 128 layers, each its own launch through the public op
        def step_fn():
            for q, st in layers:
                bnb.matmul_4bit(x, q, st)
    else:
        # ---- N-sharded decode chain (bitsandbytes_amd.parallel semantics: rows sharded, x replicated, no reduction). The model
        # is an MLP stack, up (H -> F) / down (F -> H), every layer sharded by output features over the ranks and every layer's
        # gathered y the next layer's x. H x F = world x 4096^2, so that EVERY rank's shard of EVERY layer holds 4096^2 weights - the
        # headline layer's bytes - at any world size (weak scaling): world 1: 4096 / 4096, 2: 4096 / 8192, 4: 8192 / 8192, 8: 8192 /
        # 16384. One step = one activation row through LAYERS such layers = LAYERS x (shard kernel + all-gather of y).
        geo = sharded_chain_dims(world, N, K)
        if geo is None:
            sys.exit(f"bench.py: cannot build the sharded chain for world size {world} (needs a power of two)")
        H, Fd, dims = geo                                 # dims: (rows of this rank's shard, K) of an up / a down layer
        g = torch.Generator(device=device).manual_seed(1234 + rank)
        import bitsandbytes_amd.functional as F4

        shard_mods = []
        for li in range(LAYERS):
            ns_l, k_l = dims[li & 1]
            W = (torch.randn(ns_l, k_l, device=device, generator=g) / k_l**0.5).to(torch.bfloat16)
            q, st = F4.quantize_4bit(W, blocksize=bs, quant_type=qt)
            shard_mods.append((q, st, world * ns_l))
            del W
        x = torch.randn(M, H, device=device, generator=torch.Generator(device=device).manual_seed(99)).to(torch.bfloat16)  # same on every rank
        nbytes_step = sum(algorithmic_bytes(M, int(st.shape[0]), int(st.shape[1]), bs) for _, st, _ in shard_mods)
        nbytes_layer = nbytes_step // LAYERS
        flops_step = sum(2 * M * int(st.shape[0]) * int(st.shape[1]) for _, st, _ in shard_mods)
        os.environ.setdefault("BNB_MI355X_PEER_WAIT_POLLS", "3000000")  # (a few seconds: a node on which the peer kernels cannot work falls back quickly)
        why = None
        # the separate-gather form (kernel, then a collective per layer): the one-shot peer kernel when it constructs and
        # reproduces RCCL's all-gather on live data, else RCCL's all_gather_into_tensor. Every rank takes the same branch.
        if not args.no_peer_gather:
            try:
                from bitsandbytes_amd.peer import PeerAllGather

                peer = PeerAllGather(max_bytes=64 * 1024)  # (raises on every rank or on none)
            except Exception as exc:  # noqa: BLE001
                peer, why = None, f"{type(exc).__name__}: {exc}"
            if peer is not None:
                same = True
                try:
                    for i in range(8):
                        t_chk = (torch.randn(M, N, device=device) + rank + i).bfloat16()
                        want = torch.empty(world * M, N, device=device, dtype=torch.bfloat16)
                        dist.all_gather_into_tensor(want, t_chk)
                        got = peer.all_gather(t_chk)
                        torch.cuda.synchronize()
                        # (no early exit on a rank-local result: every rank issues the same sequence of collectives whatever it
                        # sees - a rank that left the loop alone would meet the others' all-gathers with its all-reduce)
                        same = same and bool(torch.equal(got, want)) and peer.status() == 0
                except Exception as exc:  # noqa: BLE001
                    same, why = False, f"{type(exc).__name__}: {exc}"
                agree = torch.tensor([1 if same else 0], device=device)
                dist.all_reduce(agree, op=dist.ReduceOp.MIN)
                if int(agree.item()) != 1:
                    why = why or "it does not reproduce the group's all-gather on this node"
                    try:
                        peer.close()
                    except Exception:  # noqa: BLE001
                        pass
                    peer = None
            if peer is None and rank == 0:
                print(f"bench: peer all-gather not used ({why}); RCCL all_gather_into_tensor per layer", file=sys.stderr)
        separate = [ShardedLinear4bit(q, st, out_features=nf, group=None, always_gather=True, peer=peer) for q, st, nf in shard_mods]

        def separate_fn():
            y = x
            for sh in separate:
                y = sh(y)
            return y

        # the fused form (bitsandbytes_amd.peer.PeerChain: the gather inside the gemv launches - one launch per layer): used for
        # `value` when it constructs on this node AND reproduces the separate form bit for bit on live data - its authors could only
        # run it between processes sharing one GPU. The decision is collective.
        fused_mod = None
        why_chain = None
        if not args.no_peer_gather:
            try:
                from bitsandbytes_amd.parallel import ShardedLinear4bitChain
                from bitsandbytes_amd.peer import PeerChain

                chain = PeerChain(max_values=max(H, Fd))
                fused_mod = ShardedLinear4bitChain([ShardedLinear4bit(q, st, out_features=nf, group=None) for q, st, nf in shard_mods], chain)
            except Exception as exc:  # noqa: BLE001
                chain, fused_mod, why_chain = None, None, f"{type(exc).__name__}: {exc}"
            ok = chain is not None
            if ok:
                try:
                    ok = bool(fused_mod.fused(x))
                    if not ok:
                        why_chain = "the chain's shapes are outside the fused form"
                    else:
                        y_sep = separate_fn()
                        for _ in range(3):
                            y_fus = fused_mod(x)
                            torch.cuda.synchronize()
                            ok = ok and bool(torch.equal(y_fus, y_sep)) and chain.status() == 0
                        if not ok:
                            why_chain = "it does not reproduce the separate-gather form on this node"
                except Exception as exc:  # noqa: BLE001
                    ok, why_chain = False, f"{type(exc).__name__}: {exc}"
            agree = torch.tensor([1 if ok else 0], device=device)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if int(agree.item()) != 1:
                why_chain = why_chain or "another rank could not use it"
                if chain is not None:
                    try:
                        chain.close()
                    except Exception:  # noqa: BLE001
                        pass
                chain = fused_mod = None
            if chain is None and rank == 0:
                print(f"bench: fused peer chain not used ({why_chain}); kernel + separate all-gather per layer", file=sys.stderr)
        chain_info = {"H": H, "F": Fd, "up_shard": list(dims[0]), "down_shard": list(dims[1]),
                      "gather": "fused into the gemv launches (peer.PeerChain)" if chain is not None else
                                ("separate one-shot peer kernel per layer (peer.PeerAllGather)" if peer is not None else "RCCL all_gather_into_tensor per layer"),
                      "fused_chain_not_used_because": why_chain}

        def step_fn():
            if fused_mod is not None:
                fused_mod(x)
            else:
                separate_fn()

    if args.prof_child:
        if args.prof_eager:
            for _ in range(2):  # PMC passes serialise and count every dispatch: two eager sweeps over the rotation
                step_fn()
            torch.cuda.synchronize()
            return
        g = capture(step_fn)
        for _ in range(8):
            g.replay()
        torch.cuda.synchronize()
        return

    step_graph = None
    try:
        # (RCCL collectives are capturable; where a stack refuses, the per-layer step is enqueued eagerly: same work, host-bound)
        step_graph = capture(step_fn)
    except Exception as exc:  # noqa: BLE001
        if not multi:
            raise
        torch.cuda.synchronize()
        print(f"bench: per-layer step not capturable here ({type(exc).__name__}: {exc}); enqueuing eagerly", file=sys.stderr)

    def run_steps(nsteps):
        """Enqueue exactly nsteps steps."""
        for _ in range(nsteps):
            if step_graph is not None:
                step_graph.replay()
            else:
                step_fn()

    def barrier():
        if multi:
            dist.barrier()

    # ---- warm-up
    run_steps(args.warmup)
    torch.cuda.synchronize()

    # ---- timed region: exactly args.steps steps
    # Two clocks on the same region: HIP events recorded on the stream the kernels are launched on (torch's current stream; the
    # clock the contract names for roofline.achieved) and the host's wall clock around synchronize (the clock `value` uses and the
    # driver can check from outside; it can only be the slower of the two).
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    run_steps(args.steps)
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed_events = ev0.elapsed_time(ev1) * 1e-3
    if multi:
        t = torch.tensor([elapsed, elapsed_events], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_events = float(t[0].item()), float(t[1].item())

    # ---- secondary timings (outside the timed region)
    def graph_us_per_launch(fn, launches, reps=10):
        g = capture(fn)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * launches) * 1e3

    def per_launch_us(m_rows, reps=10):
        xm = x if m_rows == M else torch.randn(m_rows, K, device=device).to(torch.bfloat16)

        def fn():
            for q, st in layers:
                bnb.matmul_4bit(xm, q, st)
        return graph_us_per_launch(fn, LAYERS, reps)

    side_forms = None
    if not multi:
        kernel_us_events = per_launch_us(M)
    else:
        # the same chain in its other forms, beside `value` (every rank enqueues the same work; MAX over ranks):
        #   kernels alone: the shard kernels of the chain on resident inputs, no exchange at all - what a layer costs without the gather
        #   separate gather: shard kernel, then a collective per layer (round 3's form)
        def alone_fn():
            for (q, st, _nf) in shard_mods:
                bnb.matmul_4bit(xs_by_k[int(st.shape[1])], q, st)

        xs_by_k = {int(st.shape[1]): torch.randn(M, int(st.shape[1]), device=device).to(torch.bfloat16) for _, st, _ in shard_mods}

        def timed_form(fn):
            barrier()
            try:
                us = graph_us_per_launch(fn, LAYERS, reps=5)
            except Exception:  # noqa: BLE001  (a collective that cannot be captured here)
                torch.cuda.synchronize()
                return None
            t_ = torch.tensor([us], device=device, dtype=torch.float64)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            return round(float(t_.item()), 3)

        kernel_us_events = timed_form(alone_fn) or round(elapsed / args.steps / LAYERS * 1e6, 3)
        side_forms = {"us_per_layer_kernels_alone": kernel_us_events,
                      "us_per_layer_kernel_plus_separate_gather": timed_form(separate_fn),
                      "us_per_layer_timed_region": round(elapsed / args.steps / LAYERS * 1e6, 3)}

    sweep = grouped = None
    if not multi and (args.sweep or not args.no_sweep) and rank == 0:
        sweep = []
        for m_rows in (1, 2, 4, 8, 16, 32, 64):
            t_us = per_launch_us(m_rows, reps=5)
            fam = KERNEL_FAMILIES.get(int(bnb.lib.bnb_mi355x_last_gemm_kernel()), "?")  # (the family the capture's last call launched)
            gbps, tfl = algorithmic_bytes(m_rows, N, K, bs) / t_us / 1e3, 2 * m_rows * N * K / t_us / 1e6
            sweep.append({"M": m_rows, "us_per_launch": round(t_us, 2), "kernel": fam, "GBps": round(gbps, 1), "TFLOPs": round(tfl, 2),
                          "frac_hbm": round(gbps / HBM_PEAK_GBS, 4), "frac_mfma": round(tfl / MFMA_PEAK_TFLOPS, 4)})
        gsz = 4

        def grouped_us(m_rows):
            xm = x if m_rows == M else torch.randn(m_rows, K, device=device).to(torch.bfloat16)

            def grouped_fn():
                for i in range(0, LAYERS, gsz):
                    grp = layers[i:i + gsz]
                    bnb.matmul_4bit_grouped(xm, [q for q, _ in grp], [st for _, st in grp])
            t = graph_us_per_launch(grouped_fn, LAYERS, reps=5)
            return t, KERNEL_FAMILIES.get(int(bnb.lib.bnb_mi355x_last_gemm_kernel()), "?")
        t_grp, _ = grouped_us(M)
        grouped = {"group_size": gsz, "us_per_layer": round(t_grp, 3), "GBps": round(nbytes_layer / t_grp / 1e3, 1),
                   "frac_of_hbm_peak": round(nbytes_layer / t_grp / 1e3 / HBM_PEAK_GBS, 4),
                   "what": "the same 128 layers as 32 launches of matmul_4bit_grouped (4 matrices sharing x per launch: Q/K/V/O- or "
                           "gate/up-style); informational - the headline keeps one launch per layer. sweep: the same at M = 1 ... 32 rows "
                           "(one launch of the streaming kernel at one row, of the streaming MFMA kernel from two rows on - in row passes of 16 above 16 rows)"}
        gsweep = []
        for m_rows in (1, 2, 4, 8, 16, 32):
            t_us, fam = grouped_us(m_rows)
            gbps = algorithmic_bytes(m_rows, N, K, bs) / t_us / 1e3
            gsweep.append({"M": m_rows, "us_per_layer": round(t_us, 2), "kernel": fam, "GBps": round(gbps, 1), "frac_hbm": round(gbps / HBM_PEAK_GBS, 4)})
        grouped["sweep"] = gsweep

    config3 = None
    if not multi and not args.no_config3 and rank == 0 and (M, N, K) == (1, 4096, 4096):
        try:
            config3 = config3_leg(device, with_pmc=not args.no_pmc)
        except Exception as exc:  # informational: never lose the headline over it
            config3 = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    if rank == 0:
        total_steps = args.steps * world
        value = nbytes_step * total_steps / elapsed / 1e9
        extra = ["--m", str(M), "--n", str(N), "--k", str(K), "--blocksize", str(bs), "--quant-type", qt]
        avg_ns, kt_detail = (None, {"skipped": "--no-rocprof or multi-GPU run"})
        if not multi and not args.no_rocprof:
            avg_ns, kt_detail = rocprof_kernel_stats(extra, args.profile_out)
        span_us, span_detail = (None, {"skipped": "--no-span or multi-GPU run"})
        if not multi and not args.no_span:
            span_us, span_detail = kernel_span(extra, args.span_out)
        # Round 5: the figure of record is the one the PRODUCT binary's own clock gives - HIP events around the timed region divided by
        # the launches in it (kernel + the boundary to the next dependent launch): what the contract prescribes and what the driver's
        # clock can check. The in-kernel span (measurement build of the library: a different binary) and the rocprofv3 average (the
        # profiler serialises dispatches: 128 x its figure exceeds the step) stay on the line as frac_span / frac_rocprof.
        kernel_us = elapsed_events / args.steps / LAYERS * 1e6
        method = (f"frac: HIP events (recorded on torch's current stream, the stream the kernels are launched on) around the timed region - {args.steps} "
                  f"hipGraph replays of the {LAYERS}-layer step - divided by the launches in it: the dominant kernel + its boundary to the next "
                  "dependent launch, product library. frac_wall: the same region on the host's wall clock (time.perf_counter around "
                  "torch.cuda.synchronize: the clock `value` and ms_per_step use). frac_span: the kernel's own span from in-kernel s_memrealtime "
                  "stamps (measurement build, child process). frac_rocprof: average duration from `rocprofv3 --kernel-trace --stats` over the "
                  "same workload (serialised dispatches)")
        achieved = nbytes_layer / (kernel_us * 1e-6) / 1e9
        line = {
            "metric": "NF4 gemv_4bit / Linear4bit decode forward GB/s (M=1, N=K=4096; algorithmic bytes / time)",
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "tflops": round(flops_step * total_steps / elapsed / 1e12, 4),
            "config": {
                "workload": (f"gemv_4bit {qt.upper()} bf16 M={M} N=K={N}x{K} blocksize={bs} fp32-absmax (BASELINE.json configs[1]); "
                             f"one step = one activation row through {LAYERS} distinct layers, one launch per layer") if not multi else
                            (f"gemv_4bit {qt.upper()} bf16 M={M} blocksize={bs} fp32-absmax, N-sharded x{world}: decode step through {LAYERS} layers of an "
                             f"MLP stack H={chain_info['H']} / F={chain_info['F']} (up shard {chain_info['up_shard'][0]}x{chain_info['up_shard'][1]}, down shard "
                             f"{chain_info['down_shard'][0]}x{chain_info['down_shard'][1]}: {N}x{K} weights per rank and layer = BASELINE.json configs[1]'s layer at every N)"),
                "layers_per_step": LAYERS,
                "bytes_per_layer": nbytes_layer,
                "bytes_per_step": nbytes_step,
                "us_per_layer": round(elapsed / args.steps / LAYERS * 1e6, 3),
                "launch": (f"hipGraph replay per step ({LAYERS} dependent launches of bitsandbytes_amd.matmul_4bit), every step of warm-up and timed region"
                           if not multi else f"{LAYERS} x (shard kernel + all-gather of its outputs) per step, "
                                             f"{'one hipGraph per step' if step_graph is not None else 'eager'}; gather = {chain_info['gather']}"),
                "timed_region_s": round(elapsed, 6),
                "parallelism": (f"rows (output features) sharded x{world}, x replicated, one all-gather of y per layer, no reduction "
                                f"(bitsandbytes_amd.parallel); gather = {chain_info['gather']}, checked bit for bit against the next simpler form at "
                                "start-up on every rank; N > 1 is unmeasured on hardware by the builder (1-GPU boxes only)") if multi else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_label + (", 16 wavefronts" if M == 1 and KERNEL_SUBSTR == "gemv4_stream_kernel" else ""),
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "frac_span": None if span_us is None else round(nbytes_layer / (span_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "frac_wall": round(nbytes_layer / (elapsed / args.steps / LAYERS) / 1e9 / HBM_PEAK_GBS, 4),
                "frac_rocprof": None if avg_ns is None else round(nbytes_layer / (avg_ns * 1e-9) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": None,
                "kernel_us": round(kernel_us, 3),
                "kernel_us_launch_to_launch_events": round(kernel_us_events, 3),
                "kernel_us_rocprof_stats": None if avg_ns is None else round(avg_ns / 1e3, 3),
                "kernel_span": span_detail,
                "kernel_us_span": None if span_us is None else round(span_us, 3),
                "fits_in_step": None if span_us is None else bool(span_us * LAYERS <= elapsed / args.steps * 1e6),
                "timed_region_events_s": round(elapsed_events, 6),
                "method": method,
                "kernel_trace": kt_detail,
            },
        }
        if not multi and not args.no_pmc:
            traffic, detail = pmc_traffic(extra)
            line["roofline"]["traffic"] = None if traffic is None else round(traffic)
            line["roofline"]["traffic_detail"] = detail
        if side_forms is not None:
            line["sharded_chain"] = {**chain_info, **side_forms}
        if sweep is not None:
            line["headline_sweep_N4096_K4096"] = sweep
        if grouped is not None:
            line["grouped"] = grouped
        if config3 is not None:
            line["config3"] = config3
        if not multi and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(M, N, K, bs, qt)
            except Exception as exc:  # baseline is informational; never lose the GPU line over it
                line["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": f"failed: {exc}"}
    if multi:
        for obj in (chain, peer):
            if obj is not None:
                obj.close()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST line of stdout: RCCL prints its version banner through C stdio, which is block-buffered
        # when stdout is a pipe and would otherwise be flushed after this line, at exit
        import ctypes

        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
