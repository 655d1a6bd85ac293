#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X 4-bit path (contract in the task statement).

Workload at N = 1 (BASELINE.json configs[1], the configuration the headline metric is quoted on):
    gemv_4bit / Linear4bit forward, NF4, bf16, M = 1, N = K = 4096, blocksize 64, fp32 absmax.
One "step" = one forward pass of one such layer on one synthetic activation row. Steps rotate over
LAYERS = 64 distinct layers (64 x 9.45 MB = 605 MB > the 256 MiB Infinity Cache), so every step
streams its weights from HBM; inputs are resident in HBM before the timed region.

    value = algorithmic bytes per step x steps / wall time      [GB/s, whole job, all ranks]
    algorithmic bytes per step = N*K/2 + 4*N*K/bs + 2*M*K + 2*M*N = 9 453 568   

Timed region: the K steps are enqueued as replays of a hipGraph holding GRAPH_CHUNK consecutive
steps (launch-bound inner loop -> graph, as on a real decode loop), bracketed by barrier +
synchronize; MAX over ranks.

--gpus N > 1 (weak scaling): every rank owns a full-size 4096-row shard of an N*4096-row layer
(bitsandbytes_amd.parallel semantics: x replicated, rows sharded, no reduction). The y shards are
re-assembled by RCCL all-gathers, bucketed GRAPH_CHUNK steps per collective and issued on a side
stream so they overlap the next chunk's weight streaming.

Extra objects on the JSON line: "roofline" (dominant kernel, HIP-event timed per launch inside the
same process; "traffic" from two rocprofv3 PMC passes run as child processes), at N = 1 on rank 0
"cpu_baseline" (the reference's own AVX512-BF16 fused CPU gemv from oracle/_ref when the host supports it,
else the scalar port from oracle/) and "headline_sweep_N4096_K4096" (M = 1..64, the other half of
BASELINE.json's metric; --no-sweep skips it).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
LAYERS = 64
GRAPH_CHUNK = 64


def algorithmic_bytes(M, N, K, bs, elt=2):
    return N * K // 2 + 4 * (N * K // bs) + elt * M * K + elt * M * N


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
        "ms_per_step": round(dt / iters * 1e3, 4),
        "sample": f"{iters} forward passes of the same M={M} N=K={N} workload in {dt:.1f} s; {what}",
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
        out["unfused_dequantize_linear"] = {"ms_per_step": round(ms, 3), "iters": n, "what": deq_kind,
                                            "torch_threads": torch.get_num_threads()}
        ms, n = timed(lambda: O.quantize_4bit(W, blocksize, quant_type))
        out["quantize_4bit"] = {"ms_per_call": round(ms, 3), "iters": n, "cores": 1,
                                "what": f"oracle scalar port of backends/default/ops.py:225-259 on a {N}x{K} bf16 weight"}
    except Exception as exc:  # informational
        out["other_legs_error"] = f"{type(exc).__name__}: {exc}"[:200]
    return out


def pmc_traffic(extra_args, kernel_substr="gemv4_dot_kernel", timeout_s=90):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC counters, collected the way
    /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
    passes (they do not fit one TCC pass), no tracing domains besides the kernel trace; both counters are
    reported in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes,
    so it is doubled. WRITE_SIZE is uncalibrated on gfx950 (guide) and is only ~0.1 % of this kernel's
    traffic. Each pass re-runs this script with --pmc-child (two eager sweeps over the 64-layer rotation).
    Returns (bytes_per_launch or None, detail dict)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, {"error": "rocprofv3 not on PATH"}
    # never nest profilers: when this process is itself being traced (rocprofv3 -- python bench.py) the counter
    # passes are skipped instead of starting a second profiler inside the first
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ) or \
            "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, {"skipped": "bench.py is already running under a profiler; run it bare for the PMC passes"}
    detail = {}
    vals = {}
    here = os.path.abspath(__file__)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix="bnb_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--",
               sys.executable, here, "--pmc-child", "--no-cpu-baseline"] + list(extra_args)
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=timeout_s, check=True)
            rows = []
            for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for r in csv.DictReader(fh):
                        if kernel_substr in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                            rows.append(float(r["Counter_Value"]))
            if not rows:
                raise RuntimeError("no counter rows for the kernel")
            rows = rows[len(rows) // 2:]  # second sweep over the rotation
            vals[counter] = sum(rows) / len(rows)
            detail[counter + "_KiB_per_launch_raw"] = round(vals[counter], 1)
            detail[counter + "_dispatches"] = len(rows)
        except Exception as exc:  # informational: never lose the bench line over the profiler
            detail["error"] = f"{counter}: {type(exc).__name__}: {exc}"[:300]
            return None, detail
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    traffic = 2.0 * vals["FETCH_SIZE"] * 1024.0 + vals["WRITE_SIZE"] * 1024.0
    detail["correction"] = "bytes = 2 x FETCH_SIZE[KiB] x 1024 (gfx950 64-B tally of 128-B requests) + WRITE_SIZE[KiB] x 1024"
    return traffic, detail


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6400)
    ap.add_argument("--warmup", type=int, default=640)
    ap.add_argument("--m", type=int, default=1, help="activation rows (headline: 1)")
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--blocksize", type=int, default=64)
    ap.add_argument("--quant-type", default="nf4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="enqueue steps eagerly instead of replaying a hipGraph")
    ap.add_argument("--sweep", action="store_true", help="(default at N = 1) also time M = 1..64 at N = K = 4096: the headline sweep of BASELINE.json's metric")
    ap.add_argument("--no-sweep", action="store_true", help="skip the M = 1..64 sweep")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC passes that fill roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # workload run under rocprofv3 --pmc
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    import bitsandbytes_amd as bnb
    from bitsandbytes_amd.backends import hip

    assert bnb.lib, "native HIP library missing: run `python -c 'import __graft_entry__ as g; g.build()'`"

    M, N, K, bs, qt = args.m, args.n, args.k, args.blocksize, args.quant_type
    layers, x = build_layers(device, LAYERS, N, K, M, bs, qt, seed=1234 + rank)
    nbytes_step = algorithmic_bytes(M, N, K, bs)
    flops_step = 2 * M * N * K

    # output buckets: GRAPH_CHUNK steps of this rank's y shard, double-buffered
    buckets = [torch.empty(GRAPH_CHUNK, M, N, device=device, dtype=torch.bfloat16) for _ in range(2)]
    gathered = [torch.empty(world * GRAPH_CHUNK, M, N, device=device, dtype=torch.bfloat16) for _ in range(2)] if world > 1 else None

    def run_step(i, out):
        q, st = layers[i % LAYERS]
        hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, out=out)

    def run_chunk_eager(base, bucket):
        for j in range(GRAPH_CHUNK):
            run_step(base + j, bucket[j])

    if args.pmc_child:
        # two eager passes over the HBM-resident rotation; rocprofv3 --pmc serialises and counts every dispatch
        for base in (0, GRAPH_CHUNK):
            run_chunk_eager(base, buckets[0])
        torch.cuda.synchronize()
        return

    # LAYERS == GRAPH_CHUNK, so every chunk touches the same layer sequence: one graph per bucket
    graphs = None
    if not args.no_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run_chunk_eager(0, buckets[0])  # warm-up outside capture
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphs = []
        for b in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run_chunk_eager(0, buckets[b])
            graphs.append(g)

    comm_stream = torch.cuda.Stream() if world > 1 else None
    pending = [None, None]

    def run_steps(nsteps):
        """Enqueue exactly nsteps steps (+ the bucketed all-gathers when world > 1)."""
        done, c = 0, 0
        while done < nsteps:
            b = c & 1
            if world > 1 and pending[b] is not None:
                torch.cuda.current_stream().wait_event(pending[b])  # bucket b's previous gather finished
            left = nsteps - done
            if left >= GRAPH_CHUNK:
                if graphs is not None:
                    graphs[b].replay()
                else:
                    run_chunk_eager(done, buckets[b])
                n_now = GRAPH_CHUNK
            else:
                for j in range(left):
                    run_step(done + j, buckets[b][j])
                n_now = left
            if world > 1:
                ready = torch.cuda.Event()
                ready.record()
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(ready)
                    dist.all_gather_into_tensor(gathered[b].view(world * GRAPH_CHUNK * M, N),
                                                buckets[b].view(GRAPH_CHUNK * M, N))
                    ev = torch.cuda.Event()
                    ev.record()
                pending[b] = ev
            done += n_now
            c += 1
        if world > 1:
            torch.cuda.current_stream().wait_stream(comm_stream)

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warm-up
    run_steps(args.warmup)
    torch.cuda.synchronize()

    # ---- timed region: exactly args.steps steps
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline leg: launch-to-launch time of the dominant kernel, HIP events on the launch stream.
    # Primary figure: events bracket whole hipGraph replays of GRAPH_CHUNK back-to-back launches
    # (HBM-resident rotation), divided by the launch count - i.e. what one launch costs in a dependent
    # stream, boundary included. Secondary: per-launch event brackets around eager launches (these
    # also include the event-record commands, so they over-state the kernel by ~3 us).
    def per_launch_us(m_rows, reps=10):
        xm = x if m_rows == M else torch.randn(m_rows, K, device=device).to(torch.bfloat16)
        outs = torch.empty(GRAPH_CHUNK, m_rows, N, device=device, dtype=torch.bfloat16)

        def chunk():
            for j in range(GRAPH_CHUNK):
                q, st = layers[j % LAYERS]
                hip._gemm_4bit_fused(xm, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None,
                                     out=outs[j])

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chunk()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chunk()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * GRAPH_CHUNK) * 1e3

    kernel_us = per_launch_us(M)
    achieved = nbytes_step / (kernel_us * 1e-6) / 1e9

    n_ev = min(args.steps, 256)
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev)]
    for i in range(n_ev):
        starts[i].record()
        run_step(i, buckets[0][i % GRAPH_CHUNK])
        ends[i].record()
    torch.cuda.synchronize()
    per_launch_ms = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
    kept = per_launch_ms[: max(1, int(0.9 * n_ev))]
    kernel_us_events = sum(kept) / len(kept) * 1e3

    sweep = None
    if (args.sweep or (world == 1 and not args.no_sweep)) and rank == 0:
        sweep = []
        for m_rows in (1, 2, 4, 8, 16, 32, 64):
            t_us = per_launch_us(m_rows, reps=5)
            sweep.append({"M": m_rows, "us_per_launch": round(t_us, 2),
                          "GBps": round(algorithmic_bytes(m_rows, N, K, bs) / t_us / 1e3, 1),
                          "TFLOPs": round(2 * m_rows * N * K / t_us / 1e6, 2)})

    if rank == 0:
        total_steps = args.steps * world
        value = nbytes_step * total_steps / elapsed / 1e9
        line = {
            "metric": "NF4 Linear4bit forward GB/s (gemv_4bit M=1, N=K=4096; algorithmic bytes / time)",
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
                "workload": f"gemv_4bit {qt.upper()} bf16 M={M} N=K={N}x{K} blocksize={bs} fp32-absmax "
                            "(BASELINE.json configs[1]); one step = one layer forward",
                "layers_in_rotation": LAYERS,
                "hbm_resident_bytes_rotated": LAYERS * nbytes_step,
                "bytes_per_step": nbytes_step,
                "launch": "eager" if graphs is None else f"hipGraph x{GRAPH_CHUNK} steps",
                "parallelism": f"rows sharded x{world}, all-gather bucketed x{GRAPH_CHUNK}" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "gemv4_dot_kernel<bf16>",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None,
                "kernel_us": round(kernel_us, 3),
                "kernel_us_event_brackets": round(kernel_us_events, 3),
                "method": f"HIP events around 10 hipGraph replays of {GRAPH_CHUNK} back-to-back launches over "
                          f"{LAYERS} distinct HBM-resident layers, divided by the launch count",
            },
        }
        if world == 1 and not args.no_pmc:
            extra = ["--m", str(M), "--n", str(N), "--k", str(K), "--blocksize", str(bs), "--quant-type", qt]
            traffic, detail = pmc_traffic(extra)
            line["roofline"]["traffic"] = None if traffic is None else round(traffic)
            line["roofline"]["traffic_detail"] = detail
        if sweep is not None:
            line["headline_sweep_N4096_K4096"] = sweep
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(M, N, K, bs, qt)
            except Exception as exc:  # baseline is informational; never lose the GPU line over it
                line["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": f"failed: {exc}"}
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
