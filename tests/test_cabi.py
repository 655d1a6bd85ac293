"""CPU: the C-ABI library builds, loads and exports every symbol include/bnb_mi355x.h declares.
No compute calls (there is no GPU here)."""
import ctypes as ct
import os
import re

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "bnb_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:void\*?|void|int|size_t|const char\*)\s+\*?([A-Za-z_][A-Za-z0-9_]*)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_declares_reference_abi():
    names = _header_symbols()
    for d in ("fp32", "fp16", "bf16"):
        assert f"cgemm_4bit_{d}" in names
        assert f"cgemm_4bit_inference_naive_{d}" in names
        assert f"cquantize_blockwise_{d}" in names and f"cdequantize_blockwise_{d}" in names
        for q in ("nf4", "fp4"):
            assert f"cquantize_blockwise_{d}_{q}" in names and f"cdequantize_blockwise_{d}_{q}" in names
    assert "get_context" in names and "cget_managed_ptr" in names


def test_library_exports_every_declared_symbol():
    from bitsandbytes_amd import cextension as ce

    assert ce.lib, f"{ce.LIB_PATH} not built: run __graft_entry__.build() / make -C bitsandbytes_amd/csrc"
    dll = ct.CDLL(str(ce.LIB_PATH))
    missing = [n for n in _header_symbols() if not hasattr(dll, n)]
    assert not missing, f"declared in include/bnb_mi355x.h but not exported: {missing}"
    assert set(ce.EXPORTED_SYMBOLS) <= set(_header_symbols())
    assert ce.lib.bnb_mi355x_version().decode().endswith("gfx950")


def test_library_exports_nothing_but_the_declared_c_abi():
    """Round-5 review: a drop-in .so exposes the reference's ABI + the documented extensions and nothing else. The library is built
    with -fvisibility=hidden and a linker version script (csrc/exports.map): its dynamic symbol table holds exactly the names
    include/bnb_mi355x.h declares - no bnb:: C++ symbol, no compiler-generated marker."""
    import shutil
    import subprocess

    from bitsandbytes_amd import cextension as ce

    nm = shutil.which("nm")
    if nm is None:
        import pytest

        pytest.skip("binutils nm not available")
    out = subprocess.run([nm, "-D", "--defined-only", str(ce.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    declared = set(_header_symbols())
    assert exported, "empty dynamic symbol table?"
    assert not (exported - declared), f"exported but not declared in include/bnb_mi355x.h: {sorted(exported - declared)[:10]}"
    assert not (declared - exported), f"declared but not exported: {sorted(declared - exported)[:10]}"


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """The product path must raise, never fall back, when the HIP library is absent."""
    import importlib

    import pytest

    from bitsandbytes_amd import cextension as ce

    stub = ce.MissingNativeLibrary("simulated")
    assert not stub
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        stub.cgemm_4bit_bf16
    monkeypatch.setenv("BNB_MI355X_LIBRARY", str(tmp_path / "nope.so"))
    ce2 = importlib.reload(ce)
    try:
        assert isinstance(ce2.lib, ce2.MissingNativeLibrary)
    finally:
        monkeypatch.delenv("BNB_MI355X_LIBRARY")
        importlib.reload(ce)


def test_native_dispatch_library_registers_gemm_4bit():
    """csrc/torch_dispatch.cpp -> libbitsandbytes_mi355x_torch.so: loads (linked against the product library next to it),
    provides the device kernel of bitsandbytes::gemm_4bit, and the Python glue steps aside; BNB_MI355X_PYTHON_DISPATCH=1 keeps
    the Python kernel instead."""
    import subprocess
    import sys

    code = (
        "import torch, bitsandbytes_amd as b\n"
        "from bitsandbytes_amd.backends import hip\n"
        "assert torch._C._dispatch_has_kernel_for_dispatch_key('bitsandbytes::gemm_4bit', 'CUDA')\n"
        "maps = open('/proc/self/maps').read()\n"
        "print('NATIVE' if hip.NATIVE_DISPATCH else 'PYTHON', 'libbitsandbytes_mi355x_torch.so' in maps)\n"
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert "NATIVE True" in out.stdout, out.stderr[-2000:]
    env = dict(os.environ, BNB_MI355X_PYTHON_DISPATCH="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env)
    assert "PYTHON False" in out.stdout, out.stderr[-2000:]


def test_no_cpu_kernels_registered_by_product():
    """Calling an op on CPU tensors must fail (no CPU kernel) unless a test registered the oracle."""
    import subprocess
    import sys

    code = (
        "import torch, bitsandbytes_amd as b\n"
        "try:\n"
        "    torch.ops.bitsandbytes.quantize_4bit.default(torch.randn(64), 64, 'nf4', torch.uint8)\n"
        "    print('COMPUTED')\n"
        "except NotImplementedError:\n"
        "    print('RAISED')\n"
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT).stdout
    assert "RAISED" in out and "COMPUTED" not in out


def test_only_the_checkers_touch_the_oracle():
    """oracle/ is test infrastructure: nothing in the product package, in tools/ or in bench.py outside its
    cpu_baseline leg may import it; importing the product does not pull it in either."""
    import subprocess
    import sys

    offenders = []
    for top in ("bitsandbytes_amd", "tools", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".sh")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|oracle/|libbnb4_oracle", text, flags=re.M):
                        offenders.append(os.path.relpath(os.path.join(dirpath, f), ROOT))
    assert not offenders, f"oracle referenced outside tests/, smoke() and bench.py's cpu_baseline: {offenders}"
    bench_src = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle import|import oracle", bench_src)]
    assert len(uses) == 1 and bench_src.rfind("def ", 0, uses[0]) == bench_src.find("def cpu_baseline"), \
        "bench.py may import the oracle only inside cpu_baseline()"
    out = subprocess.run([sys.executable, "-c", "import sys, bitsandbytes_amd; print(any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules))"],
                         capture_output=True, text=True, cwd=ROOT).stdout
    assert out.strip() == "False"


def _device_kernel_metadata():
    """(name, private_segment_fixed_size, vgpr_count) of every kernel in the library's gfx950 code objects,
    read from the HSA metadata notes (no GPU needed)."""
    import re
    import subprocess
    import tempfile
    from pathlib import Path

    llvm = Path("/opt/rocm/lib/llvm/bin")
    tools = [llvm / "llvm-objcopy", llvm / "clang-offload-bundler", llvm / "llvm-readelf"]
    if not all(t.exists() for t in tools):
        pytest.skip("ROCm LLVM binutils not available")
    from bitsandbytes_amd.cextension import LIB_PATH

    out = []
    with tempfile.TemporaryDirectory() as td:
        fat = Path(td) / "fat.bin"
        subprocess.check_call([str(tools[0]), "-O", "binary", "--only-section=.hip_fatbin", str(LIB_PATH), str(fat)])
        blob = fat.read_bytes()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        assert starts, "no offload bundles found in .hip_fatbin"
        for i, s in enumerate(starts):
            piece = Path(td) / f"bundle{i}.bin"
            piece.write_bytes(blob[s : starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = Path(td) / f"dev{i}.co"
            subprocess.check_call([str(tools[1]), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   f"--input={piece}", f"--output={co}"], stderr=subprocess.DEVNULL)
            notes = subprocess.run([str(tools[2]), "--notes", str(co)], capture_output=True, text=True).stdout
            for block in notes.split("- .agpr_count")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block)
                scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
                vgpr = re.search(r"\.vgpr_count:\s+(\d+)", block)
                if name and scratch:
                    out.append((name.group(1), int(scratch.group(1)), int(vgpr.group(1)) if vgpr else -1))
    return out


def test_no_kernel_spills_to_scratch():
    """A register spill in these kernels is a performance bug with a correctness smell: a scratch reload is a
    vector-memory op, its wait is vmcnt(0), and that drains the DMA rings / weight streams the kernels keep
    in flight with counted waits. Every kernel of the library must have a zero private segment."""
    meta = _device_kernel_metadata()
    assert len(meta) > 100, f"expected the full kernel set, found {len(meta)}"
    spilled = [(n, s) for n, s, _ in meta if s != 0]
    assert not spilled, f"kernels using scratch: {spilled[:5]}"


def _device_disassembly(symbol_substring: str):
    """{kernel symbol: [instruction lines]} for the kernels of the library's gfx950 code objects whose name contains
    `symbol_substring` (llvm-objdump; no GPU needed)."""
    import re
    import subprocess
    import tempfile
    from pathlib import Path

    llvm = Path("/opt/rocm/lib/llvm/bin")
    tools = [llvm / "llvm-objcopy", llvm / "clang-offload-bundler", llvm / "llvm-objdump"]
    if not all(t.exists() for t in tools):
        pytest.skip("ROCm LLVM binutils not available")
    from bitsandbytes_amd.cextension import LIB_PATH

    kernels = {}
    with tempfile.TemporaryDirectory() as td:
        fat = Path(td) / "fat.bin"
        subprocess.check_call([str(tools[0]), "-O", "binary", "--only-section=.hip_fatbin", str(LIB_PATH), str(fat)])
        blob = fat.read_bytes()
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
        for i, s0 in enumerate(starts):
            piece = Path(td) / f"bundle{i}.bin"
            piece.write_bytes(blob[s0 : starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = Path(td) / f"dev{i}.co"
            subprocess.check_call([str(tools[1]), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   f"--input={piece}", f"--output={co}"], stderr=subprocess.DEVNULL)
            syms = subprocess.run([str(tools[2]), "-t", str(co)], capture_output=True, text=True).stdout
            if symbol_substring not in syms:
                continue
            text = subprocess.run([str(tools[2]), "-d", "--no-show-raw-insn", str(co)], capture_output=True, text=True).stdout
            cur = None
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1) if symbol_substring in m.group(1) and not m.group(1).endswith(".kd") else None
                    if cur:
                        kernels[cur] = []
                elif cur and line.strip():
                    kernels[cur].append(line.strip())
    return kernels


def test_rt_kernel_isa_keeps_its_loads_in_flight():
    """gemm4_mfma_rt_kernel leaves every wait to the compiler. Round 3 read its ISA and found three source patterns that made
    hipcc drain the in-order memory queue or serialise the loads: branches around loads (exec-masked rows), a
    run-time blocksize branch whose sides load into the same registers, scalar offsets that end up in VGPRs (every buffer load
    wrapped in a v_readfirstlane loop). This pins the repaired state on the shipped (branch-free, BL) instances:
      * no readfirstlane loop around a load anywhere in the kernel;
      * the first vector-memory wait behind the table barrier - the top of the chunk loop - is a COUNTED one that leaves at
        least the chunk's eight activation loads in flight (round 2's loop had vmcnt(1) / vmcnt(0) there)."""
    import re

    kernels = _device_disassembly("gemm4_mfma_rt_kernel")
    shipped = {k: v for k, v in kernels.items()
               if re.search(r"rt_kernelI\w+?Li\dELi\d+ELb[01]ELb[01]ELb1ELb[01]E", k)}  # <T, MT, WAVES, NESTED, DIRECT, BL = true, BS64>
    assert len(shipped) >= 28, f"expected the branch-free instances of the kernel, found {len(shipped)} of {len(kernels)}"
    for name, lines in shipped.items():
        ops = [ln.split()[0] for ln in lines]
        for i, op in enumerate(ops):
            if op == "v_readfirstlane_b32":
                window = ops[i : i + 10]
                assert not ("s_cbranch_execnz" in window and any(o.startswith("buffer_load") for o in window)), \
                    f"{name}: a buffer load inside a readfirstlane loop (scalar offset not uniform for the compiler)"
        first_barrier = ops.index("s_barrier")
        first_wait = next(ln for ln in lines[first_barrier:] if ln.startswith("s_waitcnt") and "vmcnt(" in ln)
        n = int(re.search(r"vmcnt\((\d+)\)", first_wait).group(1))
        assert n >= 8, f"{name}: the top of the chunk loop waits `{first_wait}` - the queue is drained there"


def test_stream_kernel_isa_waits_for_x_with_an_exact_count_and_never_drains_in_front_of_its_first_barrier():
    """gemv4_stream_kernel keeps its weight ring in flight across the first barrier: the wait for the activation image in front of
    it is `s_waitcnt vmcnt(NS x 2)` (the x pieces are older than the NS ring stages of two loads each), spelled in asm. Round 5's
    ring-late experiment (a second position of the ring under a wavefront-uniform run-time branch) showed how easily that
    is lost: in every instance that holds OTHER compiler-visible loads across such a branch (nested statistics, grouped, multi-phase,
    peer chain) hipcc answered with `vmcnt(0)` at the join - the whole ring drained in front of the barrier - invisible in the source
    and in every test of values. Pinned on every single-phase instance of the built library:
      * the last vector-memory wait in front of the first barrier is the counted one, vmcnt(2 NS);
      * plain instances (no nested statistics, no caller-supplied code table, not the peer chain, not grouped) have NO vmcnt(0)
        anywhere in front of that barrier, and exactly NS weight requests (one ring position)."""
    import re

    kernels = _device_disassembly("gemv4_stream_kernel")
    assert len(kernels) >= 150, len(kernels)
    checked = plain = 0
    for name, lines in kernels.items():
        m = re.search(r"stream_kernelI(\w+?)Li(\d)ELi(\d+)ELi(\d)ELi(\d+)E", name)  # <T, MB, WAVES, NS, FLAGS>
        assert m, name
        ns, flags = int(m.group(4)), int(m.group(5))
        if flags & 32:
            continue  # (multi-phase instances: the first barrier in program TEXT is the phase loop's, not the one behind the x wait)
        checked += 1
        ops = [ln.split()[0] for ln in lines]
        head = lines[: ops.index("s_barrier")]
        vm_waits = [int(re.search(r"vmcnt\((\d+)\)", ln).group(1)) for ln in head if ln.startswith("s_waitcnt") and "vmcnt(" in ln]
        if not flags & 64:  # (the peer-chain instances wait for their granule fetches, tag by tag, between the x wait and the barrier)
            assert vm_waits and vm_waits[-1] == 2 * ns, f"{name}: waits in front of the first barrier {vm_waits}, expected the last one to be vmcnt({2 * ns})"
        if not (flags & (1 | 2 | 16 | 64)):  # nested | caller's code table | grouped | peer
            plain += 1
            assert 0 not in vm_waits, f"{name}: vmcnt(0) in front of the first barrier - the weight ring is drained there ({vm_waits})"
            ring_loads = sum(1 for ln in head if ln.startswith("buffer_load_dwordx4"))
            assert ring_loads == ns, f"{name}: {ring_loads} weight requests in front of the barrier, expected one ring position ({ns})"
    assert checked >= 100 and plain >= 24, (checked, plain)


def test_sm_kernel_isa_counts_its_waits_and_keeps_the_ring_in_flight():
    """gemm4_mfma_sm_kernel mixes LDS-DMA spelled in asm (the activation staging) with compiler-visible buffer loads (the weight
    ring): the hand-written wait for the first fragments must leave EXACTLY the ring in flight - vmcnt(NS x LPS), LPS = two weight
    loads + the lane's scale (nested: + the second-level absmax), NS = 1 for the single-item instances, else 2 - and in the ring
    instances nothing between the table barrier and the first MFMA may drain the queue (the second stage stays in flight while the
    first item is decoded). Every instance stages its first chunk with (staged rows) / 2 DMA instructions in front of the barrier (+ one for
    the nested code table)."""
    import re

    kernels = _device_disassembly("gemm4_mfma_sm_kernel")
    assert len(kernels) >= 40, len(kernels)
    for name, lines in kernels.items():
        m = re.search(r"sm_kernelI(\w+?)Li(\d+)ELi(\d+)ELi(\d)ELb([01])ELb([01])ELi(\d)E", name)  # <T, ROWS, WAVES, TT, NESTED, SINGLE, ORDER>
        assert m, name
        rows, nested, single, order = int(m.group(2)), m.group(5) == "1", m.group(6) == "1", int(m.group(7))
        # (32-row instances: 16 virtual staged rows - two 16-row blocks of a 128-k chunk - and ONE 128-k weight load per stage)
        staged, halves = (16, 1) if rows == 32 else (rows, 2)
        ns, lps = (1 if single else 2), halves + (2 if nested else 1)
        ops = [ln.split()[0] for ln in lines]
        bar = ops.index("s_barrier")
        dmas = sum(1 for ln in lines[:bar] if ln.startswith("buffer_load_dwordx4") and " lds" in ln)
        assert dmas == staged // 2 + (1 if nested else 0), f"{name}: {dmas} DMA instructions in front of the table barrier"
        first_mfma = next(i for i, op in enumerate(ops) if op.startswith("v_mfma"))
        waits = [int(re.search(r"vmcnt\((\d+)\)", ln).group(1)) for ln in lines[bar:first_mfma] if ln.startswith("s_waitcnt") and "vmcnt(" in ln]
        assert waits, name
        if order == 0:
            assert waits[0] == ns * lps, f"{name}: the wait for the first fragments is vmcnt({waits[0]}), expected vmcnt({ns * lps})"
        if not single:
            assert 0 not in waits, f"{name}: vmcnt(0) between the table barrier and the first MFMA - the ring is drained there ({waits})"


def test_launch_plan_workspace_query_is_pure_host_logic():
    """bnb_mi355x_gemm_4bit_workspace_bytes runs the launch plans on the host (no GPU needed): for every BASELINE
    shape the split-K workspace is a whole number of fp32 slabs - [M, N] ones, or the K-quarter kernel's (whole 128-column x
    32 / 64-row workgroup tiles in the accumulator layout: gemm4_mfma_kq.hip) - bounded by one slab per 512 k of K
    (>= 2 chunks of 256 k per slice), zero where no MFMA split-K launch can happen, and the query is a pure
    function (same answer twice)."""
    from bitsandbytes_amd import cextension as ce

    assert ce.lib
    q = ce.lib.bnb_mi355x_gemm_4bit_workspace_bytes
    BF16 = 2
    shapes = [(4096, 4096), (8192, 8192), (11008, 4096), (4096, 11008), (1376, 4096), (512, 11008), (128, 512),
              (1000, 2816)]
    for N, K in shapes:
        for M in (1, 2, 3, 4, 8, 16, 17, 32, 33, 48, 49, 64, 100, 128):
            for kernel in (0, 2):
                w = q(kernel, BF16, M, N, K, 64)
                assert w == q(kernel, BF16, M, N, K, 64)
                slab = M * N * 4
                rows = 64 if M > 32 else 32
                slab_kq = -(-M // rows) * rows * -(-N // 128) * 128 * 4
                assert w % slab == 0 or w % slab_kq == 0, (M, N, K, w)
                ks = w // slab if w % slab == 0 else w // slab_kq
                assert ks == 0 or 2 <= ks <= max(2, K // 512), (M, N, K, ks)
                if kernel == 0 and M <= 2:
                    assert w == 0, "M <= 2 runs the dot kernel or the streaming MFMA kernel: no workspace"
        assert q(1, BF16, 64, N, K, 64) == 0, "explicit dot kernel never needs a workspace"
        assert q(0, 0, 64, N, K, 64) == 0, "fp32 activations never take the MFMA path"
        # blocksize 32 (round 5): on the MFMA route through the register-transposed kernel's BS32 instances from 5 rows on (3 on
        # >= 12 M weights) wherever K % 256 == 0, whole [M, N] slabs of workspace where that kernel splits K; blocksize 16 stays outside
        route = ce.lib.bnb_mi355x_gemm_4bit_route
        for M32 in (1, 2, 3, 4, 5, 16, 64, 200):
            want = int(K % 256 == 0 and (M32 >= 5 or (M32 >= 3 and N * K >= (12 << 20))))
            assert route(0, BF16, M32, N, K, 32) == want, (M32, N, K)
            w32 = q(0, BF16, M32, N, K, 32)
            assert (w32 == 0 if not want else w32 % (M32 * N * 4) == 0), (M32, N, K, w32)
        assert route(0, BF16, 64, N, K, 16) == 0 and q(0, BF16, 64, N, K, 16) == 0, "blocksize 16 is outside the MFMA kernels' preconditions"
        assert route(0, 0, 64, N, K, 32) == 0, "fp32 activations never take the MFMA path"
    # headline shape: single-launch plans (no finalize pass) for every batch up to 48 rows (the register-transposed kernel:
    # 256 column tiles fill the chip without K slices); large matrices with tall tiles split K
    assert q(0, BF16, 16, 4096, 4096, 64) == 0
    assert q(0, BF16, 48, 4096, 4096, 64) == 0
    assert q(0, BF16, 64, 4096, 4096, 64) > 0
    assert q(0, BF16, 64, 8192, 8192, 64) > 0
    assert q(0, BF16, 64, 4096, 4100, 64) == 0, "K % 256 != 0 falls back to the dot kernel"


def test_streaming_mfma_routing_and_grouped_route_are_pure_host_logic():
    """The route queries run on the host (256 CUs assumed without a device): the streaming MFMA kernel's measured table
    (csrc/gemm4_mfma.hip: sm_selected - profiles/r6_sm_v3_ab_full.txt, r6_sm_small_n_ab.txt) and what a group that shares x becomes
    (bnb_mi355x_gemm_4bit_grouped_route: 2 = one launch of the streaming MFMA kernel, 1 = one launch of the streaming kernel,
    0 = matrix by matrix; csrc/c_api.hip, profiles/r6_grouped_ab.txt). K % 64 == 0 shapes take no split-K workspace there."""
    import ctypes as ct

    from bitsandbytes_amd import cextension as ce

    lib = ce.lib
    BF16 = 2
    route, ws = lib.bnb_mi355x_gemm_4bit_route, lib.bnb_mi355x_gemm_4bit_workspace_bytes

    def grouped(heights, M, K, bs=64, dt=BF16):
        return lib.bnb_mi355x_gemm_4bit_grouped_route(dt, len(heights), (ct.c_int * len(heights))(*heights), M, K, bs)

    # single matrices: 2 ... 16 rows on >= 128-row matrices (MFMA route = 1), K tails to 64 rows, the measured exceptions
    for (M, N, K, want) in ((2, 4096, 4096, 1), (16, 4096, 4096, 1), (2, 1376, 4096, 1), (4, 512, 11008, 1), (2, 1024, 8192, 0), (2, 64, 4096, 0),
                            (64, 1376, 2752, 1), (128, 4096, 2752, 1), (129, 4096, 2752, 0), (12, 96, 2752, 0), (1, 4096, 4096, 0)):
        assert route(0, BF16, M, N, K, 64) == want, (M, N, K)
        if want and M <= 16:
            assert ws(0, BF16, M, N, K, 64) == 0, (M, N, K)
    assert route(0, 0, 4, 4096, 4096, 64) == 0, "fp32 activations never take the MFMA path"
    # groups
    qkvo = (4096,) * 4
    assert grouped(qkvo, 1, 4096) == 1
    assert all(grouped(qkvo, M, 4096) == 2 for M in (2, 3, 4, 8, 16, 17, 32, 48, 64))
    assert grouped(qkvo, 65, 4096) == 0 and grouped((512,) * 3, 64, 4096) == 2          # (17 ... 64 rows: <= 96 M weights to 32 rows, 72 M to 64)
    assert grouped((11008, 11008), 16, 4096) == 2 and grouped((11008, 11008), 32, 4096) == 2 and grouped((11008, 11008), 48, 4096) == 0
    assert grouped((14336, 14336), 32, 4096) == 0
    assert grouped((4096, 64), 4, 4096) == 0, "a member below the streaming MFMA kernel's range, another above the streaming kernel's: one by one"
    assert grouped((64, 64), 4, 4096) == 1 and grouped((64, 64), 5, 4096) == 0
    assert grouped((4096,) * 9, 2, 4096) == 0 and grouped(qkvo, 2, 4096 + 32) == 0 and grouped(qkvo, 2, 4096, bs=32) in (0, 1)
    assert grouped(qkvo, 4, 4096, dt=0) == 1 and grouped(qkvo, 5, 4096, dt=0) == 0, "fp32 activations: the streaming kernel's grouped launch to 4 rows"
    assert grouped(qkvo, 8, 4096) == grouped(qkvo, 8, 4096), "pure function"


def test_bench_sharded_chain_geometry():
    """bench.py --gpus N: every rank's shard of every layer of the N-sharded MLP chain holds the headline layer's 4096^2 weights
    at N = 1, 2, 4, 8; each layer's gathered y is the next layer's x; every shape lies inside the fused peer chain's
    preconditions (rows in fours, K <= 16384 for a consumed x); a world size that does not divide the model is refused."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    want = {1: (4096, 4096), 2: (4096, 8192), 4: (8192, 8192), 8: (8192, 16384)}
    for world, (H, Fd) in want.items():
        h, f, dims = bench.sharded_chain_dims(world, 4096, 4096)
        assert (h, f) == (H, Fd)
        (ns_up, k_up), (ns_dn, k_dn) = dims
        assert ns_up * k_up == 4096 * 4096 == ns_dn * k_dn
        assert world * ns_up == k_dn and world * ns_dn == k_up   # gathered y of one = x of the next
        assert ns_up % 4 == 0 and ns_dn % 4 == 0 and max(k_up, k_dn) <= 16384 and k_up % 256 == 0 and k_dn % 256 == 0
    assert bench.sharded_chain_dims(3, 4096, 4096) is None and bench.sharded_chain_dims(6, 4096, 4096) is None


def test_bench_contract_flags_and_algorithmic_bytes():
    """bench.py keeps the driver's contract (--gpus/--steps/--warmup, defaults that finish in minutes) and
    prices a step with SURVEY section 8(d)'s algorithmic-byte formula."""
    import subprocess
    import sys

    import bench

    assert bench.algorithmic_bytes(1, 4096, 4096, 64) == 9_453_568
    assert bench.algorithmic_bytes(64, 8192, 8192, 64) == 33_554_432 + 4_194_304 + 2 * 1_048_576
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.LAYERS * bench.algorithmic_bytes(1, 4096, 4096, 64) > 256 * 2**20
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_native_dispatch_routing_constants_match_python():
    """csrc/torch_dispatch.cpp repeats the routing thresholds of backends/hip.py (the two kernels of bitsandbytes::gemm_4bit must
    route identically; the GPU test compares their results, this one pins the constants on a host without a GPU)."""
    from bitsandbytes_amd.backends import hip

    src = open(os.path.join(ROOT, "bitsandbytes_amd", "csrc", "torch_dispatch.cpp")).read()

    def const(name):
        m = re.search(rf"constexpr\s+int64_t\s+{name}\s*=\s*(\d+)\s*;", src)
        assert m, name
        return int(m.group(1))

    assert const("kFusedMaxM") == hip.FUSED_MAX_M
    assert const("kFusedMaxMLongRows") == hip.FUSED_MAX_M_LONG_ROWS
    assert const("kFusedMaxMSquare") == hip.FUSED_MAX_M_SQUARE
    assert const("kFusedTallWeights") == hip.FUSED_TALL_WEIGHTS
    assert const("kStreamOnlyMaxM") == hip.STREAM_ONLY_MAX_M
    assert const("kSmTailMaxM") == hip.SM_TAIL_MAX_M and const("kSmMinRows") == hip.SM_MIN_ROWS
    assert const("kFusedMaxMBs32") == hip.FUSED_MAX_M_BS32
    assert const("kReferenceCustomMaxM") == hip._REFERENCE_CUSTOM_MAX_M
    # fp32 activations: fused up to 4 rows in both
    assert const("kFusedMaxMFp32") == 4
    import torch

    assert hip._gemm_4bit_route(torch.float32, 4, 64, 64, 64) == "fused" and hip._gemm_4bit_route(torch.float32, 5, 64, 64, 64) == "unfused"
    # K = 64 is not a multiple of 256: the MFMA kernels do not serve it, the streaming kernel's passes stop at STREAM_ONLY_MAX_M
    assert hip._gemm_4bit_route(torch.bfloat16, hip.STREAM_ONLY_MAX_M, 64, 64, 64) == "fused"
    assert hip._gemm_4bit_route(torch.bfloat16, hip.STREAM_ONLY_MAX_M + 1, 64, 64, 64) == "unfused"
    # (round 6: K % 64 == 0 rows on >= SM_MIN_ROWS-row matrices take the streaming MFMA kernel's row passes up to SM_TAIL_MAX_M rows)
    assert hip._gemm_4bit_route(torch.bfloat16, 128, 4096, 2752, 64) == "fused" and hip._gemm_4bit_route(torch.bfloat16, 129, 4096, 2752, 64) == "unfused"
    assert hip._gemm_4bit_route(torch.bfloat16, 128, 1376, 2752, 64) == "fused" and hip._gemm_4bit_route(torch.bfloat16, 129, 1376, 2752, 64) == "unfused"
    assert hip._gemm_4bit_route(torch.bfloat16, 17, 96, 2752, 64) == "unfused" and hip._gemm_4bit_route(torch.bfloat16, 12, 96, 2752, 64) == "fused"
    assert hip._gemm_4bit_route(torch.bfloat16, 17, 4096, 2752 + 32, 64) == "unfused"
    assert hip._gemm_4bit_route(torch.bfloat16, 64, 4096, 4096, 32, True) == "unfused" and hip._gemm_4bit_route(torch.bfloat16, 64, 4096, 4096, 32) == "fused"
    assert hip._gemm_4bit_route(torch.bfloat16, hip.FUSED_MAX_M, 8192, 8192, 64) == "fused"
    assert hip._gemm_4bit_route(torch.bfloat16, hip.FUSED_MAX_M + 1, 8192, 8192, 64) == "unfused"
    assert hip.fused_max_m(4096, 11008) == 1024 and hip.fused_max_m(4096, 4096) == 640 and hip.fused_max_m(11008, 4096) == 512
    assert hip.fused_max_m(8192, 28672) == 512  # (long rows, but beyond the measured 48 M weights)
    # the C++ function itself, compiled out of the source, against the Python twin on a grid of shapes
    m = re.search(r"int64_t fused_max_m\(int64_t N, int64_t K, int64_t blocksize, bool nested\) \{.*?\n\}\n", src, re.S)
    assert m, "fused_max_m not found in torch_dispatch.cpp"
    consts = "\n".join(re.findall(r"constexpr\s+int64_t\s+k\w+\s*=\s*\d+\s*;", src))
    prog = ("#include <cstdint>\n#include <cstdio>\n#include <initializer_list>\n" + consts + "\n" + m.group(0) +
            "int main() { const long Ns[] = {64, 1376, 2048, 3072, 4096, 5120, 8192, 11008, 14336, 28672};\n"
            "const long Ks[] = {64, 1088, 2048, 2752, 3072, 3584, 4096, 5120, 8192, 11008, 14336, 28672};\n"
            "for (long n : Ns) for (long k : Ks) for (long bs : {32L, 64L, 128L}) for (int ne = 0; ne < 2; ++ne)\n"
            "  std::printf(\"%ld %ld %ld %d %ld\\n\", n, k, bs, ne, (long)fused_max_m(n, k, bs, ne != 0));\nreturn 0; }\n")
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        cpp, exe = os.path.join(td, "f.cpp"), os.path.join(td, "f")
        open(cpp, "w").write(prog)
        subprocess.run(["g++", "-std=c++17", "-O0", cpp, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, stdout=subprocess.PIPE, text=True).stdout
    rows = [tuple(int(v) for v in line.split()) for line in out.strip().splitlines()]
    assert len(rows) == 10 * 12 * 3 * 2
    for n, k, bs, ne, got in rows:
        assert got == hip.fused_max_m(n, k, bs, bool(ne)), (n, k, bs, ne, got)
