"""TEST INFRASTRUCTURE — Python face of ``oracle/bnb4_oracle.c`` (the CPU restatement of the
reference's 4-bit path) plus a thin ctypes face of ``oracle/_ref/libbitsandbytes_cpu.so`` (the
reference's own CPU library built by ``oracle/build_ref.sh``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
The product package ``bitsandbytes_amd`` never does.

All functions take / return CPU ``torch`` tensors so parity tests read like the reference's tests.
Citations (relative to /root/reference) are in ``bnb4_oracle.c``.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess
from typing import Optional

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(HERE, "libbnb4_oracle.so")
_REF_LIB_PATH = os.path.join(HERE, "_ref", "libbitsandbytes_cpu.so")

_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
_QT_CODE = {"fp4": 1, "nf4": 2}


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, <1 s). Returns the path of the shared object."""
    src = os.path.join(HERE, "bnb4_oracle.c")
    if force or not os.path.isfile(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "libbnb4_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> ct.CDLL:
    global _lib
    if _lib is None:
        _lib = ct.CDLL(build())
        _lib.oracle_bf16_to_f32.restype = ct.c_float
        _lib.oracle_f16_to_f32.restype = ct.c_float
        _lib.oracle_f32_to_bf16.restype = ct.c_uint16
        _lib.oracle_f32_to_f16.restype = ct.c_uint16
        _lib.oracle_f32_to_bf16.argtypes = [ct.c_float]
        _lib.oracle_f32_to_f16.argtypes = [ct.c_float]
    return _lib


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ct.c_void_p(t.data_ptr())


def _cpu_contig(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to("cpu").contiguous()


def get_4bit_code(quant_type: str) -> torch.Tensor:
    out = torch.empty(16, dtype=torch.float32)
    lib().oracle_get_4bit_code(ct.c_int(_QT_CODE[quant_type]), _p(out))
    return out


def quantize_4bit(A: torch.Tensor, blocksize: int = 64, quant_type: str = "nf4"):
    """-> (packed uint8 [(n+1)//2, 1], absmax fp32 [ceil(n/bs)]); reference default/ops.py:225-259."""
    A = _cpu_contig(A)
    n = A.numel()
    packed = torch.empty(((n + 1) // 2, 1), dtype=torch.uint8)
    absmax = torch.empty((-(n // -blocksize),), dtype=torch.float32)
    lib().oracle_quantize_4bit(
        _p(A), ct.c_int(_DTYPE_CODE[A.dtype]), ct.c_long(n), ct.c_int(blocksize), ct.c_int(_QT_CODE[quant_type]),
        _p(packed), _p(absmax),
    )
    return packed, absmax


def dequantize_4bit(packed: torch.Tensor, absmax: torch.Tensor, blocksize: int, quant_type: str, shape, dtype):
    packed = _cpu_contig(packed)
    if packed.dtype != torch.uint8:
        packed = packed.view(torch.uint8)
    absmax = _cpu_contig(absmax).float()
    out = torch.empty(tuple(shape), dtype=dtype)
    lib().oracle_dequantize_4bit(
        _p(packed), _p(absmax), ct.c_long(out.numel()), ct.c_int(blocksize), ct.c_int(_QT_CODE[quant_type]),
        ct.c_int(_DTYPE_CODE[dtype]), _p(out),
    )
    return out


def quantize_blockwise(A: torch.Tensor, code: torch.Tensor, blocksize: int, fma_mode: int = 1):
    """8-bit LUT-rule quantizer (csrc/cpu_ops.cpp:496-665). fma_mode: see bnb4_oracle.c."""
    A = _cpu_contig(A)
    code = _cpu_contig(code).float()
    n = A.numel()
    out = torch.empty(A.shape, dtype=torch.uint8)
    absmax = torch.empty((-(n // -blocksize),), dtype=torch.float32)
    lib().oracle_quantize_blockwise(
        _p(code), _p(A), ct.c_int(_DTYPE_CODE[A.dtype]), ct.c_long(n), ct.c_int(blocksize), _p(out), _p(absmax),
        ct.c_int(fma_mode),
    )
    return out, absmax


def dequantize_blockwise(A: torch.Tensor, absmax: torch.Tensor, code: torch.Tensor, blocksize: int, dtype):
    A = _cpu_contig(A)
    absmax = _cpu_contig(absmax).float()
    code = _cpu_contig(code).float()
    out = torch.empty(A.shape, dtype=dtype)
    lib().oracle_dequantize_blockwise(
        _p(code), _p(A), _p(absmax), ct.c_long(A.numel()), ct.c_int(blocksize), ct.c_int(_DTYPE_CODE[dtype]), _p(out)
    )
    return out


def gemm_4bit(
    A: torch.Tensor,
    B: torch.Tensor,
    shapeB,
    absmax: torch.Tensor,
    blocksize: int,
    quant_type: str,
    bias: Optional[torch.Tensor] = None,
    absmax_8bit: Optional[torch.Tensor] = None,
    absmax_code: Optional[torch.Tensor] = None,
    absmax_offset: Optional[torch.Tensor] = None,
):
    """-> (out in A.dtype, out_fp32). Reference default/ops.py:323-345 (see bnb4_oracle.c)."""
    A = _cpu_contig(A)
    B = _cpu_contig(B)
    if B.dtype != torch.uint8:
        B = B.view(torch.uint8)
    N, K = int(shapeB[0]), int(shapeB[1])
    M = A.numel() // K
    absmax = _cpu_contig(absmax).float()
    bias = None if bias is None else _cpu_contig(bias).to(A.dtype)
    a8 = None if absmax_8bit is None else _cpu_contig(absmax_8bit)
    ac = None if absmax_code is None else _cpu_contig(absmax_code).float()
    ao = None if absmax_offset is None else _cpu_contig(absmax_offset).float().reshape(1)
    out = torch.empty((*A.shape[:-1], N), dtype=A.dtype)
    out32 = torch.empty((*A.shape[:-1], N), dtype=torch.float32)
    lib().oracle_gemm_4bit(
        _p(A), ct.c_int(_DTYPE_CODE[A.dtype]), _p(B), _p(absmax), _p(a8), _p(ac), _p(ao), _p(bias),
        ct.c_int(M), ct.c_int(N), ct.c_int(K), ct.c_int(blocksize), ct.c_int(_QT_CODE[quant_type]), _p(out), _p(out32),
    )
    return out, out32


def gemv_4bit_f32acc(A: torch.Tensor, B: torch.Tensor, shapeB, absmax: torch.Tensor, blocksize: int, quant_type: str):
    """Scalar single-thread dequant+dot (the 'port' CPU timing baseline)."""
    A = _cpu_contig(A)
    B = _cpu_contig(B)
    N, K = int(shapeB[0]), int(shapeB[1])
    out = torch.empty(N, dtype=torch.float32)
    lib().oracle_gemv_4bit_f32acc(
        _p(A), ct.c_int(_DTYPE_CODE[A.dtype]), _p(B), _p(_cpu_contig(absmax).float()), ct.c_int(N), ct.c_int(K),
        ct.c_int(blocksize), ct.c_int(_QT_CODE[quant_type]), _p(out),
    )
    return out


# --------------------------------------------------------------------------------------------------
# The reference's own CPU library (oracle/_ref), called straight through its C ABI
# (csrc/pythonInterface.cpp:784-833). Present only after oracle/build_ref.sh ran where
# /root/reference exists; the built .so travels to the GPU box with the snapshot.
# --------------------------------------------------------------------------------------------------

_REF_REQUIRED_CPU_FLAGS = ("avx512f", "avx512bw", "avx512dq", "avx512vl", "avx2", "fma", "f16c")


def _host_cpu_flags() -> set:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def ref_lib_usable() -> bool:
    """The reference TUs are compiled with -mavx512{f,bw,dq,vl} for the whole file, so only load
    the binary on hosts that have those ISA extensions (otherwise SIGILL)."""
    return os.path.isfile(_REF_LIB_PATH) and all(f in _host_cpu_flags() for f in _REF_REQUIRED_CPU_FLAGS)


def ref_has_avx512bf16() -> bool:
    return ref_lib_usable() and "avx512_bf16" in _host_cpu_flags()


_ref = None


def ref_lib() -> ct.CDLL:
    global _ref
    if _ref is None:
        if not ref_lib_usable():
            raise RuntimeError("oracle/_ref/libbitsandbytes_cpu.so missing or host lacks AVX512")
        _ref = ct.CDLL(_REF_LIB_PATH)
    return _ref


def ref_dequantize_4bit(packed: torch.Tensor, absmax: torch.Tensor, blocksize: int, quant_type: str, shape, dtype):
    """cdequantize_blockwise_cpu_{nf4,fp4}_{fp32,bf16,fp16} (pythonInterface.cpp:784-818)."""
    packed = _cpu_contig(packed).view(torch.uint8)
    absmax = _cpu_contig(absmax).float()
    shape = tuple(shape) if len(shape) > 1 else (1, shape[0])
    m = 1
    for s in shape[:-1]:
        m *= s
    n = shape[-1]
    out = torch.empty(shape, dtype=dtype)
    name = {torch.float32: "fp32", torch.bfloat16: "bf16", torch.float16: "fp16"}[dtype]
    fn = getattr(ref_lib(), f"cdequantize_blockwise_cpu_{quant_type}_{name}")
    fn(_p(packed), _p(absmax), _p(out), ct.c_longlong(blocksize), ct.c_longlong(m), ct.c_longlong(n))
    return out


def ref_quantize_blockwise(A: torch.Tensor, code: torch.Tensor, blocksize: int):
    """cquantize_blockwise_cpu_{fp32,bf16,fp16} (pythonInterface.cpp:770-782)."""
    A = _cpu_contig(A)
    code = _cpu_contig(code).float()
    n = A.numel()
    out = torch.empty(A.shape, dtype=torch.uint8)
    absmax = torch.empty((-(n // -blocksize),), dtype=torch.float32)
    name = {torch.float32: "fp32", torch.bfloat16: "bf16", torch.float16: "fp16"}[A.dtype]
    fn = getattr(ref_lib(), f"cquantize_blockwise_cpu_{name}")
    fn(_p(code), _p(A), _p(absmax), _p(out), ct.c_longlong(blocksize), ct.c_longlong(n))
    return out, absmax


def ref_dequantize_blockwise(A: torch.Tensor, absmax: torch.Tensor, code: torch.Tensor, blocksize: int, dtype):
    A = _cpu_contig(A)
    out = torch.empty(A.shape, dtype=dtype)
    name = {torch.float32: "fp32", torch.bfloat16: "bf16", torch.float16: "fp16"}[dtype]
    fn = getattr(ref_lib(), f"cdequantize_blockwise_cpu_{name}")
    fn(_p(_cpu_contig(code).float()), _p(A), _p(_cpu_contig(absmax).float()), _p(out), ct.c_longlong(blocksize),
       ct.c_longlong(A.numel()))
    return out


# --------------------------------------------------------------------------------------------------
# The reference's fused CPU gemv (csrc/cpu_ops.cpp:687-915, AVX512-BF16), used as the timed CPU
# baseline ("kind": "reference") by bench.py. It needs the weight in the reference's CPU packing,
# produced in the reference by Python code that cannot travel to the GPU box
# (bitsandbytes/functional.py:1676-1727, _convert_weight_packed_for_cpu); restated here.
# --------------------------------------------------------------------------------------------------
def ref_pack_for_cpu_gemv(packed: torch.Tensor, absmax: torch.Tensor, N: int, K: int, blocksize: int):
    """[N*K/2] bytes (hi nibble = even element) -> the 32-row interleaved layout + bf16 absmax [K/bs, N]."""
    q = _cpu_contig(packed).view(torch.uint8).reshape(-1)
    nib = torch.empty(q.numel() * 2, dtype=torch.uint8)
    nib[0::2] = q >> 4
    nib[1::2] = q & 0xF
    assert N % 32 == 0 and K % 2 == 0
    w = nib.reshape(N // 32, 32, K // 2, 2).transpose(1, 2).contiguous().reshape(-1, 64)
    out = ((w[:, 32:] << 4) | w[:, :32]).reshape(N, K // 2).contiguous()
    am = _cpu_contig(absmax).float().reshape(N, K // blocksize).t().to(torch.bfloat16).contiguous()
    return out, am


def ref_fused_gemv(x: torch.Tensor, w_cpu_packed: torch.Tensor, absmax_bf16_t: torch.Tensor, N: int, K: int,
                   blocksize: int, quant_type: str = "nf4") -> torch.Tensor:
    """gemv_4bit_inference_cpu_{nf4,fp4}_bf16 (csrc/pythonInterface.cpp:820-833); x: [M, K] bf16."""
    x = _cpu_contig(x).to(torch.bfloat16).reshape(-1, K)
    M = x.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16)
    fn = getattr(ref_lib(), f"gemv_4bit_inference_cpu_{quant_type}_bf16")
    fn(ct.c_int64(M), ct.c_int64(N), ct.c_int64(K), _p(x), _p(w_cpu_packed), _p(absmax_bf16_t), _p(out),
       ct.c_int64(blocksize), ct.c_int64(x.stride(0)), ct.c_int64(out.stride(0)))
    return out
