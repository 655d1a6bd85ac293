"""Device kernels for the ``bitsandbytes::*`` 4-bit ops on MI355X — the ``"cuda"`` dispatch key is
what PyTorch-ROCm uses for HIP devices (reference ``bitsandbytes/__init__.py:26-36``).

Takes the place of the 4-bit part of the reference's ``bitsandbytes/backends/cuda/ops.py:299-982``:
same per-op glue (contiguity, output allocation, raw current stream, device guard), but the
dispatcher for ``gemm_4bit`` is MI355X-specific (:func:`_gemm_4bit_route`) instead of the reference's
per-arch heuristic tables (:583-843), and every native call goes to ``libbitsandbytes_mi355x.so``.

PyTorch is plumbing here (allocation, streams); all arithmetic of the path happens in the HIP
library. The one library GEMM that remains is the reference's own large-M / misaligned-K strategy:
dequantize once, then ``F.linear`` on hipBLASLt (reference :904-916).
"""
from __future__ import annotations

import functools
from collections.abc import Sequence
from math import prod
from typing import Optional
from warnings import warn

import torch

from .._ops import register_kernel
from ..cextension import lib

_DT_NAME = {torch.float32: "fp32", torch.float16: "fp16", torch.bfloat16: "bf16"}
_DT_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
_QT_CODE = {"fp4": 1, "nf4": 2}

# Largest M routed to the fused kernels; above it one dequantize + hipBLASLt GEMM moves fewer bytes per FLOP than re-streaming the
# packed weight per 64-row pass. Calibrated on MI355X, us per call, fused vs dequantize + hipBLASLt:
#  * M = 512 (profiles/r4_route_ab.txt): 4096^2 24 vs 41, 8192^2 89 vs 111, 11008 x 4096 70 vs 78, 4096 x 11008 59 vs 94 - 512 rows everywhere;
#  * 513 ... 1024 rows (round 5, profiles/r5_tall_small_ab.txt): the row passes of a call run side by side over grid.z, so on matrices
#    with LONG rows (K >= 2 N: 2048 x 8192, 1376 x 4096, 4096 x 11008) the fused call stays 8 - 30 % ahead up to 1024 rows, on square-ish
#    ones (K >= 0.7 N: 4096^2, 3072^2, 5120^2, 5120 x 3584) up to 640 rows, and on wide ones (8192 x 2048, 11008 x 4096) it loses from
#    576 on. Measured up to 45 M weights: the extended ranges apply up to FUSED_TALL_WEIGHTS, larger matrices keep 512 (8192^2 at
#    M = 1024: 178 vs 144).
FUSED_MAX_M = 512
FUSED_MAX_M_LONG_ROWS = 1024   # K >= 2 N
FUSED_MAX_M_SQUARE = 640       # 10 K >= 7 N
FUSED_TALL_WEIGHTS = 48 << 20
# Calls the MFMA kernels do not serve (K not a multiple of 256 - e.g. K = 2752, a 4-way shard of an 11008-wide projection -, or
# blocksize 32 with double-quantised statistics) run the streaming kernel in 4-row passes: ahead of dequantize + GEMM up to 12 rows,
# level at 16, 3 - 4 x behind at 64 (profiles/r5_tall_small_ab.txt, second table; until round 5 they were sent there up to 512 rows).
STREAM_ONLY_MAX_M = 16
# Round 6: rows that are not whole 256-k chunks (K % 64 == 0: K = 2752, 1344, 1088 ...) on matrices of >= SM_MIN_ROWS rows run the
# streaming MFMA kernel (csrc/gemm4_mfma_sm.hip), above 16 rows its 32-row instances in row passes over grid.y. Fused vs dequantize + GEMM,
# us (profiles/r6_sm_rows32_ab.txt): 4096 x 2752 M = 64 / 96 / 128 11.5 / 16.7 / 21.2 vs 30.2 / 36.8 / 36.7; 11008 x 1344 13.9 / 20.0 / 25.7 vs
# 31.6 / 29.9 / 30.2; 8192 x 2752 15.5 / 21.7 / 28.4 vs 43.4 / 44.1 / 44.5; 14336 x 1088 15.6 / 22.3 / 29.0 vs 30.9 / 32.6 / 28.8: ahead or level to 128 rows.
SM_TAIL_MAX_M = 128
SM_MIN_ROWS = 128  # (csrc/gemm4_mfma.hip: sm_selected / kSmMinRows; small matrices, profiles/r6_sm_small_n_ab.txt: 1376 x 2752 M = 16 / 64 4.5 / 8.1 us
# against 16.2 / 56.6 for the streaming kernel's passes)
# Blocksize 32 (plain statistics) runs the register-transposed kernel's BS32 instances, 64-row passes one after the other: 4096^2
# fused vs unfused 11.5 vs 28.9 us at 64 rows, 21.0 vs 37.5 at 128, 39.1 vs 33.1 at 256, 75 vs 40 at 512 (profiles/r5_tall_small_ab.txt, table 3).
FUSED_MAX_M_BS32 = 128
_REFERENCE_CUSTOM_MAX_M = 256  # reference backends/cuda/ops.py:816 (_gemm_4bit_custom_max_m on ROCm)


def _stream(t: torch.Tensor) -> int:
    return torch._C._cuda_getCurrentRawStream(t.device.index)


@functools.lru_cache(maxsize=None)
def _DEVICE_COUNT() -> int:  # (torch.cuda.device_count() costs ~1 us per call; the count cannot change under a live process)
    return torch.cuda.device_count()


class _device_of:
    """Make the tensor's device current for the duration of a native call (multi-GPU processes only;
    reference functional.py:80-88)."""

    def __init__(self, t: torch.Tensor):
        self.idx = t.device.index
        self.prev = None

    def __enter__(self):
        if _DEVICE_COUNT() > 1:
            self.prev = torch.cuda.current_device()
            if self.prev != self.idx:
                torch.cuda.set_device(self.idx)
            else:
                self.prev = None

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _check_c_int_count(n: int, what: str) -> None:
    """The reference ABI carries element counts as C ``int`` (csrc/pythonInterface.cpp:346-444;
    tests/test_functional.py:698-716 pins 2**31 - 1 as the largest supported size). ctypes would wrap a
    larger value silently, so refuse it here."""
    if n >= 2**31:
        raise ValueError(f"{what}: {n} elements exceed the C-ABI limit of 2**31 - 1")


# ------------------------------------------------------------------------------------------ quantize_4bit
@register_kernel("bitsandbytes::quantize_4bit", "cuda")
def _(A: torch.Tensor, blocksize: int, quant_type: str, quant_storage: torch.dtype):
    if blocksize not in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        raise ValueError(f"invalid blocksize {blocksize}")
    if quant_type not in _QT_CODE:
        raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type!r}")
    if A.dtype not in _DT_CODE:
        raise ValueError(f"Blockwise 4bit quantization only supports 16/32-bit floats, but got {A.dtype}")
    A = A.contiguous()
    n = A.numel()
    absmax = torch.empty((-(n // -blocksize),), device=A.device, dtype=torch.float32)
    out = torch.empty(((n + 1) // (quant_storage.itemsize * 2), 1), device=A.device, dtype=quant_storage)
    with _device_of(A):
        # stream-ordered variant of cquantize_blockwise_<T>_<q> (which is pinned to the NULL stream)
        lib.bnb_mi355x_quantize_4bit(
            A.data_ptr(), _DT_CODE[A.dtype], absmax.data_ptr(), out.data_ptr(), blocksize, n, _QT_CODE[quant_type],
            _stream(A),
        )
    return out, absmax


@register_kernel("bitsandbytes_amd::quantize_4bit_nested", "cuda")
def _(A: torch.Tensor, code8: torch.Tensor, blocksize: int, quant_type: str, quant_storage: torch.dtype):
    """quantize_4bit + the statistics of double quantisation behind ONE native call (three launches: 4-bit encoder, partial sums of
    the fp32 absmax, shifted 8-bit encoder). Reference bitsandbytes/functional.py:925-951 is four operator calls."""
    if blocksize not in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        raise ValueError(f"invalid blocksize {blocksize}")
    if quant_type not in _QT_CODE:
        raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type!r}")
    if A.dtype not in _DT_CODE:
        raise ValueError(f"Blockwise 4bit quantization only supports 16/32-bit floats, but got {A.dtype}")
    if code8.dtype != torch.float32 or code8.numel() != 256 or code8.device != A.device:
        raise ValueError("code8 must be 256 float32 values on A's device")
    A = A.contiguous()
    code8 = code8.contiguous()
    n = A.numel()
    if n == 0:
        raise ValueError("quantize_4bit_nested: empty input")
    blocks = -(n // -blocksize)
    out = torch.empty(((n + 1) // (quant_storage.itemsize * 2), 1), device=A.device, dtype=quant_storage)
    scratch = torch.empty((blocks + 1536,), device=A.device, dtype=torch.float32)  # include/bnb_mi355x.h: absmax, partial sums, encoder tables
    absmax_8bit = torch.empty((blocks,), device=A.device, dtype=torch.uint8)
    absmax2 = torch.empty((-(blocks // -256),), device=A.device, dtype=torch.float32)
    offset = torch.empty((), device=A.device, dtype=torch.float32)
    with _device_of(A):
        lib.bnb_mi355x_quantize_4bit_nested(
            A.data_ptr(), _DT_CODE[A.dtype], n, blocksize, _QT_CODE[quant_type], out.data_ptr(), scratch.data_ptr(),
            code8.data_ptr(), absmax_8bit.data_ptr(), absmax2.data_ptr(), offset.data_ptr(), _stream(A),
        )
    return out, absmax_8bit, absmax2, offset


# ------------------------------------------------------------------------------------------ dequantize_4bit
def _dequantize_4bit_impl(A, absmax, blocksize, quant_type, dtype, out):
    if dtype not in _DT_NAME:
        raise ValueError(f"Blockwise 4bit dequantization only supports 16/32-bit floats, but got {dtype}")
    if quant_type not in _QT_CODE:
        raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type!r}")
    if absmax.dtype != torch.float32:
        raise ValueError(f"absmax must be float32, got {absmax.dtype}")
    _check_c_int_count(out.numel(), "dequantize_4bit")
    A = A.contiguous()
    absmax = absmax.contiguous()
    fn = getattr(lib, f"cdequantize_blockwise_{_DT_NAME[dtype]}_{quant_type}")
    with _device_of(A):
        fn(None, A.data_ptr(), absmax.data_ptr(), out.data_ptr(), blocksize, out.numel(), _stream(A))


@register_kernel("bitsandbytes::dequantize_4bit", "cuda")
def _(A, absmax, blocksize: int, quant_type: str, shape: Sequence[int], dtype: torch.dtype) -> torch.Tensor:
    _check_c_int_count(prod(shape), "dequantize_4bit")
    out = torch.empty(tuple(shape), dtype=dtype, device=A.device)
    _dequantize_4bit_impl(A, absmax, blocksize, quant_type, dtype, out)
    return out


@register_kernel("bitsandbytes::dequantize_4bit.out", "cuda")
def _(A, absmax, blocksize: int, quant_type: str, shape: Sequence[int], dtype: torch.dtype, out: torch.Tensor):
    if tuple(out.shape) != tuple(shape):
        raise ValueError(f"Expected out.shape == {tuple(shape)}, got {tuple(out.shape)}")
    if out.dtype != dtype:
        raise ValueError(f"Expected out.dtype == {dtype}, got {out.dtype}")
    if not out.is_contiguous():
        raise ValueError("out must be contiguous")
    _dequantize_4bit_impl(A, absmax, blocksize, quant_type, dtype, out)


def _dequantize_4bit_nested_impl(A, absmax_8bit, absmax2, code8, offset, blocksize, quant_type, dtype, out):
    """One launch: the fp32 absmax of every block is reconstructed inside the dequantize kernel (csrc/dequantize4.hip, NESTED)."""
    if dtype not in _DT_CODE:
        raise ValueError(f"Blockwise 4bit dequantization only supports 16/32-bit floats, but got {dtype}")
    if quant_type not in _QT_CODE:
        raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type!r}")
    if blocksize not in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        raise ValueError(f"invalid blocksize {blocksize}")
    n = out.numel()
    blocks = -(n // -blocksize)
    if absmax_8bit.dtype != torch.uint8 or absmax_8bit.numel() != blocks:
        raise ValueError(f"absmax_8bit must hold {blocks} uint8 codes, got {tuple(absmax_8bit.shape)} {absmax_8bit.dtype}")
    if absmax2.dtype != torch.float32 or absmax2.numel() != -(blocks // -256):
        raise ValueError(f"absmax2 must hold {-(blocks // -256)} float32 values (second-level blocksize 256)")
    if code8.numel() != 256 or offset.numel() != 1:
        raise ValueError("code8 must hold 256 values and offset one value")
    # (a state rebuilt by QuantState.from_dict under another default dtype - model loaders set fp16 / bf16 - carries its offset in that
    # dtype; the host-side sequence promotes it in `absmax + offset`: the same value)
    code8 = code8.to(torch.float32)
    offset = offset.to(torch.float32)
    for t in (absmax_8bit, absmax2, code8, offset):
        if t.device != A.device:
            raise ValueError("all statistics must live on A's device")
    A = A.contiguous()
    with _device_of(A):
        lib.bnb_mi355x_dequantize_4bit_nested(
            _DT_CODE[dtype], A.data_ptr(), absmax_8bit.contiguous().data_ptr(), absmax2.contiguous().data_ptr(),
            code8.contiguous().data_ptr(), offset.data_ptr(), out.data_ptr(), blocksize, n, _QT_CODE[quant_type], _stream(A),
        )
    return out


def _nested_one_launch_ok(A, absmax_8bit, absmax2, code8, offset, blocksize, n) -> bool:
    """Whether the one-launch nested dequantize serves these statistics - what the three-operator sequence tolerates and that kernel
    does not: tensors on another device, a code array that is not one uint8 per block, missing code / offset."""
    if absmax_8bit is None or code8 is None or offset is None:
        return False
    blocks = -(n // -blocksize)
    return (absmax_8bit.dtype == torch.uint8 and absmax_8bit.numel() == blocks and absmax2.dtype == torch.float32
            and absmax2.numel() == -(blocks // -256) and code8.numel() == 256 and offset.numel() == 1
            and all(t.device == A.device for t in (absmax_8bit, absmax2, code8, offset)))


def _dequantize_4bit_any(A, absmax, blocksize, quant_type, dtype, out, absmax_8bit=None, absmax_code=None, absmax_offset=None):
    """Dequantize into ``out`` with plain or nested statistics: the one-launch nested kernel where it serves the call, else the
    host-side sequence of the reference (bitsandbytes/functional.py:1002-1006)."""
    if absmax_8bit is None:
        return _dequantize_4bit_impl(A, absmax, blocksize, quant_type, dtype, out)
    if _nested_one_launch_ok(A, absmax_8bit, absmax, absmax_code, absmax_offset, blocksize, out.numel()):
        return _dequantize_4bit_nested_impl(A, absmax_8bit, absmax, absmax_code, absmax_offset, blocksize, quant_type, dtype, out)
    if absmax_code is None or absmax_offset is None:
        raise RuntimeError("nested statistics need absmax_code and absmax_offset")
    scales = torch.ops.bitsandbytes.dequantize_blockwise.default(absmax_8bit.to(A.device), absmax.to(A.device), absmax_code.to(A.device), 256,
                                                                  torch.float32)
    scales = (scales + absmax_offset.to(A.device)).float()
    return _dequantize_4bit_impl(A, scales, blocksize, quant_type, dtype, out)


@register_kernel("bitsandbytes_amd::dequantize_4bit_nested", "cuda")
def _(A, absmax_8bit, absmax2, code8, offset, blocksize: int, quant_type: str, shape: Sequence[int], dtype: torch.dtype):
    out = torch.empty(tuple(shape), dtype=dtype, device=A.device)
    return _dequantize_4bit_nested_impl(A, absmax_8bit, absmax2, code8, offset, blocksize, quant_type, dtype, out)


@register_kernel("bitsandbytes_amd::dequantize_4bit_rows", "cuda")
def _(A, absmax, indices, row_len: int, blocksize: int, quant_type: str, dtype: torch.dtype) -> torch.Tensor:
    if dtype not in _DT_CODE:
        raise ValueError(f"Blockwise 4bit dequantization only supports 16/32-bit floats, but got {dtype}")
    if quant_type not in _QT_CODE:
        raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type!r}")
    if absmax.dtype != torch.float32:
        raise ValueError(f"absmax must be float32, got {absmax.dtype}")
    if indices.dtype not in (torch.int32, torch.int64):
        raise ValueError(f"indices must be int32 or int64, got {indices.dtype}")
    if row_len % blocksize != 0 or row_len % 8 != 0:
        raise ValueError(f"row_len ({row_len}) must be a multiple of blocksize ({blocksize}) and of 8")
    A = A.contiguous()
    absmax = absmax.contiguous()
    indices = indices.contiguous()
    num_rows = (A.numel() * A.element_size() * 2) // row_len
    out = torch.empty((*indices.shape, row_len), dtype=dtype, device=A.device)
    with _device_of(A):
        lib.bnb_mi355x_dequantize_4bit_rows(
            _DT_CODE[dtype], A.data_ptr(), absmax.data_ptr(), indices.data_ptr(), indices.element_size(), out.data_ptr(),
            indices.numel(), num_rows, row_len, blocksize, _QT_CODE[quant_type], _stream(A),
        )
    return out


# ------------------------------------------------------------------------------------------ 8-bit blockwise
@register_kernel("bitsandbytes::quantize_blockwise", "cuda")
def _(A: torch.Tensor, code: torch.Tensor, blocksize: int):
    if code.dtype != torch.float32:
        raise ValueError(f"code must be float32, got {code.dtype}")
    if blocksize not in (64, 128, 256, 512, 1024, 2048, 4096):
        raise ValueError(f"invalid blocksize {blocksize}")
    if A.dtype not in _DT_CODE:
        raise ValueError(f"Blockwise quantization only supports 16/32-bit floats, but got {A.dtype}")
    A = A.contiguous()
    code = code.contiguous()
    n = A.numel()
    absmax = torch.empty((-(n // -blocksize),), device=A.device, dtype=torch.float32)
    out = torch.empty_like(A, dtype=torch.uint8)
    with _device_of(A):
        lib.bnb_mi355x_quantize_8bit(
            code.data_ptr(), A.data_ptr(), _DT_CODE[A.dtype], absmax.data_ptr(), out.data_ptr(), blocksize, n, _stream(A)
        )
    return out, absmax


def _dequantize_blockwise_impl(A, absmax, code, blocksize, dtype, out):
    if dtype not in _DT_NAME:
        raise ValueError(f"Blockwise dequantization only supports 16/32-bit floats, but got {dtype}")
    if A.dtype != torch.uint8:
        raise ValueError(f"A must be uint8, got {A.dtype}")
    if blocksize <= 0 or (blocksize & (blocksize - 1)):
        raise ValueError(f"blocksize must be a positive power of two, got {blocksize}")
    _check_c_int_count(A.numel(), "dequantize_blockwise")
    A = A.contiguous()
    fn = getattr(lib, f"cdequantize_blockwise_{_DT_NAME[dtype]}")
    with _device_of(A):
        fn(code.contiguous().data_ptr(), A.data_ptr(), absmax.contiguous().data_ptr(), out.data_ptr(), blocksize,
           A.numel(), _stream(A))


@register_kernel("bitsandbytes::dequantize_blockwise", "cuda")
def _(A, absmax, code, blocksize: int, dtype: torch.dtype) -> torch.Tensor:
    out = torch.empty_like(A, dtype=dtype)
    _dequantize_blockwise_impl(A, absmax, code, blocksize, dtype, out)
    return out


@register_kernel("bitsandbytes::dequantize_blockwise.out", "cuda")
def _(A, absmax, code, blocksize: int, dtype: torch.dtype, out: torch.Tensor) -> None:
    if out.dtype != dtype:
        raise ValueError(f"Expected out.dtype == {dtype}, got {out.dtype}")
    if out.shape != A.shape:
        raise ValueError(f"Expected out.shape == {A.shape}, got {out.shape}")
    _dequantize_blockwise_impl(A, absmax, code, blocksize, dtype, out)


# ------------------------------------------------------------------------------------------ gemv_4bit (legacy)
def _gemv_4bit_impl(A, B, shapeB, absmax, code, blocksize, out):
    if blocksize not in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        raise ValueError(f"invalid blocksize {blocksize}")
    if A.dtype not in _DT_NAME:
        raise ValueError(f"A must be float16, bfloat16, or float32, got {A.dtype}")
    if A.numel() != A.shape[-1]:
        raise ValueError(f"gemv_4bit: A must be a single row vector, got shape {tuple(A.shape)}")
    N, K = int(shapeB[0]), int(shapeB[1])
    A = A.contiguous()
    B = B.contiguous()
    fn = getattr(lib, f"cgemm_4bit_inference_naive_{_DT_NAME[A.dtype]}")
    with _device_of(A):
        # (m = N, n = 1, k = K, ..., lda = N, ldb = (K+1)//2, ldc = N) as reference backends/cuda/ops.py:550-556
        fn(N, 1, K, A.data_ptr(), B.data_ptr(), absmax.contiguous().data_ptr(), code.contiguous().data_ptr(),
           out.data_ptr(), N, (K + 1) // 2, N, blocksize, _stream(A))


@register_kernel("bitsandbytes::gemv_4bit", "cuda")
def _(A, B, shapeB: Sequence[int], absmax, code, blocksize: int) -> torch.Tensor:
    out = torch.empty((*A.shape[:-1], shapeB[0]), device=A.device, dtype=A.dtype)
    _gemv_4bit_impl(A, B, shapeB, absmax, code, blocksize, out)
    return out


@register_kernel("bitsandbytes::gemv_4bit.out", "cuda")
def _(A, B, shapeB: Sequence[int], absmax, code, blocksize: int, out: torch.Tensor) -> None:
    expected = (*A.shape[:-1], shapeB[0])
    if tuple(out.shape) != tuple(expected):
        raise ValueError(f"Expected out.shape == {expected}, got {tuple(out.shape)}")
    if out.dtype != A.dtype:
        raise ValueError(f"Expected out.dtype == {A.dtype}, got {out.dtype}")
    _gemv_4bit_impl(A, B, shapeB, absmax, code, blocksize, out)


# ------------------------------------------------------------------------------------------ gemm_4bit
def _gemm_4bit_route(dtype: torch.dtype, M: int, N: int, K: int, blocksize: int, nested: bool = False) -> str:
    """MI355X routing: 'fused' (HIP dot / MFMA kernels, chosen inside the library by M) or 'unfused'
    (dequantize + hipBLASLt). Replaces reference backends/cuda/ops.py:814-843,921-962."""
    if K % blocksize != 0:
        # the reference only warns where its custom kernel could have run (M <= 256 on ROCm,
        # backends/cuda/ops.py:956-962): larger batches take dequantize + linear silently
        if M <= _REFERENCE_CUSTOM_MAX_M:
            warn(
                f"inner dimension ({K}) is not aligned for fast kernel with blocksize={blocksize}, "
                "falling back to slower implementation.",
                UserWarning,
            )
        return "unfused"
    if dtype == torch.float32:
        # fp32 activations: the streaming kernel (fp32 FMA decode, same code path as bf16/fp16) for decode-sized batches;
        # no fp32 MFMA path worth having above that (1/16 of the bf16 matrix rate)
        return "fused" if M <= 4 else "unfused"
    return "fused" if M <= fused_max_m(N, K, blocksize, nested) else "unfused"


def fused_max_m(N: int, K: int, blocksize: int = 64, nested: bool = False) -> int:
    """Largest batch (rows of A) the fused 16-bit kernels are used for on an N x K weight (csrc/torch_dispatch.cpp: fused_max_m)."""
    if K % 256 != 0 and K % 64 == 0 and blocksize >= 64 and N >= SM_MIN_ROWS:
        return SM_TAIL_MAX_M  # (the streaming MFMA kernel's row passes)
    if K % 256 != 0 or blocksize < 32 or (blocksize == 32 and nested):
        return STREAM_ONLY_MAX_M  # (the MFMA kernels' preconditions, csrc/gemm4_mfma.hip: gemm_4bit_mfma_supported)
    if blocksize == 32:
        return FUSED_MAX_M_BS32
    # (the extended ranges were measured on the K-quarter kernel: plain statistics at any blocksize >= 64, nested ones at blocksize 64
    # and K <= 16384 - csrc/gemm4_mfma_kq.hip: gemm_4bit_kq_serves; other nested calls run the producer/consumer kernel and keep 512)
    if N * K <= FUSED_TALL_WEIGHTS and blocksize >= 64 and (not nested or (blocksize == 64 and K <= 16384)):
        if K >= 2 * N:
            return FUSED_MAX_M_LONG_ROWS
        if 10 * K >= 7 * N:
            return FUSED_MAX_M_SQUARE
    return FUSED_MAX_M


def _gemm_4bit_fused(A, B, shapeB, absmax, blocksize, quant_type, bias, absmax_8bit, absmax_code, absmax_offset,
                     kernel: int = 0, out: Optional[torch.Tensor] = None, code16: Optional[torch.Tensor] = None):
    K = A.shape[-1]
    M = A.numel() // K
    N = int(shapeB[0])
    if K != shapeB[1]:
        raise RuntimeError(f"A inner dim ({K}) does not match weight ({shapeB[1]})")
    if absmax.dtype != torch.float32:
        raise RuntimeError(f"absmax must be float32, got {absmax.dtype}")
    if bias is not None:
        if bias.ndim != 1:
            raise RuntimeError(f"bias must be 1D, got {bias.ndim}D")
        if bias.dtype != A.dtype:
            raise RuntimeError(f"bias dtype ({bias.dtype}) must match A dtype ({A.dtype})")
        bias = bias.contiguous()
    if A.dtype not in _DT_CODE:
        raise RuntimeError(f"unsupported dtype {A.dtype}")
    A = A.contiguous()
    B = B.contiguous()
    if out is None:
        out = torch.empty((*A.shape[:-1], N), dtype=A.dtype, device=A.device)
    elif out.dtype != A.dtype or out.numel() != M * N or not out.is_contiguous():
        raise RuntimeError("out must be a contiguous tensor of A's dtype with M*N elements")
    offset32 = absmax_offset.to(dtype=torch.float32) if absmax_offset is not None else None
    # split-K scratch for the MFMA kernel comes from torch's caching allocator: stream-ordered and
    # legal under hipGraph capture (the library never has to allocate)
    ws = None
    with _device_of(A):
        # (M <= 2 always runs the streaming kernel, which needs no scratch: the decode hot path skips the size query; the
        # query sits inside the guard because the launch plan depends on the current device's CU count)
        ws_bytes = lib.bnb_mi355x_gemm_4bit_workspace_bytes(kernel, _DT_CODE[A.dtype], M, N, K, blocksize) if M > 2 else 0
        if ws_bytes:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=A.device)
        lib.bnb_mi355x_gemm_4bit(
            kernel, _DT_CODE[A.dtype], A.data_ptr(), B.data_ptr(), absmax.contiguous().data_ptr(),
            _ptr(absmax_8bit if absmax_8bit is None else absmax_8bit.contiguous()),
            _ptr(absmax_code if absmax_code is None else absmax_code.contiguous()), _ptr(offset32), _ptr(code16),
            out.data_ptr(), _ptr(bias), M, N, K, blocksize, _QT_CODE[quant_type], _ptr(ws), ws_bytes, _stream(A),
        )
    return out


# ------------------------------------------------------------------------------------------ gemm_4bit backward
# Largest batch the fused backward is used for: the kernel decodes the weights once per 64-row pass, so above two passes
# one dequantize_4bit + hipBLASLt GEMM is cheaper (measured on MI355X, profiles/r2_backward_bench.txt: 4096^2 M = 64 15.9 vs
# 31.6 us, M = 128 17.4 vs 40.6; 11008 x 4096 M = 64 22.8 vs 99 us; but 4096^2 M = 256 35 vs 32 us).
FUSED_BACKWARD_MAX_M = 128


def grad_input_fused_ok(dtype: torch.dtype, M: int, N: int, K: int, blocksize: int) -> bool:
    """Whether ``bitsandbytes_amd::gemm_4bit_grad_input`` runs the fused kernel for this problem."""
    if dtype not in (torch.float16, torch.bfloat16) or M < 1 or M > FUSED_BACKWARD_MAX_M:
        return False
    return bool(lib.bnb_mi355x_gemm_4bit_grad_input_supported(_DT_CODE[dtype], M, N, K, blocksize))


@register_kernel("bitsandbytes_amd::gemm_4bit_grad_input", "cuda")
def _(grad_out, B, shapeB: Sequence[int], absmax, blocksize: int, quant_type: str, absmax_8bit=None, absmax_code=None,
      absmax_offset=None):
    N, K = int(shapeB[0]), int(shapeB[1])
    if grad_out.shape[-1] != N:
        raise RuntimeError(f"grad_out inner dim ({grad_out.shape[-1]}) does not match weight rows ({N})")
    if quant_type not in _QT_CODE:
        raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type!r}")
    if absmax.dtype != torch.float32:
        raise RuntimeError(f"absmax must be float32, got {absmax.dtype}")
    M = grad_out.numel() // N if N else 0
    G = grad_out.contiguous()
    if M == 0 or not grad_input_fused_ok(G.dtype, M, N, K, blocksize) or G.data_ptr() % 16 or B.data_ptr() % 16:
        # unfused, like the reference: dequantize the weight, dense matmul (also the path for fp32 gradients and odd shapes)
        # (nested statistics: reconstructed inside the dequantize launch where that kernel serves them - the batches of QLoRA training
        # land here, M in the thousands - else the reference's host-side sequence)
        W = torch.empty((N, K), dtype=G.dtype, device=G.device)
        _dequantize_4bit_any(B, absmax, blocksize, quant_type, G.dtype, W, absmax_8bit, absmax_code, absmax_offset)
        return torch.matmul(G, W)
    out = torch.empty((*G.shape[:-1], K), dtype=G.dtype, device=G.device)
    B = B.contiguous()
    offset32 = absmax_offset.to(dtype=torch.float32) if absmax_offset is not None else None
    ws = None
    with _device_of(G):
        # (inside the guard: the launch plan depends on the current device's CU count)
        ws_bytes = lib.bnb_mi355x_gemm_4bit_grad_input_workspace_bytes(M, N, K)
        if ws_bytes:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=G.device)  # stream-ordered, legal under graph capture
        lib.bnb_mi355x_gemm_4bit_grad_input(
            _DT_CODE[G.dtype], G.data_ptr(), B.data_ptr(), absmax.contiguous().data_ptr(),
            _ptr(absmax_8bit if absmax_8bit is None else absmax_8bit.contiguous()),
            _ptr(absmax_code if absmax_code is None else absmax_code.contiguous()), _ptr(offset32), out.data_ptr(),
            M, N, K, blocksize, _QT_CODE[quant_type], _ptr(ws), ws_bytes, _stream(G),
        )
    return out


def gemm_4bit_grouped(A: torch.Tensor, mats, blocksize: int, quant_type: str, outs=None):
    """``[A @ dequant(B_i).T (+ bias_i) for i]`` for several packed weights that share the activations, in ONE launch
    (``bnb_mi355x_gemm_4bit_grouped``): of the streaming MFMA kernel at 2 ... 16 rows, of the streaming kernel at M <= 4 where
    that is the members' own route; other groups are issued matrix by matrix. ``mats``: sequence of ``(B, shapeB, absmax, bias, absmax_8bit, absmax_code,
    absmax_offset)`` with the argument meaning of the ``gemm_4bit`` op; all matrices share K, blocksize, quant_type
    and nested-ness. Results are bit-identical to separate ``gemm_4bit`` calls up to 16 rows; a group of 17 ... 64 rows that the library
    takes as one launch (row passes of the streaming MFMA kernel, include/bnb_mi355x.h) equals them within the fused calls' tolerance. ``outs``: optional pre-allocated contiguous
    ``[*, N_i]`` result tensors (e.g. slices of one communication buffer); they are returned."""
    import ctypes as ct

    K = A.shape[-1]
    M = A.numel() // K if K else 0
    if A.dtype not in _DT_CODE:
        raise RuntimeError(f"unsupported dtype {A.dtype}")
    if quant_type not in _QT_CODE:
        raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type!r}")
    count = len(mats)
    if count == 0:
        return []
    nested = mats[0][4] is not None
    one_launch = False
    if K % blocksize == 0 and 0 < count <= 8 and M > 0:
        ns = (ct.c_int * count)(*[int(m[1][0]) for m in mats])
        one_launch = lib.bnb_mi355x_gemm_4bit_grouped_route(_DT_CODE[A.dtype], count, ns, M, K, blocksize) != 0
    if not one_launch:
        # not a group the library serves with one launch (a member that the single-matrix op hands to another MFMA kernel: other
        # arithmetic, and faster there; more than 16 rows; more than 8 matrices): the single-matrix op (its own routing, its own
        # split-K workspace from torch's allocator)
        if outs is None:
            return [torch.ops.bitsandbytes.gemm_4bit.default(A, B, shapeB, absmax, blocksize, quant_type, bias, a8, ac, ao)
                    for (B, shapeB, absmax, bias, a8, ac, ao) in mats]
        if len(outs) != count:
            raise ValueError("outs must have one tensor per matrix")
        for o, (B, shapeB, absmax, bias, a8, ac, ao) in zip(outs, mats):
            N = int(shapeB[0])
            direct = (K % blocksize == 0 and A.dtype in _DT_CODE and o.dtype == A.dtype and o.is_contiguous() and o.numel() == M * N and o.device == A.device
                      and _gemm_4bit_route(A.dtype, M, N, K, blocksize, a8 is not None) == "fused")
            if direct:  # (the fused call writes the caller's tensor: no copy launch behind it)
                _gemm_4bit_fused(A, B, shapeB, absmax, blocksize, quant_type, bias, a8, ac, ao, out=o)
            else:
                o.copy_(torch.ops.bitsandbytes.gemm_4bit.default(A, B, shapeB, absmax, blocksize, quant_type, bias, a8, ac, ao).view(o.shape))
        return list(outs)
    A = A.contiguous()
    given = None if outs is None else list(outs)
    if given is not None and len(given) != count:
        raise ValueError("outs must have one tensor per matrix")
    keep, outs = [], []
    cols = {k: [] for k in ("B", "absmax", "a8", "ac", "ao", "out", "bias")}
    Ns = []
    for (B, shapeB, absmax, bias, a8, ac, ao) in mats:
        N = int(shapeB[0])
        if int(shapeB[1]) != K:
            raise RuntimeError(f"A inner dim ({K}) does not match weight ({shapeB[1]})")
        if (a8 is not None) != nested:
            raise RuntimeError("gemm_4bit_grouped: all matrices must be nested (double-quantised) or none")
        if absmax.dtype != torch.float32:
            raise RuntimeError(f"absmax must be float32, got {absmax.dtype}")
        if bias is not None:
            if bias.ndim != 1 or bias.dtype != A.dtype:
                raise RuntimeError("bias must be 1D and of A's dtype")
            bias = bias.contiguous()
        B = B.contiguous()
        absmax = absmax.contiguous()
        if given is None:
            out = torch.empty((*A.shape[:-1], N), dtype=A.dtype, device=A.device)
        else:
            out = given[len(outs)]
            if out.dtype != A.dtype or out.device != A.device or out.numel() != M * N or not out.is_contiguous():
                raise ValueError("outs[i] must be a contiguous [*, N_i] tensor of A's dtype on A's device")
        a8 = None if a8 is None else a8.contiguous()
        ac = None if ac is None else ac.contiguous()
        ao = None if ao is None else ao.to(dtype=torch.float32)
        keep += [B, absmax, bias, a8, ac, ao]
        for k, t in zip(("B", "absmax", "a8", "ac", "ao", "out", "bias"), (B, absmax, a8, ac, ao, out, bias)):
            cols[k].append(_ptr(t))
        Ns.append(N)
        outs.append(out)

    def arr(vals):
        return (ct.c_void_p * count)(*vals)

    with _device_of(A):
        lib.bnb_mi355x_gemm_4bit_grouped(
            _DT_CODE[A.dtype], A.data_ptr(), count, arr(cols["B"]), arr(cols["absmax"]),
            arr(cols["a8"]) if nested else None, arr(cols["ac"]) if nested else None, arr(cols["ao"]) if nested else None,
            arr(cols["out"]), arr(cols["bias"]), (ct.c_int32 * count)(*Ns), M, K, blocksize, _QT_CODE[quant_type], _stream(A),
        )
    return outs


def _gemm_4bit_unfused(A, B, shapeB, absmax, blocksize, quant_type, bias, absmax_8bit, absmax_code, absmax_offset):
    W = torch.empty(tuple(shapeB), dtype=A.dtype, device=A.device)
    # (nested statistics: reconstructed inside the dequantize launch - one launch, no fp32 absmax vector; was three)
    _dequantize_4bit_any(B, absmax, blocksize, quant_type, A.dtype, W, absmax_8bit, absmax_code, absmax_offset)
    return torch.nn.functional.linear(A, W, bias)


def _gemm_4bit_python_kernel(
    A: torch.Tensor,
    B: torch.Tensor,
    shapeB: Sequence[int],
    absmax: torch.Tensor,
    blocksize: int,
    quant_type: str,
    bias: Optional[torch.Tensor] = None,
    absmax_8bit: Optional[torch.Tensor] = None,
    absmax_code: Optional[torch.Tensor] = None,
    absmax_offset: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    K = A.shape[-1]
    M = A.numel() // K if K else 0
    route = _gemm_4bit_route(A.dtype, M, int(shapeB[0]), K, blocksize, absmax_8bit is not None)
    args = (A, B, shapeB, absmax, blocksize, quant_type, bias, absmax_8bit, absmax_code, absmax_offset)
    if route == "fused":
        return _gemm_4bit_fused(*args)
    return _gemm_4bit_unfused(*args)


def _load_native_dispatch() -> bool:
    """bitsandbytes::gemm_4bit's device kernel registered from C++ (csrc/torch_dispatch.cpp -> libbitsandbytes_mi355x_torch.so):
    the same glue as `_gemm_4bit_python_kernel` over the same C ABI without a Python frame and ctypes marshalling per call
    (eager Linear4bit.forward: profiles/r2_host_overhead.txt). Not used when another kernel library was selected with
    BNB_MI355X_LIBRARY (the dispatcher library is linked against the product library) or with BNB_MI355X_PYTHON_DISPATCH=1."""
    import os

    from ..cextension import LIB_NAME, LIB_PATH, PACKAGE_DIR

    if os.environ.get("BNB_MI355X_PYTHON_DISPATCH") == "1" or LIB_PATH != PACKAGE_DIR / LIB_NAME:
        return False
    path = PACKAGE_DIR / "libbitsandbytes_mi355x_torch.so"
    if not path.exists():
        warn(f"{path.name} is not built (make -C bitsandbytes_amd/csrc): gemm_4bit is dispatched through the slower Python glue",
             RuntimeWarning)
        return False
    try:
        torch.ops.load_library(str(path))
    except (OSError, RuntimeError) as exc:
        # built against another torch / C++ ABI, missing libpython symbols ...: the Python glue does the same work
        warn(f"{path.name} could not be loaded ({exc}): gemm_4bit is dispatched through the slower Python glue", RuntimeWarning)
        return False
    return True


NATIVE_DISPATCH = _load_native_dispatch()
if not NATIVE_DISPATCH:
    register_kernel("bitsandbytes::gemm_4bit", "cuda")(_gemm_4bit_python_kernel)


__all__ = ["FUSED_MAX_M", "NATIVE_DISPATCH", "fused_max_m", "gemm_4bit_grouped", "prod"]
