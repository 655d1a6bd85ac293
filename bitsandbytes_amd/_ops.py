"""``torch.ops.bitsandbytes.*`` schemas and fake (meta) kernels for the 4-bit path.

The schemas are string-identical to the reference's (``bitsandbytes/_ops.py:162-406``) so that this
package is a drop-in provider of the same operators: if the real ``bitsandbytes`` is already
imported the ops exist and we only add device kernels; otherwise we define them here.

Ops: quantize_4bit, dequantize_4bit(.out), gemm_4bit, gemv_4bit(.out), quantize_blockwise,
dequantize_blockwise(.out).
"""
from __future__ import annotations

from collections.abc import Sequence
from math import prod
from typing import Optional

import torch

_VALID_BLOCKSIZES_4BIT = (32, 64, 128, 256, 512, 1024, 2048, 4096)
_FLOAT_DTYPES = (torch.float16, torch.bfloat16, torch.float32)
_STORAGE_DTYPES = (torch.uint8, torch.bfloat16, torch.float16, torch.float32)

register_fake = torch.library.register_fake
_OVERRIDE_LIB = None


def register_kernel(op: str, device_types: str, func=None):
    """torch.library.register_kernel, except that a kernel somebody else (the reference package's own backends/cuda/ops.py,
    when this package is loaded through the reference's plug-in point on a box where the reference brought a ROCm build)
    registered first for the same dispatch key is REPLACED instead of raising."""

    def deco(fn):
        global _OVERRIDE_LIB
        try:
            torch.library.register_kernel(op, device_types, fn)
        except RuntimeError as exc:
            if "already" not in str(exc):
                raise
            ns, name = op.split("::")
            if _OVERRIDE_LIB is None:
                _OVERRIDE_LIB = {}
            if ns not in _OVERRIDE_LIB:
                _OVERRIDE_LIB[ns] = torch.library.Library(ns, "IMPL")
            key = {"cuda": "CUDA", "cpu": "CPU"}[device_types]
            _OVERRIDE_LIB[ns].impl(name, fn, key, allow_override=True)
        return fn

    return deco(func) if func is not None else deco


def _op_exists(name: str) -> bool:
    ns, op = name.split("::")
    base = op.split(".")[0]
    try:
        packet = getattr(getattr(torch.ops, ns), base)
    except (AttributeError, RuntimeError):
        return False
    overload = op.split(".")[1] if "." in op else "default"
    return overload in packet.overloads()


def _define(name: str, schema: str) -> bool:
    """Define the op unless somebody (the reference package) already did. Returns True if we own it."""
    if _op_exists(name):
        return False
    torch.library.define(name, schema)
    return True


def _check_4bit_common(blocksize: int, quant_type: str) -> None:
    torch._check(blocksize in _VALID_BLOCKSIZES_4BIT, lambda: f"invalid blocksize {blocksize}")
    torch._check(quant_type in ("nf4", "fp4"), lambda: f"quant_type must be 'nf4' or 'fp4', got {quant_type!r}")


# ---------------------------------------------------------------------------------------------- quantize_4bit
if _define(
    "bitsandbytes::quantize_4bit",
    "(Tensor A, int blocksize, str quant_type, ScalarType quant_storage) -> (Tensor, Tensor)",
):

    @register_fake("bitsandbytes::quantize_4bit")
    def _(A: torch.Tensor, blocksize: int, quant_type: str, quant_storage: torch.dtype):
        _check_4bit_common(blocksize, quant_type)
        torch._check(
            A.dtype in _FLOAT_DTYPES,
            lambda: f"Blockwise 4bit quantization only supports 16/32-bit floats, but got {A.dtype}",
        )
        n = A.numel()
        absmax = torch.empty((-(n // -blocksize),), device=A.device, dtype=torch.float32)
        out = torch.empty(((n + 1) // (quant_storage.itemsize * 2), 1), device=A.device, dtype=quant_storage)
        return out, absmax


# ---------------------------------------------------------------------------------------------- dequantize_4bit
def _check_dequant_4bit(absmax, blocksize, quant_type, dtype):
    _check_4bit_common(blocksize, quant_type)
    torch._check(absmax.dtype == torch.float32, lambda: f"absmax must be float32, got {absmax.dtype}")
    torch._check(
        dtype in _FLOAT_DTYPES,
        lambda: f"Blockwise 4bit dequantization only supports 16/32-bit floats, but got {dtype}",
    )


if _define(
    "bitsandbytes::dequantize_4bit",
    "(Tensor A, Tensor absmax, int blocksize, str quant_type, int[] shape, ScalarType dtype) -> Tensor",
):

    @register_fake("bitsandbytes::dequantize_4bit")
    def _(A, absmax, blocksize: int, quant_type: str, shape: Sequence[int], dtype: torch.dtype):
        _check_dequant_4bit(absmax, blocksize, quant_type, dtype)
        return torch.empty(shape, dtype=dtype, device=A.device)


if _define(
    "bitsandbytes::dequantize_4bit.out",
    "(Tensor A, Tensor absmax, int blocksize, str quant_type, int[] shape, ScalarType dtype, Tensor! out) -> ()",
):

    @register_fake("bitsandbytes::dequantize_4bit.out")
    def _(A, absmax, blocksize: int, quant_type: str, shape: Sequence[int], dtype: torch.dtype, out: torch.Tensor):
        _check_dequant_4bit(absmax, blocksize, quant_type, dtype)
        torch._check(out.shape == shape, lambda: f"Expected out.shape == {shape}, got {out.shape}")
        torch._check(out.device == A.device, lambda: f"Expected out.device == {A.device}, got {out.device}")
        torch._check(out.dtype == dtype, lambda: f"Expected out.dtype == {dtype}, got {out.dtype}")


# ---------------------------------------------------------------------------------------------- gemm_4bit
if _define(
    "bitsandbytes::gemm_4bit",
    "(Tensor A, Tensor B, int[] shapeB, Tensor absmax, int blocksize, str quant_type, "
    "Tensor? bias=None, Tensor? absmax_8bit=None, Tensor? absmax_code=None, Tensor? absmax_offset=None) -> Tensor",
):

    @register_fake("bitsandbytes::gemm_4bit")
    def _(
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
        torch._check(len(shapeB) == 2, lambda: f"shapeB must be 2D [N, K], got {list(shapeB)}")
        torch._check(A.shape[-1] == shapeB[1], lambda: f"A inner dim ({A.shape[-1]}) must match shapeB ({shapeB[1]})")
        torch._check(A.dtype in _FLOAT_DTYPES, lambda: f"A must be float16, bfloat16, or float32, got {A.dtype}")
        torch._check(
            B.dtype in _STORAGE_DTYPES,
            lambda: f"B must be backed by storage of type uint8, bfloat16, float16, or float32, got {B.dtype}",
        )
        _check_4bit_common(blocksize, quant_type)
        torch._check(absmax.dtype == torch.float32, lambda: f"absmax must be float32, got {absmax.dtype}")
        if absmax_8bit is not None:
            torch._check(absmax_8bit.ndim == 1, lambda: f"absmax_8bit must be 1D, got {absmax_8bit.ndim}D")
            torch._check(absmax_8bit.dtype == torch.uint8, lambda: f"absmax_8bit must be uint8, got {absmax_8bit.dtype}")
            torch._check(absmax_code is not None, lambda: "absmax_code required when absmax_8bit is provided")
            torch._check(absmax_code.ndim == 1, lambda: f"absmax_code must be 1D, got {absmax_code.ndim}D")
            torch._check(
                absmax_code.shape[0] == 256, lambda: f"absmax_code must have 256 entries, got {absmax_code.shape[0]}"
            )
            torch._check(
                absmax_code.dtype == torch.float32, lambda: f"absmax_code must be float32, got {absmax_code.dtype}"
            )
            torch._check(absmax_offset is not None, lambda: "absmax_offset required when absmax_8bit is provided")
            torch._check(
                absmax_offset.ndim == 0, lambda: f"absmax_offset must be a scalar (0-dim), got {absmax_offset.ndim}D"
            )
            torch._check(
                absmax_offset.dtype == torch.float32,
                lambda: f"absmax_offset must be float32, got {absmax_offset.dtype}",
            )
        if bias is not None:
            torch._check(bias.ndim == 1, lambda: f"bias must be 1D, got {bias.ndim}D")
            torch._check(
                bias.shape[0] == shapeB[0], lambda: f"bias length ({bias.shape[0]}) must match N ({shapeB[0]})"
            )
            torch._check(bias.dtype == A.dtype, lambda: f"bias dtype ({bias.dtype}) must match A dtype ({A.dtype})")
        return torch.empty((*A.shape[:-1], shapeB[0]), dtype=A.dtype, device=A.device)


# ---------------------------------------------------------------------------------------------- gemv_4bit
def _check_gemv(A, B, blocksize):
    torch._check(blocksize in _VALID_BLOCKSIZES_4BIT, lambda: f"invalid blocksize {blocksize}")
    torch._check(A.dtype in _FLOAT_DTYPES, lambda: f"A must be float16, bfloat16, or float32, got {A.dtype}")
    torch._check(
        B.dtype in _STORAGE_DTYPES,
        lambda: f"B must be backed by storage of type uint8, bfloat16, float16, or float32, got {B.dtype}",
    )


if _define(
    "bitsandbytes::gemv_4bit",
    "(Tensor A, Tensor B, int[] shapeB, Tensor absmax, Tensor code, int blocksize) -> Tensor",
):

    @register_fake("bitsandbytes::gemv_4bit")
    def _(A, B, shapeB: Sequence[int], absmax, code, blocksize: int) -> torch.Tensor:
        _check_gemv(A, B, blocksize)
        return torch.empty((*A.shape[:-1], shapeB[0]), device=A.device, dtype=A.dtype)


if _define(
    "bitsandbytes::gemv_4bit.out",
    "(Tensor A, Tensor B, int[] shapeB, Tensor absmax, Tensor code, int blocksize, Tensor! out) -> ()",
):

    @register_fake("bitsandbytes::gemv_4bit.out")
    def _(A, B, shapeB: Sequence[int], absmax, code, blocksize: int, out: torch.Tensor) -> None:
        _check_gemv(A, B, blocksize)
        expected = (*A.shape[:-1], shapeB[0])
        torch._check(out.shape == expected, lambda: f"Expected out.shape == {expected}, got {out.shape}")
        torch._check(out.device == A.device, lambda: f"Expected out.device == {A.device}, got {out.device}")
        torch._check(out.dtype == A.dtype, lambda: f"Expected out.dtype == {A.dtype}, got {out.dtype}")


# ---------------------------------------------------------------------------------------------- 8-bit blockwise
if _define("bitsandbytes::quantize_blockwise", "(Tensor A, Tensor code, int blocksize) -> (Tensor, Tensor)"):

    @register_fake("bitsandbytes::quantize_blockwise")
    def _(A: torch.Tensor, code: torch.Tensor, blocksize: int):
        torch._check(blocksize > 0, lambda: f"blocksize must be positive, got {blocksize}")
        torch._check(
            A.dtype in _FLOAT_DTYPES, lambda: f"Blockwise quantization only supports 16/32-bit floats, but got {A.dtype}"
        )
        n = A.numel()
        absmax = torch.empty((-(n // -blocksize),), device=A.device, dtype=torch.float32)
        return torch.empty_like(A, dtype=torch.uint8), absmax


def _check_dequant_blockwise(A, blocksize, dtype):
    torch._check(blocksize > 0, lambda: f"blocksize must be positive, got {blocksize}")
    torch._check(A.dtype == torch.uint8, lambda: f"A must be uint8, got {A.dtype}")
    torch._check(
        dtype in _FLOAT_DTYPES, lambda: f"Blockwise dequantization only supports 16/32-bit floats, but got {dtype}"
    )


if _define(
    "bitsandbytes::dequantize_blockwise",
    "(Tensor A, Tensor absmax, Tensor code, int blocksize, ScalarType dtype) -> Tensor",
):

    @register_fake("bitsandbytes::dequantize_blockwise")
    def _(A, absmax, code, blocksize: int, dtype: torch.dtype) -> torch.Tensor:
        _check_dequant_blockwise(A, blocksize, dtype)
        return torch.empty_like(A, dtype=dtype)


if _define(
    "bitsandbytes::dequantize_blockwise.out",
    "(Tensor A, Tensor absmax, Tensor code, int blocksize, ScalarType dtype, Tensor! out) -> ()",
):

    @register_fake("bitsandbytes::dequantize_blockwise.out")
    def _(A, absmax, code, blocksize: int, dtype: torch.dtype, out: torch.Tensor):
        _check_dequant_blockwise(A, blocksize, dtype)
        torch._check(out.shape == A.shape, lambda: f"Expected out.shape == {A.shape}, got {out.shape}")
        torch._check(out.device == A.device, lambda: f"Expected out.device == {A.device}, got {out.device}")
        torch._check(out.dtype == dtype, lambda: f"Expected out.dtype == {dtype}, got {out.dtype}")


__all__ = ["register_kernel", "register_fake", "prod"]


# ---------------------------------------------------------------------------------------------- dequantize_4bit_rows
# Not a reference op: the fused "gather rows, then dequantize" that Embedding4bit needs (the reference
# composes two F.embedding calls and dequantize_4bit, nn/modules.py:921-951). Lives in this package's own
# namespace so it can never collide with an operator the reference defines later.
torch.library.define(
    "bitsandbytes_amd::dequantize_4bit_rows",
    "(Tensor A, Tensor absmax, Tensor indices, int row_len, int blocksize, str quant_type, ScalarType dtype) -> Tensor",
)


@register_fake("bitsandbytes_amd::dequantize_4bit_rows")
def _(A, absmax, indices, row_len: int, blocksize: int, quant_type: str, dtype: torch.dtype):
    _check_4bit_common(blocksize, quant_type)
    torch._check(dtype in _FLOAT_DTYPES, lambda: f"dtype must be a 16/32-bit float, got {dtype}")
    torch._check(indices.dtype in (torch.int32, torch.int64), lambda: f"indices must be int32/int64, got {indices.dtype}")
    torch._check(row_len % blocksize == 0 and row_len % 8 == 0, lambda: "row_len must be a multiple of blocksize and of 8")
    return torch.empty((*indices.shape, row_len), dtype=dtype, device=A.device)


# ---------------------------------------------------------------------------------------------- quantize_4bit_nested
# Not a reference op: quantize_4bit(compress_statistics=True) as one operator - the 4-bit encoder, the mean of its fp32 absmax, the
# subtraction and the 8-bit blockwise encoder (blocksize 256) of the reference's functional.py:925-951, which there are four operator
# calls. Returns (packed, absmax_8bit, absmax2, offset). code8 is the 256-entry code of the second level (the dynamic map).
torch.library.define(
    "bitsandbytes_amd::quantize_4bit_nested",
    "(Tensor A, Tensor code8, int blocksize, str quant_type, ScalarType quant_storage) -> (Tensor, Tensor, Tensor, Tensor)",
)


@register_fake("bitsandbytes_amd::quantize_4bit_nested")
def _(A, code8, blocksize: int, quant_type: str, quant_storage: torch.dtype):
    _check_4bit_common(blocksize, quant_type)
    torch._check(A.dtype in _FLOAT_DTYPES, lambda: f"Blockwise 4bit quantization only supports 16/32-bit floats, but got {A.dtype}")
    torch._check(code8.dtype == torch.float32 and code8.numel() == 256, lambda: "code8 must be 256 float32 values")
    n = A.numel()
    blocks = -(n // -blocksize)
    return (
        torch.empty(((n + 1) // (quant_storage.itemsize * 2), 1), device=A.device, dtype=quant_storage),
        torch.empty((blocks,), device=A.device, dtype=torch.uint8),
        torch.empty((-(blocks // -256),), device=A.device, dtype=torch.float32),
        torch.empty((), device=A.device, dtype=torch.float32),
    )


# ---------------------------------------------------------------------------------------------- dequantize_4bit_nested
# Not a reference op: dequantize_4bit for double-quantised statistics as one operator / one launch (the reference reconstructs the
# fp32 absmax with two more operator calls first, functional.py:1002-1006). absmax2 / code8 / offset are state2.absmax, state2.code
# and state.offset; second-level blocksize 256.
torch.library.define(
    "bitsandbytes_amd::dequantize_4bit_nested",
    "(Tensor A, Tensor absmax_8bit, Tensor absmax2, Tensor code8, Tensor offset, int blocksize, str quant_type, int[] shape, "
    "ScalarType dtype) -> Tensor",
)


@register_fake("bitsandbytes_amd::dequantize_4bit_nested")
def _(A, absmax_8bit, absmax2, code8, offset, blocksize: int, quant_type: str, shape: Sequence[int], dtype: torch.dtype):
    _check_4bit_common(blocksize, quant_type)
    torch._check(dtype in _FLOAT_DTYPES, lambda: f"dtype must be a 16/32-bit float, got {dtype}")
    torch._check(absmax_8bit.dtype == torch.uint8, lambda: f"absmax_8bit must be uint8, got {absmax_8bit.dtype}")
    return torch.empty(tuple(shape), dtype=dtype, device=A.device)


# ---------------------------------------------------------------------------------------------- gemm_4bit_grad_input
# Not a reference op: the fused backward of gemm_4bit with respect to its activations,
#   grad_A[*, K] = grad_out[*, N] @ dequantize_4bit(B)[N, K]
# (the reference dequantizes the whole weight and calls a dense matmul, autograd/_functions.py:365-386). Own namespace,
# like dequantize_4bit_rows. Argument meaning of the weight side as in bitsandbytes::gemm_4bit.
torch.library.define(
    "bitsandbytes_amd::gemm_4bit_grad_input",
    "(Tensor grad_out, Tensor B, int[] shapeB, Tensor absmax, int blocksize, str quant_type, Tensor? absmax_8bit=None, "
    "Tensor? absmax_code=None, Tensor? absmax_offset=None) -> Tensor",
)


@register_fake("bitsandbytes_amd::gemm_4bit_grad_input")
def _(grad_out, B, shapeB: Sequence[int], absmax, blocksize: int, quant_type: str, absmax_8bit=None, absmax_code=None,
      absmax_offset=None):
    _check_4bit_common(blocksize, quant_type)
    torch._check(len(shapeB) == 2, lambda: "shapeB must be [N, K]")
    torch._check(grad_out.shape[-1] == shapeB[0], lambda: f"grad_out inner dim ({grad_out.shape[-1]}) must equal N ({shapeB[0]})")
    return torch.empty((*grad_out.shape[:-1], shapeB[1]), dtype=grad_out.dtype, device=grad_out.device)
