"""``matmul_4bit`` and its autograd function — reference ``bitsandbytes/autograd/_functions.py:300-491``.

Forward is one ``bitsandbytes::gemm_4bit`` op call for every M (the op's MI355X kernel picks the
streaming dot kernel or an MFMA kernel); backward is ``grad_A = grad_out @ dequantize_4bit(B)`` - fused into one
launch on the HIP device for batches up to 128 rows (``bitsandbytes_amd::gemm_4bit_grad_input``,
``backends/hip.py: FUSED_BACKWARD_MAX_M``).
Double-quantised states pass their pieces straight into the op so the absmax reconstruction is
fused into the GEMM kernel.
"""
from __future__ import annotations

from typing import Optional
from warnings import warn

import torch

from .. import functional as F


def _is_compiling() -> bool:
    return torch.compiler.is_compiling()


def _gemm_4bit_from_state(A: torch.Tensor, B: torch.Tensor, quant_state: F.QuantState, bias):
    """Call the gemm_4bit op with the (possibly nested) statistics of ``quant_state``."""
    if not quant_state.nested:
        return torch.ops.bitsandbytes.gemm_4bit.default(
            A, B, quant_state.shape, quant_state.absmax, quant_state.blocksize, quant_state.quant_type, bias=bias
        )
    if quant_state.state2.blocksize != 256:
        raise NotImplementedError("nested quantization with state2.blocksize != 256 is not supported")
    return torch.ops.bitsandbytes.gemm_4bit.default(
        A,
        B,
        quant_state.shape,
        quant_state.state2.absmax,
        quant_state.blocksize,
        quant_state.quant_type,
        bias=bias,
        absmax_8bit=quant_state.absmax,
        absmax_code=quant_state.state2.code,
        absmax_offset=quant_state.offset,
    )


def _grad_input_from_state(grad_output: torch.Tensor, B: torch.Tensor, state: F.QuantState) -> torch.Tensor:
    fused = (
        grad_output.is_cuda
        and grad_output.dtype in (torch.float16, torch.bfloat16)
        and len(state.shape) == 2
        and grad_output.shape[-1] == state.shape[0]
        and (not state.nested or state.state2.blocksize == 256)
        # double backward (create_graph=True: gradient penalties, Hessian-vector products): the fused op has no autograd formula
        # of its own; dequantize + matmul - the reference's formulation - is differentiable in grad_output
        and not (torch.is_grad_enabled() and grad_output.requires_grad)
    )
    if not fused:
        return torch.matmul(grad_output, F.dequantize_4bit(B, state).to(grad_output.dtype))
    if state.nested:
        return torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default(
            grad_output, B, state.shape, state.state2.absmax, state.blocksize, state.quant_type,
            absmax_8bit=state.absmax, absmax_code=state.state2.code, absmax_offset=state.offset,
        )
    return torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default(
        grad_output, B, state.shape, state.absmax, state.blocksize, state.quant_type
    )


class MatMul4Bit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B, out=None, bias=None, quant_state: Optional[F.QuantState] = None):
        ctx.is_empty = A.numel() == 0
        if ctx.is_empty:
            ctx.A, ctx.B, ctx.bias = A, B, bias
            w_shape = quant_state.shape
            tail = w_shape[1:] if A.shape[-1] == w_shape[0] else w_shape[:1]
            return torch.empty(A.shape[:-1] + tail, dtype=A.dtype, device=A.device)

        B = B.view(-1, 1)  # canonical packed layout; quant_state.shape carries N and K
        output = _gemm_4bit_from_state(A, B, quant_state, bias)
        if out is not None:
            out.copy_(output)
            output = out

        ctx.state = quant_state
        ctx.dtype_A, ctx.dtype_B = A.dtype, B.dtype
        ctx.dtype_bias = None if bias is None else bias.dtype
        ctx.tensors = (None, B) if any(ctx.needs_input_grad[:2]) else (None, None)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        if ctx.is_empty:
            grad_bias = None if ctx.bias is None else torch.zeros_like(ctx.bias)
            return torch.zeros_like(ctx.A), torch.zeros_like(ctx.B), None, grad_bias, None

        need_A, _, _, need_bias, _ = ctx.needs_input_grad
        _, B = ctx.tensors
        grad_A = grad_bias = None
        if need_bias:
            grad_bias = grad_output.sum(0, dtype=ctx.dtype_bias)
        if need_A:
            # grad_out[M, N] @ dequantize(B)[N, K] = grad_A[M, K]. On the HIP device this is one fused launch for small and
            # medium batches (bitsandbytes_amd::gemm_4bit_grad_input: the weight tile is dequantized into LDS and never
            # written to HBM); the op itself falls back to dequantize + matmul for the rest - the reference's formulation.
            grad_A = _grad_input_from_state(grad_output, B, ctx.state)
        return grad_A, None, None, grad_bias, None


def matmul_4bit(
    A: torch.Tensor,
    B: torch.Tensor,
    quant_state: F.QuantState,
    out: Optional[torch.Tensor] = None,
    bias: Optional[torch.Tensor] = None,
):
    """``A @ dequant(B).T (+ bias)`` with ``B`` the packed 4-bit weight of a ``[N, K]`` matrix
    (reference autograd/_functions.py:407-491)."""
    if quant_state is None:
        raise ValueError("quant_state is required")
    if len(quant_state.shape) != 2:
        raise ValueError("matmul_4bit: quant_state.shape must be 2D [N, K]")

    B = B.view(-1, 1)
    K = A.shape[-1]

    # Legacy: weight quantised from a [K, N] tensor (A's inner dim matches shape[0], not shape[1]).
    if K == quant_state.shape[0] and K != quant_state.shape[1]:
        if not _is_compiling():
            warn(
                f"matmul_4bit: weight was quantized from a [K, N] tensor (quant_state.shape="
                f"{list(quant_state.shape)}). Re-quantize from the weight in [N, K] (out_features, in_features) "
                "orientation. This will be an error in a future version.",
                DeprecationWarning,
                stacklevel=2,
            )
        W = F.dequantize_4bit(B, quant_state).to(A.dtype)
        result = torch.nn.functional.linear(A, W.t(), bias)
        if out is not None:
            out.copy_(result)
            return out
        return result

    needs_grad = torch.is_grad_enabled() and (A.requires_grad or (bias is not None and bias.requires_grad))
    if needs_grad:
        return MatMul4Bit.apply(A, B, out, bias, quant_state)

    if A.numel() == 0:
        if out is not None:
            return out
        return torch.empty((*A.shape[:-1], quant_state.shape[0]), dtype=A.dtype, device=A.device)
    result = _gemm_4bit_from_state(A, B, quant_state, bias)
    if out is not None:
        out.copy_(result)
        return out
    return result


def matmul_4bit_grouped(A: torch.Tensor, weights, quant_states, biases=None, outs=None):
    """``[matmul_4bit(A, B_i, state_i, bias=bias_i) for i]`` for 4-bit weights that share their input - the Q/K/V
    projections of an attention block, the gate/up projections of an MLP. On MI355X a decode-sized batch (M <= 4) is
    ONE launch of the streaming kernel over the concatenated output rows (``bnb_mi355x_gemm_4bit_grouped``): one
    kernel boundary, one decode-table build and one activation copy per CU instead of one per matrix; 2 ... 16 rows: one launch
    of the streaming MFMA kernel; small groups of 17 ... 64 rows: the same in row passes. Outputs are bit-identical to the separate
    calls up to 16 rows (above: within the fused calls' tolerance); anything the grouped launch does not cover (autograd, mixed statistics
    formats, legacy [K, N] weights, CPU tensors) takes the separate calls.
    ``outs``: optional pre-allocated contiguous result tensors (slices of one communication buffer - the sharded block of
    ``parallel.py`` gathers a whole group with one collective); they are filled and returned.
    New functionality on top of the reference (which issues one gemm_4bit per Linear4bit, nn/modules.py:609-637)."""
    n = len(weights)
    biases = [None] * n if biases is None else list(biases)
    if len(quant_states) != n or len(biases) != n:
        raise ValueError("weights, quant_states and biases must have the same length")

    def separate():
        res = [matmul_4bit(A, w, s, bias=b) for w, s, b in zip(weights, quant_states, biases)]
        if outs is None:
            return res
        for o, r in zip(outs, res):
            o.copy_(r.reshape(o.shape))
        return list(outs)

    if n == 0:
        return []
    s0 = quant_states[0]
    K = A.shape[-1]
    groupable = (
        A.device.type == "cuda" and A.numel() > 0
        and not (torch.is_grad_enabled() and (A.requires_grad or any(b is not None and b.requires_grad for b in biases)))
        and not _is_compiling()
        and all(len(s.shape) == 2 and s.shape[1] == K and s.blocksize == s0.blocksize and s.quant_type == s0.quant_type
                and s.nested == s0.nested and (not s.nested or s.state2.blocksize == 256) for s in quant_states)
    )
    if not groupable:
        return separate()
    from ..backends import hip

    mats = []
    for w, s, b in zip(weights, quant_states, biases):
        if s.nested:
            mats.append((w.view(-1, 1), s.shape, s.state2.absmax, b, s.absmax, s.state2.code, s.offset))
        else:
            mats.append((w.view(-1, 1), s.shape, s.absmax, b, None, None, None))
    return hip.gemm_4bit_grouped(A, mats, s0.blocksize, s0.quant_type, outs=outs)
