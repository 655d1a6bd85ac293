"""TEST-ONLY: registers oracle-backed CPU kernels for the ``bitsandbytes::*`` 4-bit ops so that the
host-side logic (QuantState plumbing, Linear4bit, state-dict I/O, the N-sharded layer over gloo) can
be exercised without a GPU. The product package registers NO CPU kernels; this module lives under
tests/ and is imported only by tests."""
import torch

import bitsandbytes_amd  # noqa: F401  defines the op schemas
from oracle import oracle as O

_registered = False


def register():
    global _registered
    if _registered:
        return
    _registered = True
    rk = torch.library.register_kernel

    @rk("bitsandbytes::quantize_4bit", "cpu")
    def _(A, blocksize, quant_type, quant_storage):
        packed, absmax = O.quantize_4bit(A, blocksize, quant_type)
        if quant_storage != torch.uint8:
            packed = packed.reshape(-1).view(quant_storage).unsqueeze(1)
        return packed, absmax

    @rk("bitsandbytes::dequantize_4bit", "cpu")
    def _(A, absmax, blocksize, quant_type, shape, dtype):
        return O.dequantize_4bit(A, absmax, blocksize, quant_type, shape, dtype)

    @rk("bitsandbytes::quantize_blockwise", "cpu")
    def _(A, code, blocksize):
        return O.quantize_blockwise(A, code, blocksize)

    @rk("bitsandbytes::dequantize_blockwise", "cpu")
    def _(A, absmax, code, blocksize, dtype):
        return O.dequantize_blockwise(A, absmax, code, blocksize, dtype)

    @rk("bitsandbytes::gemm_4bit", "cpu")
    def _(A, B, shapeB, absmax, blocksize, quant_type, bias=None, absmax_8bit=None, absmax_code=None,
          absmax_offset=None):
        return O.gemm_4bit(A, B, shapeB, absmax, blocksize, quant_type, bias, absmax_8bit, absmax_code, absmax_offset)[0]

    @rk("bitsandbytes::gemv_4bit", "cpu")
    def _(A, B, shapeB, absmax, code, blocksize):
        qt = "fp4" if float(code[1]) > 0 else "nf4"
        return O.gemm_4bit(A, B, shapeB, absmax, blocksize, qt)[0]

    # the out= variants (reference _ops.py:178-212, 323-350, 377-406): same arithmetic, written into `out`
    @rk("bitsandbytes::dequantize_4bit.out", "cpu")
    def _(A, absmax, blocksize, quant_type, shape, dtype, out):
        out.copy_(O.dequantize_4bit(A, absmax, blocksize, quant_type, shape, dtype))

    @rk("bitsandbytes::dequantize_blockwise.out", "cpu")
    def _(A, absmax, code, blocksize, dtype, out):
        out.copy_(O.dequantize_blockwise(A, absmax, code, blocksize, dtype))

    @rk("bitsandbytes::gemv_4bit.out", "cpu")
    def _(A, B, shapeB, absmax, code, blocksize, out):
        qt = "fp4" if float(code[1]) > 0 else "nf4"
        out.copy_(O.gemm_4bit(A, B, shapeB, absmax, blocksize, qt)[0])

    @rk("bitsandbytes_amd::dequantize_4bit_rows", "cpu")
    def _(A, absmax, indices, row_len, blocksize, quant_type, dtype):
        num_rows = A.numel() * A.element_size() * 2 // row_len
        packed = A.contiguous().view(torch.uint8).view(num_rows, row_len // 2)
        rows = packed[indices.reshape(-1).long()].reshape(-1, 1)
        scales = absmax.view(num_rows, row_len // blocksize)[indices.reshape(-1).long()].reshape(-1)
        return O.dequantize_4bit(rows, scales, blocksize, quant_type, (*indices.shape, row_len), dtype)
