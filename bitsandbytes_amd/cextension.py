"""Native-library loader for the MI355X backend.

Plays the role of the reference's ``bitsandbytes/cextension.py`` (:348-405) for this path, with one
deliberate difference: there is no CPU library and no deferred-error mock. The product path needs
``libbitsandbytes_mi355x.so`` (built by ``make -C bitsandbytes_amd/csrc`` / ``__graft_entry__.build()``);
if it is missing, ``lib`` is a stub whose every attribute access raises, so any attempt to compute
without the HIP library fails loudly instead of falling back to something else.

ctypes signatures mirror the reference's (``bitsandbytes/backends/cuda/ops.py:16-66``).
"""
from __future__ import annotations

import ctypes as ct
import logging
import os
from pathlib import Path

logger = logging.getLogger(__name__)

PACKAGE_DIR = Path(__file__).resolve().parent
LIB_NAME = "libbitsandbytes_mi355x.so"
LIB_PATH = Path(os.environ.get("BNB_MI355X_LIBRARY", PACKAGE_DIR / LIB_NAME))


class MissingNativeLibrary:
    """Stands in for the library when it cannot be loaded. Import succeeds (so that host-only code —
    QuantState, op schemas, fake kernels — stays usable), any native call raises."""

    def __init__(self, reason: str):
        self._reason = reason

    def __getattr__(self, name):
        raise RuntimeError(
            f"bitsandbytes_amd: native symbol '{name}' requested but {LIB_NAME} is not loaded ({self._reason}). "
            f"Build it with `make -C {PACKAGE_DIR / 'csrc'}` (needs hipcc, targets gfx950). "
            "There is no CPU or PyTorch fallback for the 4-bit kernels in this package."
        )

    def __bool__(self):
        return False


_VOID_P = ct.c_void_p
_I32 = ct.c_int32


def _declare(dll: ct.CDLL) -> None:
    def sig(names, argtypes, restype=None):
        for n in names:
            fn = getattr(dll, n)
            fn.argtypes = argtypes
            fn.restype = restype

    dts = ("fp32", "bf16", "fp16")
    qts = ("nf4", "fp4")
    # (code, A, absmax, out, blocksize, n)
    sig([f"cquantize_blockwise_{d}_{q}" for d in dts for q in qts] + [f"cquantize_blockwise_{d}" for d in dts],
        [_VOID_P] * 4 + [_I32, _I32])
    # (code, A, absmax, out, blocksize, n, stream)
    sig([f"cdequantize_blockwise_{d}_{q}" for d in dts for q in qts] + [f"cdequantize_blockwise_{d}" for d in dts],
        [_VOID_P] * 4 + [_I32, _I32, _VOID_P])
    # (A, B, absmax, absmax_8bit, absmax_code, absmax_offset, out, bias, M, N, K, blocksize, quant_type, stream)
    sig([f"cgemm_4bit_{d}" for d in dts], [_VOID_P] * 8 + [_I32] * 5 + [_VOID_P])
    # (m, n, k, A, B, absmax, code, out, lda, ldb, ldc, blocksize, stream)
    sig([f"cgemm_4bit_inference_naive_{d}" for d in dts], [_I32] * 3 + [_VOID_P] * 5 + [_I32] * 4 + [_VOID_P])
    sig(["get_context"], [], _VOID_P)
    sig(["cget_managed_ptr"], [ct.c_size_t], _VOID_P)
    # extensions
    sig(["bnb_mi355x_quantize_4bit"], [_VOID_P, _I32, _VOID_P, _VOID_P, _I32, ct.c_long, _I32, _VOID_P])
    sig(["bnb_mi355x_quantize_8bit"], [_VOID_P, _VOID_P, _I32, _VOID_P, _VOID_P, _I32, ct.c_long, _VOID_P])
    sig(["bnb_mi355x_quantize_4bit_nested"], [_VOID_P, _I32, ct.c_long, _I32, _I32] + [_VOID_P] * 7)
    sig(["bnb_mi355x_dequantize_4bit_nested"], [_I32] + [_VOID_P] * 6 + [_I32, ct.c_long, _I32, _VOID_P])
    sig(["bnb_mi355x_dequantize_4bit_rows"],
        [_I32, _VOID_P, _VOID_P, _VOID_P, _I32, _VOID_P, ct.c_long, ct.c_long, _I32, _I32, _I32, _VOID_P])
    sig(["bnb_mi355x_gemm_4bit"], [_I32, _I32] + [_VOID_P] * 9 + [_I32] * 5 + [_VOID_P, ct.c_size_t, _VOID_P])
    sig(["bnb_mi355x_gemm_4bit_workspace_bytes"], [_I32] * 6, ct.c_size_t)
    sig(["bnb_mi355x_gemm_4bit_route"], [_I32] * 6, _I32)
    sig(["bnb_mi355x_last_gemm_kernel"], [], _I32)
    # (dtype, A, count, B[], absmax[], absmax_8bit[], absmax_code[], absmax_offset[], out[], bias[], N[], M, K, blocksize, quant_type, stream)
    sig(["bnb_mi355x_gemm_4bit_grouped"], [_I32, _VOID_P, _I32] + [_VOID_P] * 8 + [_I32] * 4 + [_VOID_P])
    sig(["bnb_mi355x_gemm_4bit_grouped_route"], [_I32, _I32, _VOID_P, _I32, _I32, _I32], _I32)  # (dtype, count, N[], M, K, blocksize)
    # (dtype, grad_out, B, absmax, absmax_8bit, absmax_code, absmax_offset, grad_A, M, N, K, blocksize, quant_type, ws, ws_bytes, stream)
    sig(["bnb_mi355x_gemm_4bit_grad_input"], [_I32] + [_VOID_P] * 7 + [_I32] * 5 + [_VOID_P, ct.c_size_t, _VOID_P])
    sig(["bnb_mi355x_gemm_4bit_grad_input_workspace_bytes"], [_I32] * 3, ct.c_size_t)
    sig(["bnb_mi355x_gemm_4bit_grad_input_supported"], [_I32] * 5, _I32)
    sig(["bnb_mi355x_peer_buffer_bytes"], [_I32, ct.c_size_t], ct.c_size_t)
    sig(["bnb_mi355x_peer_alloc"], [ct.c_size_t], _VOID_P)
    sig(["bnb_mi355x_peer_free", "bnb_mi355x_peer_close"], [_VOID_P])
    sig(["bnb_mi355x_peer_export"], [_VOID_P, _VOID_P], _I32)
    sig(["bnb_mi355x_peer_open"], [_VOID_P], _VOID_P)
    # (bufs[], world, rank, src, out, bytes, max_bytes, stream)
    sig(["bnb_mi355x_peer_allgather"], [_VOID_P, _I32, _I32, _VOID_P, _VOID_P, ct.c_size_t, ct.c_size_t, _VOID_P])
    sig(["bnb_mi355x_peer_status"], [_VOID_P], _I32)
    sig(["bnb_mi355x_peer_chain_buffer_bytes"], [ct.c_long], ct.c_size_t)
    sig(["bnb_mi355x_peer_chain_alloc"], [ct.c_size_t, _I32], _VOID_P)
    # (world, ns, K, blocksize, mode, max_values, wg_limit)
    sig(["bnb_mi355x_gemv_4bit_peer_serves"], [_I32] * 5 + [ct.c_long, _I32], _I32)
    # (bufs[], epoch_word, world, rank, dtype, A, B, absmax, absmax_8bit, absmax_code, absmax_offset, bias, out_local, ns, K, blocksize,
    #  quant_type, mode, max_values, wg_limit, epoch_offset, stream)
    sig(["bnb_mi355x_gemv_4bit_peer"], [_VOID_P, _VOID_P, _I32, _I32, _I32] + [_VOID_P] * 8 + [_I32] * 5 + [ct.c_long, _I32, _I32, _VOID_P], _I32)
    # (bufs[], epoch_word, world, rank, dtype, out, nvalues, max_values, epoch_offset, stream)
    sig(["bnb_mi355x_peer_chain_read"], [_VOID_P, _VOID_P, _I32, _I32, _I32, _VOID_P, _I32, ct.c_long, _I32, _VOID_P])
    sig(["bnb_mi355x_set_stream_tuning"], [_I32] * 5)
    sig(["bnb_mi355x_set_tuning"], [_I32] * 4)
    sig(["bnb_mi355x_set_stamp_buffer"], [_VOID_P])
    sig(["bnb_mi355x_version"], [], ct.c_char_p)


def load_native_library():
    if not LIB_PATH.exists():
        return MissingNativeLibrary(f"{LIB_PATH} does not exist")
    try:
        dll = ct.CDLL(str(LIB_PATH))
        _declare(dll)
    except (OSError, AttributeError) as exc:  # missing libamdhip64, missing symbol, ...
        logger.error("bitsandbytes_amd: failed to load %s: %s", LIB_PATH, exc)
        return MissingNativeLibrary(str(exc))
    return dll


lib = load_native_library()

# every symbol include/bnb_mi355x.h declares; tests check that the loaded library exports all of them
EXPORTED_SYMBOLS = tuple(
    [f"cquantize_blockwise_{d}_{q}" for d in ("fp32", "bf16", "fp16") for q in ("nf4", "fp4")]
    + [f"cdequantize_blockwise_{d}_{q}" for d in ("fp32", "bf16", "fp16") for q in ("nf4", "fp4")]
    + [f"cquantize_blockwise_{d}" for d in ("fp32", "bf16", "fp16")]
    + [f"cdequantize_blockwise_{d}" for d in ("fp32", "bf16", "fp16")]
    + [f"cgemm_4bit_{d}" for d in ("fp32", "bf16", "fp16")]
    + [f"cgemm_4bit_inference_naive_{d}" for d in ("fp32", "bf16", "fp16")]
    + ["get_context", "cget_managed_ptr", "bnb_mi355x_quantize_4bit", "bnb_mi355x_quantize_8bit", "bnb_mi355x_quantize_4bit_nested", "bnb_mi355x_dequantize_4bit_nested", "bnb_mi355x_dequantize_4bit_rows",
       "bnb_mi355x_gemm_4bit", "bnb_mi355x_gemm_4bit_workspace_bytes", "bnb_mi355x_gemm_4bit_route", "bnb_mi355x_last_gemm_kernel",
       "bnb_mi355x_gemm_4bit_grouped", "bnb_mi355x_gemm_4bit_grouped_route",
       "bnb_mi355x_gemm_4bit_grad_input", "bnb_mi355x_gemm_4bit_grad_input_workspace_bytes", "bnb_mi355x_gemm_4bit_grad_input_supported",
       "bnb_mi355x_peer_buffer_bytes", "bnb_mi355x_peer_alloc", "bnb_mi355x_peer_free", "bnb_mi355x_peer_export", "bnb_mi355x_peer_open",
       "bnb_mi355x_peer_close", "bnb_mi355x_peer_allgather", "bnb_mi355x_peer_status",
       "bnb_mi355x_peer_chain_buffer_bytes", "bnb_mi355x_peer_chain_alloc", "bnb_mi355x_gemv_4bit_peer_serves", "bnb_mi355x_gemv_4bit_peer",
       "bnb_mi355x_peer_chain_read",
       "bnb_mi355x_set_stream_tuning", "bnb_mi355x_set_tuning", "bnb_mi355x_set_stamp_buffer", "bnb_mi355x_version"]
)
