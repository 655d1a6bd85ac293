"""``bitsandbytes.functional`` surface of the 4-bit path, backed by the MI355X kernels.

Mirrors (names, argument meaning, error behaviour, serialised format) the 4-bit slice of the
reference's ``bitsandbytes/functional.py``:

* :class:`QuantState` ............ reference functional.py:420-610
* :func:`create_dynamic_map` ..... reference functional.py:296-348
* :func:`get_4bit_type` .......... reference functional.py:772-859
* :func:`quantize_blockwise` / :func:`dequantize_blockwise` (8-bit, used for double quantisation)
  ................................ reference functional.py:613-769
* :func:`quantize_4bit` / :func:`dequantize_4bit` (+ ``*_nf4`` / ``*_fp4``) reference functional.py:862-1077
* :func:`gemv_4bit` .............. reference functional.py:1300-1334

All arithmetic goes through ``torch.ops.bitsandbytes.*`` whose HIP-device kernels live in
``backends/hip.py``. Nothing here computes on the CPU.
"""
from __future__ import annotations

import json
from typing import Any, Optional

import torch
from torch import Tensor

from . import _ops  # noqa: F401  (defines the op schemas)

name2qmap: dict[str, Tensor] = {}

_BLOCKSIZES_4BIT = (32, 64, 128, 256, 512, 1024, 2048, 4096)
_FLOAT_DTYPES = (torch.bfloat16, torch.float16, torch.float32)


# --------------------------------------------------------------------------------------------------
# small helpers for the packed (safetensors-friendly) form of QuantState
# --------------------------------------------------------------------------------------------------
def pack_dict_to_tensor(source_dict: dict) -> Tensor:
    """JSON -> uint8 tensor (format of reference bitsandbytes/utils.py:166-181)."""
    return torch.tensor(list(json.dumps(source_dict).encode("utf-8")), dtype=torch.uint8)


def unpack_tensor_to_dict(tensor_data: Tensor) -> dict:
    """uint8 tensor -> dict (reference bitsandbytes/utils.py:184-200)."""
    return json.loads(bytes(tensor_data.cpu().numpy()).decode("utf-8"))


def _dtype_name(dtype: torch.dtype) -> str:
    # NB: the reference writes str(dtype).strip("torch.") which happens to give the same result for
    # float16/bfloat16/float32 ("float16", "bfloat16", "float32"): the on-disk strings are identical.
    return str(dtype).split(".")[-1]


# --------------------------------------------------------------------------------------------------
# code tables
# --------------------------------------------------------------------------------------------------
def create_dynamic_map(signed: bool = True, max_exponent_bits: int = 7, total_bits: int = 8) -> Tensor:
    """The 'dynamic' 8-bit code: for exponent e in [0, max_exponent_bits) the interval [0.1, 1] is cut
    into 2**(e + fraction_bits) pieces whose mid-points, scaled by 10**(e - max_exponent_bits + 1),
    become code values; plus 0 and 1; sorted ascending. Produces exactly the tensor of
    reference functional.py:296-348 (verified bit-for-bit by tests/golden)."""
    non_sign_bits = total_bits - 1
    values: list[float] = []

    def add_band(exponent_index: int, pieces: int) -> None:
        edges = torch.linspace(0.1, 1, pieces + 1, dtype=torch.float32)
        mids = (edges[:-1] + edges[1:]) / 2.0
        scaled = (10 ** (-(max_exponent_bits - 1) + exponent_index)) * mids
        values.extend(scaled.tolist())
        if signed:
            values.extend((-scaled).tolist())

    last = 0
    for i in range(max_exponent_bits):
        frac_bits = i + non_sign_bits - max_exponent_bits
        add_band(i, int(2**frac_bits if signed else 2 ** (frac_bits + 1)))
        last = i
    extra = 2 ** (non_sign_bits - max_exponent_bits) - 1
    if extra > 0:
        add_band(last, extra)

    values.append(0.0)
    values.append(1.0)
    if len(values) != 2**total_bits:
        raise AssertionError(f"dynamic map has {len(values)} entries, expected {2**total_bits}")
    values.extend([0.0] * (256 - len(values)))
    values.sort()
    return torch.tensor(values, dtype=torch.float32)


def _pad_code_with_zeros(sorted_values: list) -> list:
    """A code with fewer than 256 levels is stored in 256 entries: the spare entries are zeros placed in the
    middle of the ascending list (between its lower and upper halves), as the reference's constructors do."""
    gap = 256 - len(sorted_values)
    half = len(sorted_values) // 2
    return sorted_values[:half] + [0.0] * gap + sorted_values[half:]


def create_linear_map(signed: bool = True, total_bits: int = 8, add_zero: bool = True) -> Tensor:
    """Uniform code on [-1, 1] (or [0, 1]): ``2**total_bits`` levels, one fewer for a signed code that has to
    contain an exact zero. Same tensor as reference functional.py:150-166."""
    levels = 2**total_bits
    if signed and (add_zero or total_bits < 8):
        levels -= 1  # an odd number of levels puts one of them exactly on zero
    values = torch.linspace(-1.0 if signed else 0.0, 1.0, levels)
    if levels == 256:
        return values
    return torch.tensor(_pad_code_with_zeros(values.tolist()), dtype=torch.float32)


def create_fp8_map(signed: bool = True, exponent_bits: int = 5, precision_bits: int = 2, total_bits: int = 8) -> Tensor:
    """Code of a miniature floating-point format: every (exponent field E, mantissa field m) pair, with
    bias ``2**(exponent_bits - 1)``, subnormals at E = 0, both signs when ``signed``; normalised to max 1 and
    zero-padded to 256 entries. Same tensor as reference functional.py:227-293 (including its exponent
    convention, under which larger E means a smaller magnitude - irrelevant after sorting and normalising)."""
    if exponent_bits + precision_bits != total_bits - (1 if signed else 0):
        raise AssertionError("exponent_bits + precision_bits (+ sign) must equal total_bits")
    bias = 2 ** (exponent_bits - 1)
    values: list[float] = []
    for e_field in range(2**exponent_bits):
        for m_field in range(2**precision_bits):
            fraction = m_field / 2**precision_bits
            if e_field == 0:
                magnitude = fraction * 2.0**-bias
            else:
                magnitude = (1.0 + fraction) * 2.0 ** -(e_field - bias - 1)
            values.append(magnitude)
            if signed:
                values.append(-magnitude)
    values.extend([0.0] * (256 - len(values)))
    values.sort()
    code = torch.tensor(values, dtype=torch.float32)
    return code / code.max()


def create_normal_map(offset: float = 0.9677083, use_extra_value: bool = True) -> Tensor:
    """NormalFloat code (the construction NF4's 16 constants come from, QLoRA appendix): quantiles of N(0, 1)
    at evenly spaced probabilities between ``offset`` and 1/2 on the positive side (8 of them, or 7 without the
    extra value) and 7 mirrored ones on the negative side, plus zero, normalised to [-1, 1] and stored sorted in
    256 entries. Same tensor as reference functional.py:169-224; needs scipy."""
    try:
        from scipy.stats import norm
    except ImportError as exc:  # same failure mode as the reference
        raise ImportError("Scipy is required for `create_normal_map`.") from exc

    n_pos = 8 if use_extra_value else 7
    positive = norm.ppf(torch.linspace(offset, 0.5, n_pos + 1)[:-1]).tolist()
    negative = (-norm.ppf(torch.linspace(offset, 0.5, 8)[:-1])).tolist()
    values = torch.tensor(positive + [0.0] * (256 - len(positive) - len(negative)) + negative, dtype=torch.float32)
    values = values.sort().values
    return values / values.max()


_NF4_VALUES = [
    -1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
    -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
    0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
    0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0,
]  # quantiles of N(0,1), QLoRA; index = 4-bit code
# sign(1) | exponent(2) | mantissa(1), listed by bit pattern; normalised by max|.| = 12 below
_FP4_VALUES = [0, 0.0625, 8.0, 12.0, 4.0, 6.0, 2.0, 3.0, -0, -0.0625, -8.0, -12.0, -4.0, -6.0, -2.0, -3.0]


_DEVICE_CONSTANTS: dict = {}


def _per_device_constant(key, device, make) -> Tensor:
    """A read-only constant tensor, made once on the host by ``make()`` and kept per device. Callers that hand it out copy it."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    k = (key, d)
    t = _DEVICE_CONSTANTS.get(k)
    if t is None:
        t = make().to(d)
        # (only a real tensor is kept: under FakeTensorMode / tracing `.to` hands back a fake or traced tensor, which every later
        # real call would otherwise receive from the cache)
        if type(t) is not torch.Tensor or torch.compiler.is_compiling():
            return t
        _DEVICE_CONSTANTS[k] = t
    return t


def get_4bit_type(typename: str, device=None, blocksize: int = 64) -> Tensor:
    """16-entry fp32 code table of a 4-bit type ('nf4' or 'fp4'); reference functional.py:772-859."""
    if device is None:
        device = "cuda"
    if typename == "nf4":
        data = _NF4_VALUES
    elif typename == "fp4":
        data = _FP4_VALUES
    else:
        raise NotImplementedError(f"Typename {typename} not supported")
    # The values are constants of the format: computed once (fp32 IEEE division, as the reference's t / t.abs().max()), kept per device,
    # and every call gets its own copy - one device-side copy instead of a synchronous host-to-device transfer and three launches
    # (that was a third of the host time of one quantize_4bit call: profiles/r5_host_overhead.txt)
    def make():
        t = torch.tensor(data, dtype=torch.float32)
        return t.div_(t.abs().max())

    return _per_device_constant(("4bit", typename), device, make).clone()


# --------------------------------------------------------------------------------------------------
# QuantState
# --------------------------------------------------------------------------------------------------
class QuantState:
    """Everything needed to undo a blockwise quantisation (reference functional.py:420-610)."""

    valid_quant_types = ("fp4", "nf4")
    valid_qs_type_keys = [f"bitsandbytes__{x}" for x in valid_quant_types]
    valid_qs_keys = [
        "absmax", "quant_map", "nested_absmax", "nested_quant_map", "quant_state", "quant_type", "blocksize",
        "dtype", "shape", "nested_blocksize", "nested_dtype", "nested_offset",
    ]

    def __init__(self, absmax, shape=None, code=None, blocksize=None, quant_type=None, dtype=None, offset=None,
                 state2=None):
        self.absmax = absmax
        self.shape = shape
        self.code = code
        self.dtype = dtype
        self.blocksize = blocksize
        self.quant_type = quant_type
        self.offset = offset
        self.state2 = state2
        self.nested = state2 is not None

    # FSDP walks state_dict FQNs with getattr(): "weight.quant_state.bitsandbytes__nf4" must resolve
    def __getattr__(self, name):
        if name.startswith("bitsandbytes__"):
            packed = self.as_dict(packed=True)
            key = "quant_state." + name
            if key in packed:
                return packed[key]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __getitem__(self, idx):
        """Legacy list view: [absmax, shape, dtype, blocksize, [offset, state2] | None, quant_type]."""
        nested_part = [self.offset, self.state2] if self.nested else None
        return [self.absmax, self.shape, self.dtype, self.blocksize, nested_part, self.quant_type][idx]

    @classmethod
    def from_dict(cls, qs_dict: dict[str, Any], device) -> "QuantState":
        """Inverse of :meth:`as_dict` for both the packed and the unpacked layout."""
        packed_keys = [k for k, v in qs_dict.items() if "quant_state" in k and isinstance(v, torch.Tensor)]
        if "quant_type" not in qs_dict:
            if not packed_keys:
                raise ValueError("Expected packed or unpacked quant_state items, found neither")
            if len(packed_keys) != 1 or packed_keys[0].split(".")[-1] not in cls.valid_qs_type_keys:
                raise ValueError(
                    f"There should be exactly one `quant_state` item with ending from {cls.valid_qs_type_keys}.\n"
                    f"Detected {packed_keys}."
                )
        if len(packed_keys) == 1:
            qs_dict.update(unpack_tensor_to_dict(qs_dict.pop(packed_keys[0])))

        qs_dict = {k.split(".")[-1]: v for k, v in qs_dict.items()}
        unknown = set(qs_dict) - set(cls.valid_qs_keys)
        if unknown:
            raise AssertionError(f"unexpected quant_state keys: {sorted(unknown)}")

        offset = state2 = None
        if "nested_absmax" in qs_dict:
            offset = torch.tensor(float(qs_dict["nested_offset"])).to(device)
            state2 = cls(
                absmax=qs_dict["nested_absmax"].to(device),
                blocksize=qs_dict["nested_blocksize"],
                code=qs_dict["nested_quant_map"].to(device),
                dtype=getattr(torch, qs_dict["nested_dtype"]),
            )
        shape = qs_dict["shape"]
        return cls(
            quant_type=qs_dict["quant_type"],
            absmax=qs_dict["absmax"].to(device),
            blocksize=qs_dict["blocksize"],
            code=qs_dict["quant_map"].to(device),
            dtype=getattr(torch, qs_dict["dtype"]),
            shape=torch.Size(shape) if shape is not None else None,
            offset=offset,
            state2=state2,
        )

    def as_dict(self, packed: bool = False) -> dict[str, Any]:
        """Components for ``state_dict``; ``packed=True`` folds the non-tensor items into one uint8
        tensor under ``quant_state.bitsandbytes__<type>`` (safetensors can only store tensors)."""
        d: dict[str, Any] = {
            "quant_type": self.quant_type,
            "absmax": self.absmax,
            "blocksize": self.blocksize,
            "quant_map": self.code,
            "dtype": _dtype_name(self.dtype),
            "shape": tuple(self.shape) if self.shape is not None else None,
        }
        if self.nested:
            d.update(
                nested_absmax=self.state2.absmax,
                nested_blocksize=self.state2.blocksize,
                nested_quant_map=self.state2.code.clone(),
                nested_dtype=_dtype_name(self.state2.dtype),
                nested_offset=self.offset.item(),
            )
        if not packed or self.quant_type is None:
            return d
        tensors = {k: v for k, v in d.items() if isinstance(v, torch.Tensor)}
        scalars = {k: v for k, v in d.items() if not isinstance(v, torch.Tensor)}
        tensors["quant_state.bitsandbytes__" + self.quant_type] = pack_dict_to_tensor(scalars)
        return tensors

    def to(self, device):
        self.code = self.code.to(device)
        self.absmax = self.absmax.to(device)
        if self.nested:
            self.offset = self.offset.to(device)
            self.state2.absmax = self.state2.absmax.to(device)
            self.state2.code = self.state2.code.to(device)

    def __eq__(self, other):
        if not isinstance(other, QuantState):
            return False

        def same_opt(a, b):
            if a is None or b is None:
                return a is b
            return a == b

        return (
            torch.allclose(self.absmax, other.absmax, atol=1e-6)
            and self.shape == other.shape
            and torch.allclose(self.code, other.code, atol=1e-6)
            and self.dtype == other.dtype
            and self.blocksize == other.blocksize
            and self.quant_type == other.quant_type
            and bool(same_opt(self.offset, other.offset))
            and bool(same_opt(self.state2, other.state2))
        )


# --------------------------------------------------------------------------------------------------
# 8-bit blockwise (double-quantisation helper)
# --------------------------------------------------------------------------------------------------
def _dynamic_map(device) -> Tensor:
    """The default 8-bit code on ``device``: the SAME tensor on every call (read-only by contract; states get copies)."""
    def make():
        if "dynamic" not in name2qmap:
            name2qmap["dynamic"] = create_dynamic_map()
        return name2qmap["dynamic"]

    return _per_device_constant("dynamic", device, make)


def quantize_blockwise(A: Tensor, code: Optional[Tensor] = None, absmax: Optional[Tensor] = None,
                       out: Optional[Tensor] = None, blocksize: int = 4096, nested: bool = False):
    """8-bit blockwise quantisation with a 256-entry code (default: the dynamic map).
    Returns ``(uint8 tensor shaped like A, QuantState)``; reference functional.py:613-686."""
    if blocksize <= 0:
        raise ValueError(f"blocksize must be positive, got {blocksize}")
    if A.dtype not in _FLOAT_DTYPES:
        raise ValueError(f"Blockwise quantization only supports 16/32-bit floats, but got {A.dtype}")
    if code is None:
        code = _dynamic_map(A.device)
    q, am = torch.ops.bitsandbytes.quantize_blockwise.default(A, code.to(A.device), blocksize)

    if nested:
        offset = am.mean()
        am -= offset
        q_am, state2 = quantize_blockwise(am, blocksize=blocksize, nested=False)
        state = QuantState(absmax=q_am, code=code.to(A.device, copy=True), blocksize=blocksize, dtype=A.dtype,
                           offset=offset, state2=state2)
    else:
        state = QuantState(absmax=am, code=code.to(A.device, copy=True), blocksize=blocksize, dtype=A.dtype)

    if out is not None:
        q = out.copy_(q)
    if absmax is not None:
        state.absmax = absmax.copy_(state.absmax)
    return q, state


def dequantize_blockwise(A: Tensor, quant_state: Optional[QuantState] = None, absmax: Optional[Tensor] = None,
                         code: Optional[Tensor] = None, out: Optional[Tensor] = None, blocksize: int = 4096,
                         nested: bool = False) -> Tensor:
    """Inverse of :func:`quantize_blockwise`; reference functional.py:689-769."""
    if quant_state is None and absmax is None:
        raise ValueError("dequantize_blockwise requires either quant_state or absmax")
    if A.dtype != torch.uint8:
        raise ValueError(f"A must be uint8, got {A.dtype}")
    if quant_state is None:
        if code is None:
            code = _dynamic_map(A.device)
        quant_state = QuantState(absmax=absmax, code=code, blocksize=blocksize, dtype=torch.float32)
    if quant_state.blocksize <= 0:
        raise ValueError(f"blocksize must be positive, got {quant_state.blocksize}")

    am = quant_state.absmax
    if quant_state.nested:
        am = dequantize_blockwise(quant_state.absmax, quant_state.state2)
        am += quant_state.offset
        if am.dtype != torch.float32:
            am = am.float()

    args = (A, am, quant_state.code.to(A.device), quant_state.blocksize, quant_state.dtype)
    if out is not None:
        torch.ops.bitsandbytes.dequantize_blockwise.out(*args, out=out)
        return out
    return torch.ops.bitsandbytes.dequantize_blockwise.default(*args)


# --------------------------------------------------------------------------------------------------
# 4-bit blockwise
# --------------------------------------------------------------------------------------------------
def quantize_4bit(A: Tensor, absmax: Optional[Tensor] = None, out: Optional[Tensor] = None, blocksize=None,
                  compress_statistics: bool = False, quant_type: str = "fp4",
                  quant_storage: torch.dtype = torch.uint8):
    """Blockwise NF4/FP4 quantisation. Returns ``(packed tensor [(n+1)//(2*itemsize), 1] of
    quant_storage, QuantState)``. With ``compress_statistics`` the fp32 absmax is itself quantised
    to 8 bits in blocks of 256 around its mean (double quantisation). Reference functional.py:884-969."""
    if blocksize is None:
        blocksize = 64
    if blocksize not in _BLOCKSIZES_4BIT:
        raise ValueError(f"invalid blocksize {blocksize}")
    if quant_type not in ("nf4", "fp4"):
        raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_type!r}")
    if A.dtype not in _FLOAT_DTYPES:
        raise ValueError(f"Blockwise 4bit quantization only supports 16/32-bit floats, but got {A.dtype}")

    code = get_4bit_type(quant_type, device=A.device)
    if compress_statistics and A.device.type == "cuda" and A.numel() > 0:
        # one operator for the encoder and the statistics (three launches behind one native call; the reference's sequence below is
        # four operator calls, 86 us of host time per 4096 x 4096 layer: profiles/r5_host_overhead.txt). The offset is the mean of the
        # fp32 absmax in a fixed summation order; the 8-bit codes are what quantize_blockwise gives on absmax - offset, bit for bit.
        code8 = _dynamic_map(A.device)
        packed, q_am, am2, offset = torch.ops.bitsandbytes_amd.quantize_4bit_nested.default(A, code8, blocksize, quant_type, quant_storage)
        state2 = QuantState(absmax=am2, code=code8.clone(), blocksize=256, dtype=torch.float32)
        state = QuantState(absmax=q_am, shape=A.shape, dtype=A.dtype, blocksize=blocksize, code=code,
                           quant_type=quant_type, offset=offset, state2=state2)
        if out is not None:
            packed = out.copy_(packed)
        if absmax is not None:
            state.absmax = absmax.copy_(state.absmax)
        return packed, state

    packed, am = torch.ops.bitsandbytes.quantize_4bit.default(A, blocksize, quant_type, quant_storage)

    if compress_statistics:
        offset = am.mean()
        q_am, state2 = quantize_blockwise(am - offset, blocksize=256)
        del am
        state = QuantState(absmax=q_am, shape=A.shape, dtype=A.dtype, blocksize=blocksize, code=code,
                           quant_type=quant_type, offset=offset, state2=state2)
    else:
        state = QuantState(absmax=am, shape=A.shape, dtype=A.dtype, blocksize=blocksize, code=code,
                           quant_type=quant_type)

    if out is not None:
        packed = out.copy_(packed)
    if absmax is not None:
        state.absmax = absmax.copy_(state.absmax)
    return packed, state


def quantize_fp4(A, absmax=None, out=None, blocksize=None, compress_statistics=False, quant_storage=torch.uint8):
    return quantize_4bit(A, absmax, out, blocksize, compress_statistics, "fp4", quant_storage)


def quantize_nf4(A, absmax=None, out=None, blocksize=None, compress_statistics=False, quant_storage=torch.uint8):
    return quantize_4bit(A, absmax, out, blocksize, compress_statistics, "nf4", quant_storage)


def dequantize_4bit(A: Tensor, quant_state: Optional[QuantState] = None, absmax: Optional[Tensor] = None,
                    out: Optional[Tensor] = None, blocksize: Optional[int] = None, quant_type: str = "fp4") -> Tensor:
    """Inverse of :func:`quantize_4bit`; reference functional.py:992-1077."""
    if blocksize is None:
        blocksize = 64
    if quant_state is None:
        if absmax is None or out is None:
            raise ValueError("dequantize_4bit requires both absmax and out when quant_state is not provided")
        quant_state = QuantState(absmax=absmax, shape=out.shape, dtype=out.dtype, blocksize=blocksize,
                                 quant_type=quant_type)
    else:
        absmax = quant_state.absmax

    if quant_state.blocksize not in _BLOCKSIZES_4BIT:
        raise ValueError(f"invalid blocksize {quant_state.blocksize}")
    if quant_state.quant_type not in ("nf4", "fp4"):
        raise ValueError(f"quant_type must be 'nf4' or 'fp4', got {quant_state.quant_type!r}")
    if quant_state.dtype not in _FLOAT_DTYPES:
        raise ValueError(f"Blockwise 4bit dequantization only supports 16/32-bit floats, but got {quant_state.dtype}")

    if quant_state.nested:
        s2 = quant_state.state2
        n_blocks = -(-int(torch.Size(quant_state.shape).numel()) // quant_state.blocksize)
        if (A.device.type == "cuda" and out is None and s2.blocksize == 256 and not s2.nested and s2.absmax.dtype == torch.float32
                and s2.code.dtype == torch.float32 and quant_state.absmax.dtype == torch.uint8
                and quant_state.offset.dtype == torch.float32 and A.numel() > 0
                # (what the sequence below tolerates and the one-launch kernel does not: statistics on another device, a code
                # array that is not exactly one byte per block)
                and quant_state.absmax.device == A.device and s2.absmax.device == A.device and s2.code.device == A.device
                and quant_state.offset.device == A.device and quant_state.absmax.numel() == n_blocks):
            # one operator / one launch: the fp32 absmax is reconstructed inside the dequantize kernel with the same two roundings
            # (code2[q] * absmax2, + offset) the sequence below performs
            res = torch.ops.bitsandbytes_amd.dequantize_4bit_nested.default(
                A, quant_state.absmax, s2.absmax, s2.code, quant_state.offset, quant_state.blocksize, quant_state.quant_type,
                quant_state.shape, quant_state.dtype)
            return res.t() if A.shape[0] == 1 else res
        absmax = dequantize_blockwise(quant_state.absmax, quant_state.state2)
        absmax += quant_state.offset
        if absmax.dtype != torch.float32:
            absmax = absmax.float()

    args = (A, absmax, quant_state.blocksize, quant_state.quant_type, quant_state.shape, quant_state.dtype)
    if out is not None:
        torch.ops.bitsandbytes.dequantize_4bit.out(*args, out=out)
    else:
        out = torch.ops.bitsandbytes.dequantize_4bit.default(*args)

    # BC: a weight handed over in the transposed [1, n/2] layout gets its result transposed back
    if A.shape[0] == 1:
        return out.t()
    return out


def dequantize_fp4(A, quant_state=None, absmax=None, out=None, blocksize=None):
    return dequantize_4bit(A, quant_state, absmax, out, blocksize, "fp4")


def dequantize_nf4(A, quant_state=None, absmax=None, out=None, blocksize=None):
    return dequantize_4bit(A, quant_state, absmax, out, blocksize, "nf4")


def gemv_4bit(A: Tensor, B: Tensor, out: Optional[Tensor] = None, transposed_A=False, transposed_B=False,
              state: Optional[QuantState] = None) -> Tensor:
    """Legacy single-row fused dequant + matvec (reference functional.py:1300-1334): nested absmax is
    expanded on the host side, then the ``gemv_4bit`` op runs with the state's own code table."""
    if state is None:
        raise ValueError("state cannot be None. gemv_4bit() requires the state from quantize_4bit()")
    absmax = state.absmax
    if state.nested:
        absmax = dequantize_blockwise(absmax, state.state2) + state.offset
    args = (A, B, state.shape, absmax, state.code, state.blocksize)
    if out is not None:
        torch.ops.bitsandbytes.gemv_4bit.out(*args, out=out)
        return out
    return torch.ops.bitsandbytes.gemv_4bit.default(*args)


def has_avx512bf16() -> bool:
    """Reference ``functional.has_avx512bf16`` asks its CPU library whether the AVX512-BF16 gemv path exists
    (functional.py:1670-1674). This package ships no CPU kernels, so the answer is always False."""
    return False
