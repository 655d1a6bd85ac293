"""``Params4bit`` / ``Linear4bit`` — the module-level face of the 4-bit path
(reference ``bitsandbytes/nn/modules.py:213-716``).

Same contract as the reference: a ``Linear4bit`` holds its weight as a :class:`Params4bit`; the fp
weight is quantised lazily the first time the parameter is moved to a (HIP) device; ``forward`` is
``matmul_4bit(x, weight, bias, quant_state)``; the state-dict layout (``weight`` packed bytes plus
``weight.absmax``, ``weight.quant_map``, optional ``weight.nested_*`` and the JSON blob
``weight.quant_state.bitsandbytes__{nf4,fp4}``) is byte-compatible, so checkpoints written by either
implementation load in the other.
"""
from __future__ import annotations

import copy
import logging
from typing import Any, Optional

import torch
from torch import nn

from .. import functional as F
from ..autograd import matmul_4bit
from ..functional import QuantState

logger = logging.getLogger(__name__)

# QuantState attributes re-exported on the parameter so that FSDP's dotted-FQN traversal
# (getattr(weight, "absmax") ...) finds them. Implemented as properties, not __getattr__, which
# keeps torch.compile from graph-breaking on tensor subclasses (reference modules.py:261-339).
def _qs_property(attr_path: str, exported_name: str, none_ok: bool = False):
    def getter(self):
        obj = self.__dict__.get("quant_state")
        if obj is not None:
            for part in attr_path.split("."):
                obj = getattr(obj, part, None)
                if obj is None:
                    break
            if obj is not None or none_ok:
                return obj
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{exported_name}'")

    return property(getter)


def _rebuild_params4bit(state):
    new = Params4bit.__new__(Params4bit, data=state["data"], requires_grad=state["requires_grad"])
    new.__setstate__(state)
    return new


class Params4bit(torch.nn.Parameter):
    def __new__(
        cls,
        data: Optional[torch.Tensor] = None,
        requires_grad: bool = False,  # quantised weights are frozen by default
        quant_state: Optional[QuantState] = None,
        blocksize: Optional[int] = None,
        compress_statistics: bool = True,
        quant_type: str = "fp4",
        quant_storage: torch.dtype = torch.uint8,
        module: Optional["Linear4bit"] = None,
        bnb_quantized: bool = False,
        **kwargs,
    ) -> "Params4bit":
        if data is None:
            data = torch.empty(0)
        self = torch.Tensor._make_subclass(cls, data, requires_grad)
        self.blocksize = 64 if blocksize is None else blocksize
        self.compress_statistics = compress_statistics
        self.quant_type = quant_type
        self.quant_state = quant_state
        self.quant_storage = quant_storage
        self.bnb_quantized = bnb_quantized
        self.data = data
        self.module = module
        return self

    # ---- QuantState proxies (FSDP state-dict traversal)
    absmax = _qs_property("absmax", "absmax")
    code = _qs_property("code", "code")
    quant_map = _qs_property("code", "quant_map")
    offset = _qs_property("offset", "offset", none_ok=True)
    state2 = _qs_property("state2", "state2", none_ok=True)
    nested_absmax = _qs_property("state2.absmax", "nested_absmax")
    nested_blocksize = _qs_property("state2.blocksize", "nested_blocksize")
    nested_quant_map = _qs_property("state2.code", "nested_quant_map")
    nested_dtype = _qs_property("state2.dtype", "nested_dtype")
    nested_offset = _qs_property("offset", "nested_offset", none_ok=True)

    # ---- pickling / copying
    _STATE_FIELDS = ("blocksize", "compress_statistics", "quant_type", "quant_state", "quant_storage",
                     "bnb_quantized", "module")

    def __getstate__(self):
        state = self.__dict__.copy()
        state["data"] = self.data
        state["requires_grad"] = self.requires_grad
        return state

    def __setstate__(self, state):
        self.requires_grad = state["requires_grad"]
        for f in self._STATE_FIELDS:
            setattr(self, f, state[f])
        self.data = state["data"]

    def __reduce_ex__(self, proto):
        # torch.nn.Parameter's default reduce rebuilds a plain Parameter; keep the subclass
        return (_rebuild_params4bit, (self.__getstate__(),))

    def __deepcopy__(self, memo):
        new = type(self).__new__(type(self))
        state = self.__getstate__()
        new.__setstate__(state)
        new.quant_state = copy.deepcopy(state["quant_state"])
        new.data = copy.deepcopy(state["data"])
        return new

    def __copy__(self):
        new = type(self).__new__(type(self))
        new.__setstate__(self.__getstate__())
        return new

    # ---- construction from a pre-quantised checkpoint
    @classmethod
    def from_prequantized(cls, data: torch.Tensor, quantized_stats: dict[str, Any], requires_grad: bool = False,
                          device="cuda", module: Optional["Linear4bit"] = None, **kwargs) -> "Params4bit":
        self = torch.Tensor._make_subclass(cls, data.to(device))
        self.requires_grad = requires_grad
        self.quant_state = QuantState.from_dict(qs_dict=quantized_stats, device=device)
        self.blocksize = self.quant_state.blocksize
        self.compress_statistics = self.quant_state.nested
        self.quant_type = self.quant_state.quant_type
        self.bnb_quantized = True
        self.quant_storage = data.dtype
        self.module = module
        if module is not None:
            module.quant_state = self.quant_state
        return self

    # ---- lazy quantisation on device move
    def _quantize(self, device):
        w = self.data.contiguous().to(device)
        packed, state = F.quantize_4bit(
            w,
            blocksize=self.blocksize,
            compress_statistics=self.compress_statistics,
            quant_type=self.quant_type,
            quant_storage=self.quant_storage,
        )
        self.data = packed
        self.quant_state = state
        if self.module is not None:
            self.module.quant_state = state
        self.bnb_quantized = True
        return self

    def cpu(self):
        return self.to(device="cpu")

    def cuda(self, device=None, non_blocking: bool = False):
        return self.to(device="cuda" if device is None else device, non_blocking=non_blocking)

    def to(self, *args, **kwargs):
        device, dtype, non_blocking, _ = torch._C._nn._parse_to(*args, **kwargs)
        if device is not None and device.type != "meta" and not self.bnb_quantized:
            return self._quantize(device)
        if self.quant_state is not None:
            self.quant_state.to(device)
        return Params4bit(
            super().to(device=device, dtype=dtype, non_blocking=non_blocking),
            requires_grad=self.requires_grad,
            quant_state=self.quant_state,
            blocksize=self.blocksize,
            compress_statistics=self.compress_statistics,
            quant_type=self.quant_type,
            quant_storage=self.quant_storage,
            bnb_quantized=self.bnb_quantized,
        )

    # torch.chunk / torch.split (FSDP sharding, fused-QKV splitting) must keep the metadata
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        result = super().__torch_function__(func, types, args, kwargs)
        if func not in (torch.chunk, torch.split):
            return result
        src = args[0]

        def rewrap(t):
            return cls(
                data=t,
                requires_grad=src.requires_grad,
                quant_state=src.quant_state,
                blocksize=src.blocksize,
                compress_statistics=src.compress_statistics,
                quant_type=src.quant_type,
                quant_storage=src.quant_storage,
                module=src.module,
                bnb_quantized=src.bnb_quantized,
            )

        return tuple(rewrap(t) for t in result) if isinstance(result, tuple) else rewrap(result)


def fix_4bit_weight_quant_state_from_module(module: "Linear4bit") -> None:
    """FSDP and friends may replace the parameter by a plain tensor and lose ``quant_state``; the
    module keeps a copy so it can be put back (reference modules.py:487-501)."""
    if getattr(module.weight, "quant_state", None) is not None:
        return
    if getattr(module, "quant_state", None) is None:
        logger.warning(
            "FP4 quantization state not initialized. Please call .cuda() or .to(device) on the LinearFP4 layer first."
        )
    assert module.weight.shape[1] == 1
    if not isinstance(module.weight, Params4bit):
        module.weight = Params4bit(module.weight, quant_storage=module.quant_storage, bnb_quantized=True)
    module.weight.quant_state = module.quant_state


class _PreparedCall:
    """Owner of one handle of the C++ dispatcher's prepared-call table (csrc/torch_dispatch.cpp: linear4bit_prepare). The handle is
    released when THIS object dies, and the object never travels: ``copy.deepcopy`` / pickling of a module yield ``None`` in its
    place, so a copy (or a module loaded in another process) can never release - or call - the original's handle. The key is
    everything the prepared call captured: the quant state object, the STORAGE of weight and bias (``layer.bias.data = new`` keeps
    the Parameter object but changes ``data_ptr()``; in-place updates write the storage the prepared call aliases), dtypes."""

    __slots__ = ("handle", "quant_state", "weight_ptr", "bias_obj", "bias_ptr", "bias_dtype", "bias_grad", "compute_dtype")

    def __init__(self, handle, quant_state, weight, bias, compute_dtype):
        self.handle = handle
        self.quant_state = quant_state
        self.weight_ptr = weight.data_ptr()
        self.bias_obj = bias
        self.bias_ptr = None if bias is None else bias.data_ptr()
        self.bias_dtype = None if bias is None else bias.dtype
        self.bias_grad = bias is not None and bias.requires_grad
        self.compute_dtype = compute_dtype

    def matches(self, weight, bias, compute_dtype) -> bool:
        if self.handle is None or self.quant_state is not getattr(weight, "quant_state", None) or self.weight_ptr != weight.data_ptr():
            return False
        if bias is not self.bias_obj or self.compute_dtype is not compute_dtype:
            return False
        return bias is None or (bias.data_ptr() == self.bias_ptr and bias.dtype is self.bias_dtype
                                and bias.requires_grad == self.bias_grad)

    def release(self):
        handle, self.handle = self.handle, None
        if handle is not None:
            try:
                torch.ops.bitsandbytes_amd.linear4bit_release(handle)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass

    def __del__(self):
        self.release()

    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_no_prepared_call, ())


def _no_prepared_call():
    return None


class Linear4bit(nn.Linear):
    """QLoRA-style 4-bit linear layer (reference modules.py:504-637). Load fp weights into it, then
    ``.to("cuda")`` quantises them on the MI355X."""

    def __init__(self, input_features, output_features, bias=True, compute_dtype=None, compress_statistics=True,
                 quant_type="fp4", quant_storage=torch.uint8, device=None):
        super().__init__(input_features, output_features, bias, device)
        self.weight = Params4bit(
            self.weight.data,
            requires_grad=False,
            compress_statistics=compress_statistics,
            quant_type=quant_type,
            quant_storage=quant_storage,
            module=self,
        )
        self.compute_dtype = compute_dtype
        self.compute_type_is_set = compute_dtype is not None
        self.quant_state = None
        self.quant_storage = quant_storage

    def set_compute_type(self, x: torch.Tensor) -> None:
        if x.dtype in (torch.float32, torch.bfloat16):
            # safe to compute in the input's dtype
            self.compute_dtype = x.dtype
        elif x.dtype == torch.float16 and self.compute_dtype in (None, torch.float32):
            single = x.numel() == x.shape[-1]
            logger.warning(
                "Input type into Linear4bit is torch.float16, but bnb_4bit_compute_dtype=torch.float32 (default). "
                + ("This will lead to slow inference." if single else "This will lead to slow inference or training speed.")
            )

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)  # weight (packed) and bias
        state = getattr(self.weight, "quant_state", None)
        if state is not None:
            for k, v in state.as_dict(packed=True).items():
                destination[prefix + "weight." + k] = v if keep_vars else v.detach()

    # ---- prepared call (MI355X): eager decode is host-bound - the Python below (quant-state repair, dtype policy, matmul_4bit,
    # a ten-argument op call) costs more than the 4 us kernel it launches. Once the layer is quantised on the device, its
    # call is prepared ONCE in the C++ dispatcher library (csrc/torch_dispatch.cpp: linear4bit_prepare) and every later
    # forward that needs no autograd is a two-argument op call; anything that changes the layer (a new weight / quant state /
    # bias object, a moved weight) drops the handle. Results are those of the code below, by construction: the prepared
    # call runs the same gemm_4bit kernel glue with the same dtype policy.
    _prepared = None  # a _PreparedCall (below) or None

    def _prepared_drop(self):
        prep = self.__dict__.pop("_prepared", None)
        if prep is not None:
            prep.release()

    def _apply(self, fn, recurse=True):
        # .to() / .cuda() / .cpu() / .half(): the prepared call holds the packed weight and its statistics alive on the device
        # they were on - drop it before anything moves, the next eager forward prepares again
        self._prepared_drop()
        return super()._apply(fn, recurse)

    def _prepared_make(self, weight, quant_state, bias):
        from ..backends import hip

        self._prepared_drop()
        if not hip.NATIVE_DISPATCH or not weight.is_cuda or len(quant_state.shape) != 2:
            return None
        if quant_state.nested and quant_state.state2.blocksize != 256:
            return None
        if quant_state.nested:
            absmax, a8, code, offset = quant_state.state2.absmax, quant_state.absmax, quant_state.state2.code, quant_state.offset
        else:
            absmax, a8, code, offset = quant_state.absmax, None, None, None
        handle = torch.ops.bitsandbytes_amd.linear4bit_prepare(
            weight.data.view(-1, 1) if weight.dtype == torch.uint8 else weight.data, list(quant_state.shape), absmax, quant_state.blocksize,
            quant_state.quant_type, None if bias is None else bias.data, a8, code, offset, self.compute_dtype)
        prep = _PreparedCall(handle, quant_state, weight, bias, self.compute_dtype)
        self.__dict__["_prepared"] = prep
        return prep

    def forward(self, x: torch.Tensor):
        prep = self._prepared
        if prep is not None and x.is_cuda:
            if (prep.matches(self._parameters["weight"], self._parameters["bias"], self.compute_dtype)
                    and not (torch.is_grad_enabled() and (x.requires_grad or prep.bias_grad))
                    and not torch.compiler.is_compiling()):
                return torch.ops.bitsandbytes_amd.linear4bit_prepared(x, prep.handle)
        fix_4bit_weight_quant_state_from_module(self)
        quant_state = self.weight.quant_state

        if not self.compute_type_is_set:
            self.set_compute_type(x)
            self.compute_type_is_set = True

        inp_dtype = x.dtype
        if self.compute_dtype is not None:
            x = x.to(self.compute_dtype)

        bias = self.bias
        if bias is not None:
            if bias.dtype != x.dtype:
                bias.data = bias.data.to(x.dtype)
            bias = bias.to(self.compute_dtype)

        if (x.is_cuda and quant_state is not None and self.compute_type_is_set and not torch.compiler.is_compiling()
                and K_matches(x, quant_state)):
            # (prepared for the NEXT call; this one takes the ordinary path)
            self._prepared_make(self.weight, quant_state, self.bias)
        return matmul_4bit(x, self.weight, bias=bias, quant_state=quant_state).to(inp_dtype)


def K_matches(x: torch.Tensor, quant_state) -> bool:
    """The [N, K] orientation matmul_4bit's fused path expects (the legacy [K, N] layout keeps the ordinary path)."""
    return len(quant_state.shape) == 2 and x.shape[-1] == quant_state.shape[1]


def linear4bit_group_forward(layers, x: torch.Tensor):
    """``[layer(x) for layer in layers]`` for :class:`Linear4bit` layers that consume the same input (Q/K/V, gate/up).
    Same dtype policy as :meth:`Linear4bit.forward`; when every layer computes in the same dtype the matmuls go through
    :func:`bitsandbytes_amd.matmul_4bit_grouped` - one launch for a decode-sized batch on MI355X - and the outputs are
    bit-identical to calling the layers one by one."""
    from ..autograd import matmul_4bit_grouped

    layers = list(layers)
    # eager decode: every layer already holds a prepared call (csrc/torch_dispatch.cpp) -> ONE native call for the group
    if x.is_cuda and not torch.compiler.is_compiling():
        preps = [layer._prepared for layer in layers]
        if all(prep is not None and prep.matches(layer._parameters["weight"], layer._parameters["bias"], layer.compute_dtype)
               for prep, layer in zip(preps, layers)) and not (torch.is_grad_enabled() and (x.requires_grad or any(p.bias_grad for p in preps))):
            return list(torch.ops.bitsandbytes_amd.linear4bit_group_prepared(x, [p.handle for p in preps]))
    for layer in layers:
        fix_4bit_weight_quant_state_from_module(layer)
        if not layer.compute_type_is_set:
            layer.set_compute_type(x)
            layer.compute_type_is_set = True
    dtypes = {layer.compute_dtype for layer in layers}
    if len(dtypes) != 1:
        return [layer(x) for layer in layers]
    compute_dtype = dtypes.pop()
    inp_dtype = x.dtype
    xc = x if compute_dtype is None else x.to(compute_dtype)
    biases = []
    for layer in layers:
        bias = layer.bias
        if bias is not None:
            if bias.dtype != xc.dtype:
                bias.data = bias.data.to(xc.dtype)
            bias = bias.to(compute_dtype)
        biases.append(bias)
    if x.is_cuda and not torch.compiler.is_compiling():
        for layer in layers:  # (prepared for the NEXT call, as Linear4bit.forward does)
            qs = layer.weight.quant_state
            if qs is not None and layer._prepared is None and K_matches(xc, qs):
                layer._prepared_make(layer.weight, qs, layer.bias)
    ys = matmul_4bit_grouped(xc, [layer.weight for layer in layers], [layer.weight.quant_state for layer in layers], biases)
    return [y.to(inp_dtype) for y in ys]


class LinearFP4(Linear4bit):
    def __init__(self, input_features, output_features, bias=True, compute_dtype=None, compress_statistics=True,
                 quant_storage=torch.uint8, device=None):
        super().__init__(input_features, output_features, bias, compute_dtype, compress_statistics, "fp4",
                         quant_storage, device)


class LinearNF4(Linear4bit):
    def __init__(self, input_features, output_features, bias=True, compute_dtype=None, compress_statistics=True,
                 quant_storage=torch.uint8, device=None):
        super().__init__(input_features, output_features, bias, compute_dtype, compress_statistics, "nf4",
                         quant_storage, device)


class Embedding4bit(nn.Embedding):
    """4-bit embedding table (reference modules.py:880-978): load fp rows, ``.to("cuda")`` quantises them.

    Each row is a whole number of quantization blocks when ``embedding_dim % blocksize == 0``; the lookup is
    then ONE fused launch (gather the packed row + its scales, dequantize) instead of the reference's two
    ``F.embedding`` gathers plus ``dequantize_4bit`` - same values bit for bit. Otherwise the whole table is
    dequantized and indexed, as in the reference (slow path, warned about at construction)."""

    def __init__(self, num_embeddings, embedding_dim, dtype=None, quant_type="fp4", quant_storage=torch.uint8,
                 device=None):
        super().__init__(num_embeddings, embedding_dim, device=device, dtype=dtype)
        self.dtype = self.weight.data.dtype
        self.weight = Params4bit(
            self.weight.data,
            requires_grad=False,
            compress_statistics=None,
            quant_type=quant_type,
            quant_storage=quant_storage,
            module=self,
        )
        self.quant_state = None
        self.quant_storage = quant_storage
        if embedding_dim % self.weight.blocksize != 0:
            logger.warning(
                f"Embedding size {embedding_dim} is not divisible by block size {self.weight.blocksize}. "
                "This will lead to slow inference."
            )

    def _forward_with_partial_dequantize(self, input: torch.Tensor) -> torch.Tensor:
        state = self.weight.quant_state
        assert self.embedding_dim % state.blocksize == 0
        absmax = state.absmax
        if state.nested:  # un-nest the scales once per call; the row kernel takes plain fp32 absmax
            absmax = F.dequantize_blockwise(state.absmax, state.state2) + state.offset
            if absmax.dtype != torch.float32:
                absmax = absmax.float()
        if self.embedding_dim % 8 != 0 or input.dtype not in (torch.int32, torch.int64):
            # rows that are not whole packed dwords: compose the lookup from gathers like the reference
            packed = self.weight.data.view(torch.uint8).view(self.num_embeddings, self.embedding_dim // 2)
            rows = torch.nn.functional.embedding(input, packed).reshape(-1, 1)
            scales = torch.nn.functional.embedding(
                input, absmax.view(self.num_embeddings, self.embedding_dim // state.blocksize)
            ).reshape(-1)
            out = torch.ops.bitsandbytes.dequantize_4bit.default(
                rows, scales, state.blocksize, state.quant_type, (*input.shape, self.embedding_dim), state.dtype
            )
            return out.to(self.dtype)
        packed_rows = (self.weight.data.numel() * self.weight.data.element_size() * 2) // self.embedding_dim
        if packed_rows != self.num_embeddings or tuple(state.shape) != (self.num_embeddings, self.embedding_dim):
            raise RuntimeError(
                f"Embedding4bit: packed weight holds {packed_rows} rows of {self.embedding_dim}, quant_state.shape is "
                f"{tuple(state.shape)}, module expects ({self.num_embeddings}, {self.embedding_dim})"
            )
        out = torch.ops.bitsandbytes_amd.dequantize_4bit_rows.default(
            self.weight.data, absmax, input, self.embedding_dim, state.blocksize, state.quant_type, state.dtype
        )
        return out.to(self.dtype)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        raise NotImplementedError("Saving Embedding4bit module is not implemented")

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        fix_4bit_weight_quant_state_from_module(self)
        if self.embedding_dim % self.weight.quant_state.blocksize == 0:
            return self._forward_with_partial_dequantize(input)
        table = F.dequantize_4bit(self.weight.data, self.weight.quant_state)
        return torch.nn.functional.embedding(input, table).to(self.dtype)


class EmbeddingFP4(Embedding4bit):
    def __init__(self, num_embeddings, embedding_dim, dtype=None, quant_storage=torch.uint8, device=None):
        super().__init__(num_embeddings, embedding_dim, dtype=dtype, quant_type="fp4", quant_storage=quant_storage,
                         device=device)


class EmbeddingNF4(Embedding4bit):
    def __init__(self, num_embeddings, embedding_dim, dtype=None, quant_storage=torch.uint8, device=None):
        super().__init__(num_embeddings, embedding_dim, dtype=dtype, quant_type="nf4", quant_storage=quant_storage,
                         device=device)
