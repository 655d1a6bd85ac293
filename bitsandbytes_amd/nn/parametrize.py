"""4-bit parametrisation of arbitrary module parameters (reference ``bitsandbytes/nn/parametrize.py``).

For weights that do not live in an ``nn.Linear`` (fused MoE expert tensors, for instance): the parameter
is replaced by its packed 4-bit bytes and a ``torch.nn.utils.parametrize`` hook hands the module the
dequantized tensor whenever it reads the attribute. One ``dequantize_4bit`` launch per access; while a
forward of the owning module is running the result is cached so that repeated reads cost nothing.
The state dict keeps the clean key layout of ``Linear4bit`` (``<name>`` + ``<name>.absmax`` + ...).
"""
from __future__ import annotations

from typing import Any, Literal, Optional

import torch
import torch.nn as nn
import torch.nn.utils.parametrize as P

from .. import functional as F


class Bnb4bitParametrization(nn.Module):
    """``forward(packed) -> dequantized`` for one parameter; owns that parameter's :class:`QuantState`."""

    def __init__(self, quant_state: F.QuantState):
        super().__init__()
        self.quant_state = quant_state

    @torch.no_grad()
    def forward(self, quantized_param: torch.Tensor) -> torch.Tensor:
        return F.dequantize_4bit(quantized_param, self.quant_state)


def _require_parameter(module: nn.Module, param_name: str) -> nn.Parameter:
    if not hasattr(module, param_name):
        raise AttributeError(f"Module does not have parameter '{param_name}'")
    param = getattr(module, param_name)
    if not isinstance(param, nn.Parameter):
        raise TypeError(f"Parameter '{param_name}' is not an instance of nn.Parameter")
    return param


def replace_parameter_4bit_prequantized(module: nn.Module, param_name: str, qs_dict: dict[str, Any],
                                        device: torch.device) -> None:
    """The parameter already holds packed bytes (loaded from a checkpoint); ``qs_dict`` is its quant-state
    dict in the ``QuantState.as_dict`` layout."""
    _require_parameter(module, param_name)
    state = F.QuantState.from_dict(qs_dict, device=device)
    _attach(module, param_name, state)


def replace_parameter_4bit(module: nn.Module, param_name: str, compress_statistics: bool = False,
                           quant_type: Literal["nf4", "fp4"] = "nf4", blocksize: Optional[int] = None) -> None:
    """Quantize ``module.<param_name>`` in place (the tensor must already be on the HIP device) and make
    reads of the attribute return the dequantized value."""
    original = _require_parameter(module, param_name)
    packed, state = F.quantize_4bit(original.data, blocksize=blocksize, compress_statistics=compress_statistics,
                                    quant_type=quant_type)
    setattr(module, param_name, nn.Parameter(packed, requires_grad=False))
    del original
    _attach(module, param_name, state)


def _attach(module: nn.Module, param_name: str, state: F.QuantState) -> None:
    # unsafe=True: the parametrization changes shape and dtype (packed bytes -> fp tensor)
    P.register_parametrization(module, param_name, Bnb4bitParametrization(state), unsafe=True)
    _register_parametrization_hooks(module, param_name)


def _register_parametrization_hooks(module: nn.Module, param_name: str) -> None:
    """State-dict hook (clean key layout) + the forward hook pair that caches the dequantized tensor for the
    duration of one forward of the owning module (same private name as the reference's helper: its tests
    register the pair directly)."""
    if hasattr(module, "register_state_dict_post_hook"):
        module.register_state_dict_post_hook(_StateDictHook(param_name))
    module.register_forward_pre_hook(_enable_parametrization_cache)
    # always_call: also runs when forward raises or is aborted (non-reentrant activation checkpointing stops
    # its recompute mid-forward), otherwise the enable count leaks and the cache is never cleared again
    module.register_forward_hook(_disable_parametrization_cache, always_call=True)


def _enable_parametrization_cache(module: nn.Module, inputs: tuple[Any, ...]) -> None:
    P._cache_enabled += 1


def _disable_parametrization_cache(module: nn.Module, inputs: tuple[Any, ...], output: Any) -> None:
    # never below zero: with always_call the hook may fire without a matching pre-hook, and a negative
    # counter would read as "enabled" forever and pin every dequantized tensor in memory
    P._cache_enabled = max(0, P._cache_enabled - 1)
    if not P._cache_enabled:
        P._cache = {}


class _StateDictHook:
    """Rename ``parametrizations.<name>.original`` back to ``<name>`` and add the packed quant state."""

    def __init__(self, param_name: str):
        self.param_name = param_name

    def __call__(self, module: nn.Module, state_dict: dict[str, Any], prefix: str, local_metadata: Any) -> None:
        name = self.param_name
        raw_key = f"{prefix}parametrizations.{name}.original"
        if raw_key not in state_dict:
            return
        state_dict[f"{prefix}{name}"] = state_dict.pop(raw_key)
        assert P.is_parametrized(module, name)
        for hook in module.parametrizations[name]:
            if isinstance(hook, Bnb4bitParametrization):
                if hook.quant_state is not None:
                    for k, v in hook.quant_state.as_dict(packed=True).items():
                        state_dict[f"{prefix}{name}.{k}"] = v
                return
        raise AssertionError("Parametrization not found for the parameter.")
