"""Small host-side helpers of the reference's ``bitsandbytes/utils.py`` that callers of the 4-bit path use:
the packed-dict codec of the QuantState blob and the module-tree walk that swaps ``nn.Linear`` for a quantized
layer. (Outlier tracing and the LLM.int8() helpers of that file are outside this path.)"""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import torch

from .functional import pack_dict_to_tensor, unpack_tensor_to_dict

__all__ = ["pack_dict_to_tensor", "unpack_tensor_to_dict", "replace_linear", "sync_gpu"]


def replace_linear(model: torch.nn.Module, linear_replacement: Callable[..., torch.nn.Module],
                   skip_modules: Iterable[str] = ("lm_head",), copy_weights: bool = False,
                   post_processing_function: Optional[str] = None) -> torch.nn.Module:
    """Swap every ``nn.Linear`` child (depth-first, children before their parent's own entries) whose attribute
    name is not in ``skip_modules`` for ``linear_replacement(in_features, out_features, has_bias)`` - e.g.
    ``lambda i, o, b: Linear4bit(i, o, b, quant_type="nf4")``. With ``copy_weights`` the new layer takes over the
    old layer's ``weight`` / ``bias`` Parameters as they are (the caller wraps them, e.g. into ``Params4bit``);
    ``post_processing_function`` names a method of the REPLACED module that is called with it afterwards.
    Same contract as reference bitsandbytes/utils.py:121-163; returns ``model``."""
    skip = set(skip_modules)
    for name, child in list(model.named_children()):
        if next(child.children(), None) is not None:
            replace_linear(child, linear_replacement, skip, copy_weights, post_processing_function)
        if not isinstance(child, torch.nn.Linear) or name in skip:
            continue
        new_layer = linear_replacement(child.in_features, child.out_features, child.bias is not None)
        if copy_weights:
            new_layer.weight = child.weight
            new_layer.bias = child.bias
        model._modules[name] = new_layer
        if post_processing_function is not None:
            hook = getattr(child, post_processing_function, None)
            if hook is not None:
                hook(child)
    return model


def sync_gpu(t: torch.Tensor) -> None:
    """Wait for the device that holds ``t`` (reference bitsandbytes/utils.py:204-208); a no-op for CPU tensors."""
    if t.device.type == "cuda":
        torch.cuda.synchronize(t.device)
