from . import parametrize
from .modules import (
    Embedding4bit,
    EmbeddingFP4,
    EmbeddingNF4,
    Linear4bit,
    LinearFP4,
    LinearNF4,
    Params4bit,
    linear4bit_group_forward,
)

__all__ = ["Linear4bit", "LinearFP4", "LinearNF4", "Params4bit", "Embedding4bit", "EmbeddingFP4", "EmbeddingNF4",
           "parametrize", "linear4bit_group_forward"]
