from .modules import Linear4bit, LinearFP4, LinearNF4, Params4bit

__all__ = ["Linear4bit", "LinearFP4", "LinearNF4", "Params4bit"]
