from ._functions import MatMul4Bit, matmul_4bit

__all__ = ["MatMul4Bit", "matmul_4bit"]
