from ._functions import MatMul4Bit, matmul_4bit, matmul_4bit_grouped

__all__ = ["MatMul4Bit", "matmul_4bit", "matmul_4bit_grouped"]
