"""Launched by tests/test_gpu_parity.py::test_sharded_linear4bit_over_rccl_two_ranks under torch.distributed.run (one process per
GPU): executes the test's per-rank script (passed through the environment so that the assertions live in the test file)."""
import os

exec(compile(os.environ["BNB_TWO_RANK_SCRIPT"], "two_rank_script", "exec"))
