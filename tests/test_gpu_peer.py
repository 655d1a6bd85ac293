"""GPU: the one-shot peer all-gather (bitsandbytes_amd/peer.py, csrc/peer_gather.hip) and the pieces built on it
(ShardedLinear4bit(peer=), ShardedLinear4bitGroup, GraphedBlock). A 1-GPU box cannot run two devices, but it CAN run two
processes on one device with their gather buffers mapped into each other by hipIpc - the same kernel, the same flags, the same
ordering rules as across xGMI; the rendezvous goes over gloo. A real multi-GPU run is tests/test_gpu_parity.py's 2-rank RCCL
test (self-skipping)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, extra_env):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "checks", "peer_ranks.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=420)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"PEER_OK {rank}" in out, f"rank {rank} of {world}:\n{out[-3000:]}"


@pytest.mark.parametrize("world", [1, 2, 4])
def test_peer_allgather_processes_sharing_one_gpu(world):
    _run_ranks(world, {})


def test_peer_allgather_one_gpu_per_rank():
    """The same checks with one GPU per rank over real links (RCCL group for the rendezvous, hipIpc + peer-to-peer stores for the
    collective). Self-skips below two GPUs - the builder's boxes have one."""
    import torch

    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"needs 2 GPUs, this box has {n}")
    _run_ranks(4 if n >= 4 else 2, {"PEER_DEVICE_PER_RANK": "1"})
