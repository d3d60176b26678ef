"""RCCL under HIP-graph capture, exercised on ONE GPU: a one-rank `nccl` process group, the training step captured
with its gradient all-reduce INSIDE the graph (GraphedTrainStep(split_after=la4, capture_reduce=True): the early
bucket's collective is issued in the middle of the captured backward, the rest at its end), replayed three times and
compared with the plain step.  Runs in a fresh subprocess (the group is created before any other GPU work); the
multi-rank result itself stays covered by the gloo tests (tests/test_distributed_cpu.py) -- no multi-GPU box is
available to this round."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_captured_all_reduce_on_one_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_capture_worker.py"), str(port)], capture_output=True,
                       text=True, timeout=600)
    sys.stdout.write(r.stdout[-2000:])
    assert r.returncode == 0, r.stderr[-4000:]
    assert "rccl-capture ok" in r.stdout
