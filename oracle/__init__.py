"""CPU oracle for the MeshGraphNets hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gnn_amd/`` may import this package;
it is the checker used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The product path is HIP-only and fails
loudly when its shared library is missing.

Parity status: pinned.  The restatement in ``mgn_oracle.py`` was checked
bit-for-bit against the reference Python (imported in the build container from
its read-only checkout with stand-ins for the absent torch_geometric /
torch_scatter wheels) by ``tests/golden/make_golden.py``, which also wrote the
fixtures in ``tests/golden/`` that ``tests/test_oracle_golden.py`` replays.
"""
from .mgn_oracle import *  # noqa: F401,F403
