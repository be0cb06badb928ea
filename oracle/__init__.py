"""CPU oracle for the NPVP Stage-2 predictor hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / the timed CPU
baseline.  The product package (``npvp_amd``) never imports it and raises if
its HIP library is missing.

Parity status: PINNED.  The restatement is checked (tests/test_oracle_golden.py)
against golden vectors that ``tests/golden/make_golden.py`` produced by
importing the reference ``models`` package from /root/reference on CPU
(torch 2.10; the reference pins torch 1.9 - see DESIGN.md "Oracle").
"""
from .ops import *          # noqa: F401,F403
from .model import *        # noqa: F401,F403
