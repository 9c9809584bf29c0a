"""CPU oracle for the RepCONC PQ hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The shipped path is ``repconc_amd`` (HIP
kernels behind ``include/repconc_hip.h``), which raises if its extension is
missing instead of falling back to anything in here.
"""
