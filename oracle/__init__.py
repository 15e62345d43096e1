"""CPU oracle for the PlanGen layout->image hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker.  The product path
(``plangen_amd``) never imports this package and fails loudly when its HIP
library is missing.
"""
