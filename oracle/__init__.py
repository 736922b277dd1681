"""CPU oracle for the ME-TRPO inner loop -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and there only as the checker
(or as the timed CPU baseline), never as the thing shipped.  The product path
(``me-trpo_amd/``) never imports this package and fails loudly without its HIP library.
"""
