"""CPU oracle for the tagger hot path -- TEST INFRASTRUCTURE, never imported by the product package.

See ``oracle/restatement.py``.  Allowed importers: ``tests/``, ``__graft_entry__.smoke()``,
``bench.py``'s ``cpu_baseline`` leg.
"""
