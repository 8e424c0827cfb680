"""CPU oracle for the gnina CNN-scoring hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The shipped library (``gnina_amd``) never does; it fails loudly when its HIP
extension is missing rather than falling back to anything here.
"""
