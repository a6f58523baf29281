"""ORACLE — test infrastructure, NOT product code.

CPU restatements of the reference algorithms on the CenterPose inference hot path
(backbone forward, DCNv2, decode, PnP).  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import this package; the product
(``centerpose_amd``) never does and fails loudly when its HIP library is missing.
"""
