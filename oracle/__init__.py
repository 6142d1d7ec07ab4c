"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference hot path (tc_gan/ssnode.py + ext/ssnode.c and
the BPTT generator/critic math of tc_gan/networks).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker -- never the product path
(``tc_gan_amd``), which must fail loudly when its HIP library is missing.
"""
