"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference sampling path + deterministic synthetic inputs.
Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
the product package `medfusion_amd` never imports it (tests/test_boundary_cpu.py checks).
"""
