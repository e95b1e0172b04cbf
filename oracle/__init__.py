"""CPU oracle for the ADMM QP hot path.  TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see qps_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
``oracle.c_oracle`` wraps the C restatement (libqps_oracle.so, built by oracle/Makefile);
``oracle.qps_oracle_np`` is the independently written numpy/LAPACK mirror.
"""
from . import c_oracle, qps_oracle_np  # noqa: F401
