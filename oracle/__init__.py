"""CPU oracle for the zgml forward-inference path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
See oracle/zgml_oracle.h for what it restates and how it is pinned.
"""
