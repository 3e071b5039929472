"""CPU oracle (test infrastructure).  See oracle/aft_oracle.c for the contract:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
