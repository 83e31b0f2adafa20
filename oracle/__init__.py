"""CPU oracle (TEST INFRASTRUCTURE).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product path (jello_amd/) never does."""
