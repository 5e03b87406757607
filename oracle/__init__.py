"""oracle/ -- TEST INFRASTRUCTURE ONLY (not shipped, not on the product path).

CPU restatement of the reference's algorithm for the OpenVLA-OFT action-chunk forward/backward path.  Only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it, and only as the checker.
"""
