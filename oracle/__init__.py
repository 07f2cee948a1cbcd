"""CPU oracle for the decombine hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; decombinator_amd/ never does (tests/test_no_oracle_in_product.py
enforces it)."""
