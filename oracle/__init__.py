"""Parity oracle: CPU restatements of the reference algorithms. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product (dcvc_amd/) never does.
"""
