"""CPU oracle for the spsparse multiply() path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this package; the product (spsparse_amd/, include/) never does.
"""
