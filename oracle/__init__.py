"""CPU oracle — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this package, and only as the checker / the timed CPU baseline.
nanort_amd/ never imports it.
"""
