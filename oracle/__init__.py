"""ORACLE -- test infrastructure only.

CPU restatement of the reference's per-chunk hot path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import anything from this package; the
product package `infinisst_amd` never does.
"""
