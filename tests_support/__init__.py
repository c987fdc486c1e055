"""Checker-side helpers shared by __graft_entry__.smoke() and bench.py (may import oracle/)."""
