"""Counterparts of the reference's problem scripts (tests/eigenmode, tests/explosive_source)."""
