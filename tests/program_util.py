"""Random step programs for the rk_program tests (PolyExtStepDef lists: see raiko_amd/circuit_program.py)."""
from raiko_amd.circuit_program import synthetic_program as random_program  # noqa: F401
