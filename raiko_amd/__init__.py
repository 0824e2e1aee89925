"""raiko_amd: MI355X-native STARK proving backend for raiko's risc0 block-proof path.

Only the hot path lives here: csrc/ (HIP kernels + the C ABI of include/raiko_hip.h),
hal.py (risc0 `Hal` operator mirror + the session entry point), prover.py (raiko `Prover` trait
mirror), segment.py (segments / tap sets / synthetic workload), toy_circuit.py (host side of the
example circuit behind rk_circuit_hooks), executor.py (RV32IM executor + segmenter in front of the
path), dist.py (segment sharding across ranks).
"""
__all__ = ["hal", "prover", "segment", "toy_circuit", "executor", "dist"]
