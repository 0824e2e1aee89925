"""raiko_amd: MI355X-native STARK proving backend for raiko's risc0 block-proof path.

Only the hot path lives here: csrc/ (HIP kernels + the C ABI of include/raiko_hip.h),
hal.py (risc0 `Hal` operator mirror), prover.py (raiko `Prover` trait mirror),
segment.py (segments / tap sets / synthetic workload), pipeline.py (several segments in
flight per GPU), dist.py (segment sharding across GPUs).
"""
__all__ = ["hal", "prover", "segment", "pipeline", "dist"]
