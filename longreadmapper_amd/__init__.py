"""longreadmapper_amd -- MI355X-native seed-and-extend hot path of lisanhu/LongReadMapper.

Only what the path needs: csrc/ (HIP kernels, C-ABI, CPU index builder), the ctypes binding
of the C-ABI (capi), and thin host objects mirroring the reference's interface
(index, mapper), plus workload tooling (synth) and multi-GPU plumbing (dist)."""
from . import _build  # noqa: F401

__all__ = ["capi", "index", "mapper", "synth", "dist"]
