"""tempo_amd: MI355X-native execution engine for tempo's TSDF hot path.

`from tempo_amd import TSDF` mirrors `from tempo import TSDF` (python/tempo/__init__.py:1).
"""
from .tsdf import TSDF, _ResampledTSDF  # noqa: F401

__all__ = ["TSDF"]
