"""athenapk_amd -- MI355X-native (gfx950) implementation of AthenaPK's per-meshblock
flux-divergence update: hand-written HIP kernels behind a C-ABI (include/apk_amd.h), a C++
host driver (include/apk_host.h) and thin ctypes/torch plumbing for tests and benchmarks.
"""
from . import lib  # noqa: F401

__all__ = ["lib", "hydro", "driver", "decks"]
