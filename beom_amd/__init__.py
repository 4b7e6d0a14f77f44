"""beom_amd — MI355X-native time-step engine for the BEOM shallow-water model.

Only the hot path (continuity + Montgomery/vorticity + Leith viscosity + u/v
momentum) lives on the GPU, behind the C-ABI of include/beom_hip.h.
This package is the Python host-side mirror of the reference's interface used by
tests and bench.py; the Fortran host (beom_amd/host/) is the drop-in under main.f95.
"""
__version__ = "0.1.0"
