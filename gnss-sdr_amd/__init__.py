"""gnss-sdr_amd: MI355X (gfx950) acquisition + tracking correlator engine for gnss-sdr.

Layout
  csrc/   hand-written HIP kernels + the C ABI (include/gnss_sdr_hip.h) -> libgnss_sdr_hip.so
  host/   C++ host side mirroring the reference's classes / adapters (Cpu_Multicorrelator_Real_Codes,
          pcps_acquisition core, AcquisitionInterface / TrackingInterface adapters)
  *.py    thin ctypes face of the C ABI used by tests/ and bench.py

The directory name carries a hyphen (it is named after the reference repo), so it is imported as
``gnss_sdr_amd`` through the loader shim ``gnss_sdr_amd.py`` at the repository root.
"""
from . import _lib  # noqa: F401
from ._lib import GSH_MAX_TAPS, GshError, load  # noqa: F401
from .build import build_library  # noqa: F401
