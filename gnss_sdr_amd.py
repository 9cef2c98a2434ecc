"""Loader shim: makes the hyphenated package directory ``gnss-sdr_amd/`` importable as ``gnss_sdr_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gnss-sdr_amd")
_spec = importlib.util.spec_from_file_location("gnss_sdr_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gnss_sdr_amd"] = _mod
_spec.loader.exec_module(_mod)
