"""Import shim: exposes the package directory `markov-process-analysis-on-point-cloud_amd/`
(whose name is not a valid Python identifier) as the importable package `mpa_amd`."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "markov-process-analysis-on-point-cloud_amd")
_spec = importlib.util.spec_from_file_location("mpa_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["mpa_amd"] = _pkg
_spec.loader.exec_module(_pkg)
