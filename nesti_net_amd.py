"""Import shim: the package directory is ``nesti-net_amd/`` (hyphenated, as the
layout contract names it), which Python cannot import by name.  Importing
``nesti_net_amd`` executes this file, which loads that directory as the package
``nesti_net_amd`` and replaces itself in ``sys.modules``."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "nesti-net_amd")
_spec = importlib.util.spec_from_file_location(
    "nesti_net_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nesti_net_amd"] = _mod
_spec.loader.exec_module(_mod)
