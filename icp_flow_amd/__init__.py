"""Importable alias of the `icp-flow_amd/` package directory.

The product lives in `icp-flow_amd/` (a name Python cannot import because of the
dash); this shim makes it importable as `icp_flow_amd` by pointing the package
search path at that directory and executing its __init__.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "icp-flow_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
