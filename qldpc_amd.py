"""Import alias: the package directory is named ``qldpc-branched-off_amd`` (not a valid identifier), so this
bootstrap loads it under the importable name ``qldpc_amd``.  ``import qldpc_amd.decoding.sparse`` etc. work."""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qldpc-branched-off_amd")
_spec = importlib.util.spec_from_file_location("qldpc_amd", os.path.join(_root, "__init__.py"),
                                               submodule_search_locations=[_root])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["qldpc_amd"] = _mod
_spec.loader.exec_module(_mod)
