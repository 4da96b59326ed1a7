"""Import alias.  The package directory is named `retrieval-augmented-mds_amd/` (the name the
project layout prescribes); a hyphen is not a valid Python identifier, so this one-file shim loads
that directory as the package `retrieval_augmented_mds_amd`:

    import retrieval_augmented_mds_amd as ram
    from retrieval_augmented_mds_amd.mips import Mips
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "retrieval-augmented-mds_amd")
_spec = _ilu.spec_from_file_location(
    __name__, _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
