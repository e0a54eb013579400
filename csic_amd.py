"""Import alias: `import csic_amd` yields the package in ./chroma-subsampling-image-compressor_amd/
(whose directory name, fixed by the project layout, is not a valid Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("chroma-subsampling-image-compressor_amd")
sys.modules[__name__] = _pkg
