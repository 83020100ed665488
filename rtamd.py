"""Import shim: the product package lives in `radiativetransfer.jl_amd/` (a directory name
Python cannot import directly because of the dot).  `import rtamd` loads it as a regular
package named `rtamd` (relative imports inside it keep working)."""
import importlib.util
import sys
from pathlib import Path

_pkg = Path(__file__).resolve().parent / "radiativetransfer.jl_amd"
_spec = importlib.util.spec_from_file_location("rtamd", _pkg / "__init__.py", submodule_search_locations=[str(_pkg)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rtamd"] = _mod
_spec.loader.exec_module(_mod)
