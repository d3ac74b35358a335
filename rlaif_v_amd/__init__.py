"""Import shim: the product package lives in ``rlaif-v_amd/`` (a name Python cannot import), so this
package re-points its search path there.  ``import rlaif_v_amd.hip`` loads ``rlaif-v_amd/hip.py``."""
import os as _os

_real = _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "rlaif-v_amd"))
__path__.insert(0, _real)  # noqa: F821
__version__ = "0.1.0"
