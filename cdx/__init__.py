"""Import shim: `import cdx` resolves to the product package directory
`conditional-diffusion-model-for-compression_amd/` (a name Python cannot import
directly because of the hyphens).  No code lives here."""
import os as _os

_pkg = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                     "conditional-diffusion-model-for-compression_amd")
__path__.append(_pkg)

from ._api import *  # noqa: F401,F403,E402
from ._api import __all__  # noqa: E402
