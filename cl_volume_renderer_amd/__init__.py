"""MI355X-native volumetric path-tracing hot path (HIP for gfx950) behind the reference's
clw_* / renderer.hpp API.  The product is the C-ABI library ``libclwhip.so`` built from
``csrc/``; this Python package only holds the loader, the build recipe and the synthetic
scene generators used by tests and ``bench.py``."""

from . import scene  # noqa: F401
