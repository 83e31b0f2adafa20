"""jello_amd -- MI355X (gfx950) compute backend for the Jello/Vello 2D renderer hot path.

Python here is only glue for tests and benchmarks: the product is ``libjello_hip.so`` (HIP kernels
behind the C ABI of ``include/jello_hip.h``) plus ``libjello_host.so`` (the C++ mirror of Jello's
Scene / encoding / renderer recording surface).  See DESIGN.md and INTEGRATION.md.
"""
from ._lib import build, lib_paths, load_host, HostLibraryMissing  # noqa: F401
from .scene import (  # noqa: F401
    Scene, Path, Brush, Stroke, Color, ColorStop, RenderParams, BumpSizes, Fill, Join, Cap, Mix, Compose, Extend, Aa,
)
from .engine import Host, Recording, Engine, STAGE_NAMES, CMD  # noqa: F401
