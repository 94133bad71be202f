"""mslesions3d_amd — MI355X-native 3D-SSD lesion detection hot path.

Module names mirror the reference's flat ``lesions3d/`` files: ``ssd3d``, ``mobilenet``, ``base_network``,
``utils``.  Importing the package does not need a GPU; running anything does (there is no CPU fallback)."""
from . import _lib  # noqa: F401

__all__ = ["ssd3d", "mobilenet", "base_network", "utils", "engine", "optim"]
