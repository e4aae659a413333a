from .adapter3d_mixin import Adapter3DMixin  # noqa: F401
