"""Mirror of the reference's ``src/models`` package (module names kept so that
``from models.vit import ViViT`` style imports resolve to the build)."""
from . import vit  # noqa: F401
from .vit import ViViT  # noqa: F401
