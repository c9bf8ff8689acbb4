"""Mirror of the reference's ``src/models`` package (module names kept so that
``from models.vit import ViViT`` style imports resolve to the build)."""
from . import vit  # noqa: F401
from .vit import ViViT  # noqa: F401
from . import frame_transformer, transformer  # noqa: F401,E402
from .frame_transformer import FrameTransformer, TransformerBase, PositionalEncoding  # noqa: F401,E402
from .transformer import SimpleTransformer  # noqa: F401,E402
from . import custom_resnet, TPN  # noqa: F401,E402
