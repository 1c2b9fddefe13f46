"""Host shims of the reference's model call surface around the coder."""
from .resnet_vae import BidirectionalResidualBlock, BidirectionalResNetVAE, GraphedCompress, ModelError  # noqa: F401
from .lossy import Large2LevelVAE  # noqa: F401
