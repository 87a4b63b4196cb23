"""Name parity with pretraining/multimae/zorro_utils_quadruplet.py: the 4-modality driver (pretrain_mmae_my.py) imports
`TokenTypes` with a DNW member from there.  The blocks are the same as in zorro_utils."""
from .zorro_utils import (Attention, Block, Block_Fusion, FeedForward, GEGLU, LayerNorm, Mlp, exists)  # noqa: F401
from .zorro_utils import TokenTypesQuad as TokenTypes  # noqa: F401
