"""Mirror of the reference's `pretraining/multimae` package surface used by pretrain_mmae.py:34-39."""
from .criterion import (DINOLoss, HardNegtive_loss, MaskedCrossEntropyLoss, MaskedL1Loss, MaskedMSELoss, byol_loss_func,
                        dino_loss_func, vicreg)
from .input_adapters import FusionInputAdapter, PatchedInputAdapter, SemSegInputAdapter
from .multimae_crossattn import (MultiMAE, pretrain_multimae_base, pretrain_multimae_large, pretrain_multimae_tiny)
from .output_adapters_simple import SpatialOutputAdapter
from .zorro_utils import TokenTypes
