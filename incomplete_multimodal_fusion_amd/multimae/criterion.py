"""Loss surface of the pretraining step with the reference's names and call signatures
(reference: pretraining/multimae/criterion.py), on the gfx950 kernels of ../csrc.

Differences that are deliberate and documented in DESIGN.md:
  * an all-zero mask returns a float32 zero (the reference returns an int64 `tensor(0)`, criterion.py:101-102);
  * a sample whose mask row is empty gets a zero gradient (the reference's 0/0 gives NaN gradients there).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class _MaskedPixelLoss(nn.Module):
    kind = 0

    def __init__(self, patch_size: int = 16, stride: int = 1, norm_pix=False):
        super().__init__()
        self.patch_size, self.stride, self.norm_pix = patch_size, stride, norm_pix
        self.scale_factor = patch_size // stride

    def _norm_target(self, target):
        # per-patch standardisation (criterion.py:90-96); order inside the patch vector is irrelevant for mean/var
        p = self.scale_factor
        B, C, H, W = target.shape
        t = target.reshape(B, C, H // p, p, W // p, p)
        mean = t.mean(dim=(1, 3, 5), keepdim=True)
        var = t.var(dim=(1, 3, 5), keepdim=True)
        return ((t - mean) / torch.sqrt(var + 1e-6)).reshape(B, C, H, W)

    def forward(self, input, target, mask=None):
        if self.norm_pix:
            target = self._norm_target(target)
        return ops.masked_loss_image(input, target, mask, self.kind, self.scale_factor)

    def forward_tokens(self, tokens, target, mask=None):
        """Fused unpatchify + loss on the decoder's (B*P, C*p*p) output."""
        if self.norm_pix:
            target = self._norm_target(target)
        return ops.masked_loss_tokens(tokens, target, mask, self.kind, self.scale_factor)


class MaskedMSELoss(_MaskedPixelLoss):     # criterion.py:61-115
    kind = 0


class MaskedL1Loss(_MaskedPixelLoss):      # criterion.py:118-172
    kind = 1


def dino_loss_func(student_output, teacher_output, teacher_temp=0.04, student_temp=0.1):   # criterion.py:328-335
    return ops.dino_loss(student_output, teacher_output, teacher_temp, student_temp)


class HardNegtive_loss(nn.Module):         # criterion.py:214-268 (sic)
    def __init__(self, tau_plus=0.1, beta=1.0, temperature=0.5, alpha=256, estimator='hard'):
        super().__init__()
        if estimator != 'hard':
            raise Exception('Invalid estimator selected. Only the reference default "hard" is built.')
        self.tau_plus, self.beta, self.temperature, self.estimator, self.alpha = tau_plus, beta, temperature, estimator, alpha

    def forward(self, out_1, out_2):
        return ops.hardneg_loss(out_1, out_2, self.tau_plus, self.beta, self.temperature)


# Names the reference driver imports but never calls on this path (pretrain_mmae.py:37, :492, :497): plain torch.
def byol_loss_func(p, z, simplified=True):
    return 2 - 2 * F.cosine_similarity(p, z.detach(), dim=-1).mean()


def vicreg(repr_a, repr_b, l=25, mu=25, nu=1):
    n, d = repr_a.shape
    inv = F.mse_loss(repr_a, repr_b)
    std = sum(F.relu(1 - torch.sqrt(z.var(dim=0) + 1e-4)).mean() for z in (repr_a, repr_b))
    cov = 0
    for z in (repr_a, repr_b):
        zc = z - z.mean(dim=0)
        c = zc.T @ zc / (n - 1)
        cov = cov + (c - torch.diag(torch.diag(c))).pow(2).sum() / d
    return l * inv + mu * std + nu * cov


class DINOLoss(nn.Module):
    def __init__(self, out_dim, teacher_temp=0.04, student_temp=0.1, center_momentum=0.9):
        super().__init__()
        raise NotImplementedError("DINOLoss (criterion.py:270-317) is commented out in the reference driver "
                                  "(pretrain_mmae.py:273) and is not part of the built path")
