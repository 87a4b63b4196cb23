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


class MaskedCrossEntropyLoss(nn.Module):   # criterion.py:24-58 (the `dnw` modality of pretrain_mmae_my.py:67-74)
    def __init__(self, patch_size: int = 16, stride: int = 1, label_smoothing: float = 0.0):
        super().__init__()
        self.patch_size, self.stride, self.label_smoothing = patch_size, stride, label_smoothing
        self.scale_factor = patch_size // stride

    def forward(self, input, target, mask=None):
        return ops.masked_ce_image(input, target, mask, self.scale_factor, self.label_smoothing)

    def forward_tokens(self, tokens, target, mask=None):
        C = tokens.shape[1] // (self.scale_factor * self.scale_factor)
        return ops.masked_ce_tokens(tokens, target, mask, C, self.scale_factor, self.label_smoothing)


def dino_loss_func(student_output, teacher_output, teacher_temp=0.04, student_temp=0.1):   # criterion.py:328-335
    return ops.dino_loss(student_output, teacher_output, teacher_temp, student_temp)


class HardNegtive_loss(nn.Module):         # criterion.py:214-268 (sic)
    def __init__(self, tau_plus=0.1, beta=1.0, temperature=0.5, alpha=256, estimator='hard'):
        super().__init__()
        if estimator not in ('hard', 'easy'):
            raise Exception('Invalid estimator selected. Please use any of [hard, easy]')          # criterion.py:259-260
        self.tau_plus, self.beta, self.temperature, self.estimator, self.alpha = tau_plus, beta, temperature, estimator, alpha

    def forward(self, out_1, out_2):
        if self.estimator == 'easy':
            # Ng = neg.sum(-1) (criterion.py:257-258) is the 'hard' expression at beta = 0 (unit importance weights) and
            # tau_plus = 0; its clamp is then never active: every negative term is >= e^(-1/T), so their sum is >= N e^(-1/T)
            return ops.hardneg_loss(out_1, out_2, 0.0, 0.0, self.temperature)
        return ops.hardneg_loss(out_1, out_2, self.tau_plus, self.beta, self.temperature)


# Names the reference driver imports but never calls on this path (pretrain_mmae.py:37, :492, :497): plain torch.
def byol_loss_func(p, z, simplified=True):                                                  # criterion.py:319-326
    if simplified:
        return 2 - 2 * F.cosine_similarity(p, z.detach(), dim=-1).mean()
    return 2 - 2 * (F.normalize(p, dim=1) * F.normalize(z, dim=1).detach()).sum(dim=1).mean()


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
    """criterion.py:270-317 -- constructed only in a commented-out line of the reference driver (pretrain_mmae.py:273);
    restated for API completeness in plain torch ((B, out_dim) tensors, not on the hot path).  The reference loops over the
    ROWS of the two (B, D) inputs as if they were views: loss = mean over ordered pairs t != s of
    sum_d -softmax((teacher_t - center) / Tt)_d * log_softmax(student_s / Ts)_d  (:281-299, vectorised here); the centre is
    an EMA of the mean over ALL elements of the normalised teacher batch (`torch.cat(rows).mean(dim=0)`, :309-314)."""

    def __init__(self, out_dim, teacher_temp=0.04, student_temp=0.1, center_momentum=0.9):
        super().__init__()
        self.student_temp, self.teacher_temp, self.center_momentum = student_temp, teacher_temp, center_momentum
        self.register_buffer("center", torch.zeros(1, out_dim))

    def forward(self, student_output, teacher_output):
        student_output = F.normalize(student_output.float(), dim=1)
        teacher_output = F.normalize(teacher_output.float(), dim=1)
        s = F.log_softmax(student_output / self.student_temp, dim=-1)                             # (B, D)
        t = F.softmax((teacher_output - self.center) / self.teacher_temp, dim=-1).detach()        # (B, D)
        pair = -(t @ s.t())                                                                       # [t_idx, s_idx]
        B = pair.shape[0]
        total = (pair.sum() - torch.diagonal(pair).sum()) / (B * (B - 1))
        self.update_center(teacher_output)
        return total

    @torch.no_grad()
    def update_center(self, teacher_output):
        batch_center = teacher_output.mean().reshape(1)
        self.center = self.center * self.center_momentum + (1 - self.center_momentum) * batch_center
