"""GAN losses with the reference's class names and call signatures (reference models/Losses.py).

The loss heads are [B,1] scalars; what matters here is how they drive the discriminator: LogisticGAN's R1 penalty
differentiates D twice (``create_graph=True``), which the kernels support because every D op's backward is itself
built from differentiable ops (functional.py).
"""
import torch
import torch.nn.functional as TF

from . import functional as F


class GANLoss:
    """Base class -- reference models/Losses.py:20-51.

    ``mean_scale`` / ``batch_mean`` exist for data parallelism (the reference is single-device).  Every head
    below is a MEAN over the batch, so under a gradient all-reduce(SUM) over N ranks holding B/N samples each, the local
    mean carries ``mean_scale`` = 1/N and the sum of the rank losses is the loss of the global batch.  ``batch_mean`` is
    the mean of a [B,1] prediction over the GLOBAL batch where a head uses one as a term (the relativistic loss):
    ``torch.mean`` on one device, ``DataParallelGroup.global_mean`` (a differentiable all-reduce) on several."""

    def __init__(self, dis, mean_scale=1.0, batch_mean=None):
        self.dis = dis
        self.mean_scale = float(mean_scale)
        self.batch_mean = batch_mean if batch_mean is not None else torch.mean

    def _scaled(self, loss):
        return loss if self.mean_scale == 1.0 else loss * self.mean_scale

    def dis_loss(self, real_samps, fake_samps, height, alpha):
        raise NotImplementedError("dis_loss method has not been implemented")

    def gen_loss(self, real_samps, fake_samps, height, alpha):
        raise NotImplementedError("gen_loss method has not been implemented")


class ConditionalGANLoss:
    """Binary cross entropy on the logits of the label-conditioned discriminator -- reference models/Losses.py:54-93
    (``dis(samples, height, alpha, labels_in=labels)``).  ``mean_scale``: see GANLoss (the labels are sharded with the
    images)."""

    def __init__(self, dis, mean_scale=1.0):
        self.dis = dis
        self.mean_scale = float(mean_scale)

    _scaled = GANLoss._scaled

    def dis_loss(self, real_samps, fake_samps, labels, height, alpha):
        r_preds = torch.squeeze(self.dis(real_samps, height, alpha, labels_in=labels))
        f_preds = torch.squeeze(self.dis(fake_samps, height, alpha, labels_in=labels))
        real_loss = TF.binary_cross_entropy_with_logits(r_preds, torch.ones_like(r_preds))
        fake_loss = TF.binary_cross_entropy_with_logits(f_preds, torch.zeros_like(f_preds))
        return self._scaled((real_loss + fake_loss) / 2)

    def gen_loss(self, _, fake_samps, labels, height, alpha):
        preds = torch.squeeze(self.dis(fake_samps, height, alpha, labels_in=labels))
        return self._scaled(TF.binary_cross_entropy_with_logits(preds, torch.ones_like(preds)))


class StandardGAN(GANLoss):
    """Binary cross entropy on the logits -- reference models/Losses.py:96-134: ``(BCE(r, 1) + BCE(f, 0)) / 2`` for the
    discriminator, ``BCE(f, 1)`` for the generator.  The reference's ``gen_loss`` unpacks the discriminator's [B,1] output
    into three values (``preds, _, _ = self.dis(...)``, :131), which only runs at batch 3; the evident intent -- all
    predictions -- is what is computed here."""

    def dis_loss(self, real_samps, fake_samps, height, alpha):
        r_preds = torch.squeeze(self.dis(real_samps, height, alpha))
        f_preds = torch.squeeze(self.dis(fake_samps, height, alpha))
        real_loss = TF.binary_cross_entropy_with_logits(r_preds, torch.ones_like(r_preds))
        fake_loss = TF.binary_cross_entropy_with_logits(f_preds, torch.zeros_like(f_preds))
        return self._scaled((real_loss + fake_loss) / 2)

    def gen_loss(self, _, fake_samps, height, alpha):
        preds = torch.squeeze(self.dis(fake_samps, height, alpha))
        return self._scaled(TF.binary_cross_entropy_with_logits(preds, torch.ones_like(preds)))


class HingeGAN(GANLoss):
    """reference models/Losses.py:136-151."""

    def dis_loss(self, real_samps, fake_samps, height, alpha):
        r_preds = self.dis(real_samps, height, alpha)
        f_preds = self.dis(fake_samps, height, alpha)
        return self._scaled(torch.mean(TF.relu(1 - r_preds)) + torch.mean(TF.relu(1 + f_preds)))

    def gen_loss(self, _, fake_samps, height, alpha):
        return self._scaled(-torch.mean(self.dis(fake_samps, height, alpha)))


class RelativisticAverageHingeGAN(GANLoss):
    """reference models/Losses.py:154-189.  The two "average" terms are means over the whole batch (:166-167,183-184):
    ``batch_mean`` (the global mean under data parallelism, see GANLoss)."""

    def dis_loss(self, real_samps, fake_samps, height, alpha):
        r_preds = self.dis(real_samps, height, alpha)
        f_preds = self.dis(fake_samps, height, alpha)
        r_f_diff = r_preds - self.batch_mean(f_preds)
        f_r_diff = f_preds - self.batch_mean(r_preds)
        return self._scaled(torch.mean(TF.relu(1 - r_f_diff)) + torch.mean(TF.relu(1 + f_r_diff)))

    def gen_loss(self, real_samps, fake_samps, height, alpha):
        r_preds = self.dis(real_samps, height, alpha)
        f_preds = self.dis(fake_samps, height, alpha)
        r_f_diff = r_preds - self.batch_mean(f_preds)
        f_r_diff = f_preds - self.batch_mean(r_preds)
        return self._scaled(torch.mean(TF.relu(1 + r_f_diff)) + torch.mean(TF.relu(1 - f_r_diff)))


def logistic_heads(f_preds, r_preds, mean_scale, gen):
    """The softplus terms of the logistic loss (reference models/Losses.py:216-218,226) and their derivatives in ONE launch
    (``sgx_logistic_loss``): mean softplus(f) + mean softplus(-r) for the discriminator, mean softplus(-f) for the generator, times
    ``mean_scale``.  Device tensors only -- there is no host path in the product (the CPU-side data-parallel logic tests install their
    own stand-in for this function: tests/test_dist_gloo.py)."""
    return F.call(F.LogisticLossFn, f_preds, r_preds, mean_scale, bool(gen))


class LogisticGAN(GANLoss):
    """Non-saturating logistic loss with the R1 gradient penalty -- reference models/Losses.py:192-229.

    ``mean_scale`` (data parallelism, see GANLoss): the softplus terms are batch MEANS and carry it; the R1 term is a
    batch SUM (:210) and does not.
    """

    def __init__(self, dis, mean_scale=1.0):
        super().__init__(dis, mean_scale=mean_scale)

    def _r1_from_logit(self, real_logit, real_img):
        with F.data_grad_only():                      # only d(logit)/d(image) is needed here, not the parameter grads
            real_grads = torch.autograd.grad(outputs=real_logit, inputs=real_img,
                                             grad_outputs=torch.ones_like(real_logit),
                                             create_graph=True, retain_graph=True)[0]
        return F.call(F.SumSqFn, real_grads)            # SUM over batch and pixels (:210)

    def R1Penalty(self, real_img, height, alpha):
        real_img = real_img.detach().requires_grad_(True)
        return self._r1_from_logit(self.dis(real_img, height, alpha), real_img)

    def dis_loss(self, real_samps, fake_samps, height, alpha, r1_gamma=10.0, fake_logits=None):
        # The reference evaluates D(real) twice -- once for the logistic term (:216) and once more inside R1Penalty
        # (:201) -- with identical results (D is deterministic).  Here ONE forward of D(real) feeds both terms, and
        # the final backward walks that graph once with the two upstream gradients summed: same loss, same
        # gradients, one D forward and one D backward less per iteration.
        # ``fake_logits``: a callable returning D(fake) that the caller evaluates on another stream (StyleGAN._d_grads);
        # it is asked for as late as possible so that the D(real) forward and the R1 gradient pass overlap it.
        if r1_gamma != 0.0:
            real = real_samps.detach().requires_grad_(True)
            r_preds = self.dis(real, height, alpha)
            r1 = self._r1_from_logit(r_preds, real) * (r1_gamma * 0.5)
        else:
            r_preds = self.dis(real_samps, height, alpha)
            r1 = None
        if fake_logits is not None:
            f_preds = fake_logits()
        else:
            if callable(fake_samps):                  # produced lazily, after D(real)
                fake_samps = fake_samps()
            f_preds = self.dis(fake_samps, height, alpha)
        loss = logistic_heads(f_preds, r_preds, self.mean_scale, False)
        return loss if r1 is None else loss + r1

    def gen_loss(self, _, fake_samps, height, alpha):
        f_preds = self.dis(fake_samps, height, alpha)
        return logistic_heads(f_preds, None, self.mean_scale, True)
