"""Contrastive pre-training epochs on unlabeled slices only (config C5 of SURVEY.md; what
semi_seg/epochers/pretrain.py:23-182 does in the reference).

One class carries the behaviour, three thin public classes select it:

    PretrainEncoderEpocher            loaders must have full transform freedom, trains
    PretrainDecoderEpocher            loaders must not,                           trains
    PretrainDecoderEpocherInference   as the decoder epocher, but only evaluates the hooks' loss:
                                      augmentation off, no gradients, no optimizer step

A step: both views of the batch go through `model(cat(views), until=inference_until)` as ONE pass,
the first view's output is warped with the step's geometry (`mode="feature"`), and the hooks -- there
is no supervised term -- return the whole loss from `regularization(**kwargs)`; the kwargs are those
of the semi-supervised epocher plus `batch_data` (the raw loader batch, for hooks that need side
inputs such as superpixel maps).  The only loss meter is `reg_loss`, kept on the device.
"""
from __future__ import annotations

import random
from contextlib import contextmanager, nullcontext
from functools import partial

import torch
from torch import Tensor

from contrastyou.meters import MeterInterface
from contrastyou.utils.utils import get_lrs_from_optimizer
from semi_seg.epochers.epocher import SemiSupervisedEpocher, assert_transform_freedom
from semi_seg.epochers.helper import preprocess_input_with_twice_transformation

__all__ = ["PretrainEncoderEpocher", "PretrainDecoderEpocher", "PretrainDecoderEpocherInference"]


def _detached(loss):
    return loss.detach() if isinstance(loss, Tensor) else loss


class _PretrainEpocher(SemiSupervisedEpocher):
    loaders_have_total_freedom = False  # what `_assertion` demands of the loaders' transforms
    updates_weights = True              # False: evaluate the loss only (monitoring pass)

    def __init__(self, *, chain_dataloader, inference_until: str, **kwargs) -> None:
        super().__init__(**kwargs)
        self._chain_dataloader = chain_dataloader  # endless stream of two-view unlabeled batches
        self._inference_until = inference_until    # block name the forward stops at (None: full net)

    # ---- set-up -----------------------------------------------------------------------------------
    def _assertion(self):
        for loader in (self._labeled_loader, self._unlabeled_loader):
            if loader is not None:
                assert_transform_freedom(loader, self.loaders_have_total_freedom)

    def configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters = super().configure_meters(meters)
        meters.delete_meters(["sup_loss", "sup_dice"])  # nothing is supervised here
        return meters

    @staticmethod
    def _unzip_data(data, device):
        (view1, _), (view2, _), filename, partition, group = preprocess_input_with_twice_transformation(data, device)
        return (view1, view2), None, filename, partition, group

    # ---- the epoch --------------------------------------------------------------------------------
    def _run(self, **kwargs):
        self.meters["lr"].add(get_lrs_from_optimizer(self._optimizer))
        self._model.train()
        return self._run_implement(**kwargs)

    @contextmanager
    def disable_rising_augmentation(self):
        """identity augmentation inside the block (name kept from the reference's rising-based code)"""
        with self._affine_transformer.disabled():
            yield

    def _run_implement(self, **kwargs):
        quiet = nullcontext() if self.updates_weights else self.disable_rising_augmentation()
        grad = nullcontext() if self.updates_weights else torch.no_grad()
        with quiet, grad:
            for self.cur_batch_num, data in zip(self.indicator, self._chain_dataloader):
                seed = random.randint(0, int(1e7))
                (image, image_cf), _, filename, partition, group = self._unzip_data(data, self._device)
                image_tf = self.transform_with_seed(image_cf, mode="image", seed=seed)
                self.batch_update(cur_batch_num=self.cur_batch_num, unlabeled_image=image,
                                  unlabeled_image_tf=image_tf, seed=seed, unl_group=group, unl_partition=partition,
                                  unlabeled_filename=filename, batch_data=data)
                self._report(self.cur_batch_num, self.cur_batch_num == self.num_batches - 1)

    # ---- the step ---------------------------------------------------------------------------------
    def _forward_pass(self, unlabeled_image, unlabeled_image_tf):  # noqa: signature of this epocher family
        n = len(unlabeled_image)
        out = self._model(torch.cat([unlabeled_image, unlabeled_image_tf], dim=0), until=self._inference_until)
        return torch.split(out, (n, n), dim=0)

    def _hook_loss(self, *, unlabeled_image, unlabeled_image_tf, seed, unl_group, unl_partition,
                   unlabeled_filename, **extra):
        warp = partial(self.transform_with_seed, seed=seed, mode="feature")
        with self.autocast:
            out, out_of_tf = self.forward_pass(unlabeled_image=unlabeled_image, unlabeled_image_tf=unlabeled_image_tf)
            return self.regularization(
                seed=seed, unlabeled_image=unlabeled_image, unlabeled_image_tf=unlabeled_image_tf,
                unlabeled_tf_logits=out_of_tf, unlabeled_logits_tf=warp(out), label_group=unl_group,
                partition_group=unl_partition, unlabeled_filename=unlabeled_filename, affine_transformer=warp,
                **extra)

    def _batch_update(self, *, cur_batch_num: int, batch_data=None, **step):
        if not self.updates_weights:
            self.meters["reg_loss"].add(_detached(self._hook_loss(**step)))
            return
        self.optimizer_zero(self._optimizer, cur_iter=cur_batch_num)
        loss = self._hook_loss(batch_data=batch_data, **step)
        self.scale_loss(loss).backward()
        self.optimizer_step(self._optimizer, cur_iter=cur_batch_num)
        if self.meters:
            self.meters["reg_loss"].add(_detached(loss))


class PretrainEncoderEpocher(_PretrainEpocher):
    loaders_have_total_freedom = True


class PretrainDecoderEpocher(_PretrainEpocher):
    loaders_have_total_freedom = False


class PretrainDecoderEpocherInference(PretrainDecoderEpocher):
    updates_weights = False
