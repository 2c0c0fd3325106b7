"""Pretrain epochers: contrastive pre-training of the encoder (or encoder+decoder up to a named
block) on unlabeled slices only -- config C5 of SURVEY.md.

Mirrors semi_seg/epochers/pretrain.py:23-182 of the reference: same mixin layering, constructor
kwargs (`chain_dataloader`, `inference_until`), step skeleton (two views through
`model(cat(views), until=inference_until)`, feature-mode affine of the first view's output, hooks
provide the whole loss) and `regularization(**kwargs)` contract (+ `batch_data`).  The only
loss meter is `reg_loss`, kept on the device.
"""
from __future__ import annotations

import random
from abc import ABC, ABCMeta
from contextlib import contextmanager
from functools import partial

import torch
from torch import Tensor

from contrastyou.meters import MeterInterface
from contrastyou.utils.utils import get_lrs_from_optimizer
from semi_seg.epochers.epocher import SemiSupervisedEpocher, assert_transform_freedom
from semi_seg.epochers.helper import preprocess_input_with_twice_transformation


class _PretrainEpocherMixin(metaclass=ABCMeta):

    def __init__(self, *, chain_dataloader, inference_until: str, **kwargs) -> None:
        super().__init__(**kwargs)
        self._chain_dataloader = chain_dataloader
        self._inference_until = inference_until

    def configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters = super().configure_meters(meters)
        meters.delete_meters(["sup_loss", "sup_dice"])
        return meters

    def _run(self, **kwargs):
        self.meters["lr"].add(get_lrs_from_optimizer(self._optimizer))
        self._model.train()
        return self._run_implement(**kwargs)

    def _run_implement(self, **kwargs):
        for self.cur_batch_num, data in zip(self.indicator, self._chain_dataloader):
            seed = random.randint(0, int(1e7))
            (unlabeled_image, unlabeled_image_tf), _, unlabeled_filename, unl_partition, unl_group = \
                self._unzip_data(data, self._device)
            unlabeled_image_tf = self.transform_with_seed(unlabeled_image_tf, mode="image", seed=seed)
            self.batch_update(cur_batch_num=self.cur_batch_num, unlabeled_image=unlabeled_image,
                              unlabeled_image_tf=unlabeled_image_tf, seed=seed, unl_group=unl_group,
                              unl_partition=unl_partition, unlabeled_filename=unlabeled_filename,
                              batch_data=data)
            self._report(self.cur_batch_num, self.cur_batch_num == self.num_batches - 1)

    def _reg_loss(self, *, unlabeled_image, unlabeled_image_tf, seed, unl_group, unl_partition,
                  unlabeled_filename, **kwargs):
        with self.autocast:
            unlabeled_logits, unlabeled_tf_logits = self.forward_pass(
                unlabeled_image=unlabeled_image, unlabeled_image_tf=unlabeled_image_tf)
            unlabeled_logits_tf = self.transform_with_seed(unlabeled_logits, seed=seed, mode="feature")
            return self.regularization(
                seed=seed, unlabeled_image=unlabeled_image, unlabeled_image_tf=unlabeled_image_tf,
                unlabeled_tf_logits=unlabeled_tf_logits, unlabeled_logits_tf=unlabeled_logits_tf,
                label_group=unl_group, partition_group=unl_partition, unlabeled_filename=unlabeled_filename,
                affine_transformer=partial(self.transform_with_seed, seed=seed, mode="feature"), **kwargs)

    def _batch_update(self, *, cur_batch_num: int, unlabeled_image, unlabeled_image_tf, seed, unl_group,
                      unl_partition, unlabeled_filename, **kwargs):
        self.optimizer_zero(self._optimizer, cur_iter=cur_batch_num)
        reg_loss = self._reg_loss(unlabeled_image=unlabeled_image, unlabeled_image_tf=unlabeled_image_tf,
                                  seed=seed, unl_group=unl_group, unl_partition=unl_partition,
                                  unlabeled_filename=unlabeled_filename, **kwargs)
        self.scale_loss(reg_loss).backward()
        self.optimizer_step(self._optimizer, cur_iter=cur_batch_num)
        if self.meters:
            self.meters["reg_loss"].add(reg_loss.detach() if isinstance(reg_loss, Tensor) else reg_loss)

    def _forward_pass(self, unlabeled_image, unlabeled_image_tf):  # noqa
        n_unl = len(unlabeled_image)
        out = self._model(torch.cat([unlabeled_image, unlabeled_image_tf], dim=0), until=self._inference_until)
        return torch.split(out, (n_unl, n_unl), dim=0)

    @staticmethod
    def _unzip_data(data, device):
        (image, _), (image_ct, _), filename, partition, group = \
            preprocess_input_with_twice_transformation(data, device)
        return (image, image_ct), None, filename, partition, group


class _PretrainInferenceEpocherMixin(metaclass=ABCMeta):
    """loss evaluation without augmentation, gradients or optimizer steps (pretrain.py:104-158)"""

    def _batch_update(self, *, cur_batch_num: int, unlabeled_image, unlabeled_image_tf, seed, unl_group,
                      unl_partition, unlabeled_filename, **kwargs):
        kwargs.pop("batch_data", None)
        reg_loss = self._reg_loss(unlabeled_image=unlabeled_image, unlabeled_image_tf=unlabeled_image_tf,
                                  seed=seed, unl_group=unl_group, unl_partition=unl_partition,
                                  unlabeled_filename=unlabeled_filename)
        self.meters["reg_loss"].add(reg_loss.detach() if isinstance(reg_loss, Tensor) else reg_loss)

    @contextmanager
    def disable_rising_augmentation(self):
        with self._affine_transformer.disabled():
            yield

    def _run_implement(self, **kwargs):
        with self.disable_rising_augmentation(), torch.no_grad():
            return super()._run_implement(**kwargs)


class PretrainEncoderEpocher(_PretrainEpocherMixin, SemiSupervisedEpocher, ABC):
    def _assertion(self):
        assert_transform_freedom(self._labeled_loader, True)
        if self._unlabeled_loader is not None:
            assert_transform_freedom(self._unlabeled_loader, True)


class PretrainDecoderEpocher(_PretrainEpocherMixin, SemiSupervisedEpocher, ABC):
    def _assertion(self):
        assert_transform_freedom(self._labeled_loader, False)
        if self._unlabeled_loader is not None:
            assert_transform_freedom(self._unlabeled_loader, False)


class PretrainDecoderEpocherInference(_PretrainInferenceEpocherMixin, _PretrainEpocherMixin,
                                      SemiSupervisedEpocher, ABC):
    def _assertion(self):
        assert_transform_freedom(self._labeled_loader, False)
        if self._unlabeled_loader is not None:
            assert_transform_freedom(self._unlabeled_loader, False)
