"""The epochers of the hot path: `SemiSupervisedEpocher` (one training step = labeled +
unlabeled + transformed-unlabeled forward, supervised KL loss, hook regularisers, backward,
optimizer step), `FineTuneEpocher` (supervised only) and `EvalEpocher` (Dice).

Same classes, constructor kwargs, hook call order and `regularization(**kwargs)` contract as
semi_seg/epochers/epocher.py:39-418 of the reference; differences are confined to where the
reference forces host syncs inside the step:
  * losses / Dice are accumulated on the device (meters read them back at summary time);
  * the supervised loss goes through KL_div.from_logits (fused softmax+KL kernel) when the
    criterion supports it, instead of softmax -> one_hot -> `unique()` assert -> KL;
  * `statistics()` for the progress bar is sampled every `report_every` batches.
"""
from __future__ import annotations

import random
from contextlib import nullcontext
from functools import partial
from typing import Any, Dict, Optional

import torch
from torch import Tensor, nn

from contrastyou.epochers.base import EpocherBase as _EpocherBase
from contrastyou.meters import AverageValueMeter, MeterInterface, UniversalDice
from contrastyou.utils.general import class2one_hot
from cyhip import graphed, ops
from contrastyou.utils.utils import (class_name, disable_tracking_bn_stats, get_dataset, get_lrs_from_optimizer,
                                     get_model)
from semi_seg.augment import AffineAugment
from semi_seg.epochers.helper import (preprocess_input_with_single_transformation,
                                      preprocess_input_with_twice_transformation)


def assert_transform_freedom(dataloader, is_true):
    transform = get_dataset(dataloader).transforms
    assert transform._total_freedom is is_true  # noqa


def _sup_loss(criterion, logits: Tensor, target: Tensor, num_classes: int) -> Tensor:
    """criterion(softmax(logits), one_hot(target)) -- fused when the criterion offers it"""
    labels = target.squeeze(1)
    if hasattr(criterion, "from_logits"):
        return criterion.from_logits(logits, labels)
    return criterion(logits.softmax(1), class2one_hot(labels, num_classes))


class EpocherBase(_EpocherBase):
    """adds the batch_update / forward_pass / regularization hook protocol (epocher.py:39-116)"""

    @property
    def num_classes(self):
        return get_model(self._model).num_classes

    def __init__(self, *, model: nn.Module, num_batches: int, cur_epoch=0, device="cpu", scaler, **kwargs) -> None:
        super().__init__(model=model, num_batches=num_batches, cur_epoch=cur_epoch, device=device, scaler=scaler,
                         **kwargs)
        self._retain_graph = False
        self.report_every = 50

    def init(self, trainer=None) -> None:
        super().init(trainer=trainer)
        self._assertion()

    @property
    def retain_graph(self):
        return self._retain_graph

    @retain_graph.setter
    def retain_graph(self, enable):
        self._retain_graph = enable

    def run(self, **kwargs):
        if not self._initialized:
            raise RuntimeError(f"Call {class_name(self)}.init() before {class_name(self)}.run()")
        return super().run(**kwargs)

    def _assertion(self):
        pass

    def _batch_update(self, **kwargs) -> Optional[Dict[str, Any]]:
        ...

    def batch_update(self, **kwargs) -> Optional[Dict[str, Any]]:
        for h in self._hooks:
            h.call_before_batch_update(**kwargs)
        result = self._batch_update(**kwargs)
        for h in self._hooks:
            h.call_after_batch_update(**kwargs, result_dict=result)
        return result

    def forward_pass(self, **kwargs):
        for h in self._hooks:
            h.call_before_forward_pass(**kwargs)
        result = self._forward_pass(**kwargs)
        for h in self._hooks:
            h.call_after_forward_pass(**kwargs, result_dict=result)
        return result

    def _forward_pass(self, **kwargs) -> Any:
        ...

    def regularization(self, **kwargs):
        for h in self._hooks:
            h.call_before_regularization(**kwargs)
        result = self._regularization(**kwargs)
        for h in self._hooks:
            h.call_after_regularization(**kwargs, result_dict=result)
        return result

    def _regularization(self, **kwargs):
        return torch.tensor(0, dtype=torch.float, device=self._device)

    def _report(self, i: int, last: bool):
        if last or (self.report_every and i % self.report_every == 0 and self.verbose):
            self.indicator.set_postfix_statics2(dict(self.meters.statistics()), force_update=last)


class EvalEpocher(EpocherBase):
    meter_focus = "eval"

    def get_score(self) -> float:
        return self.meters["dice"].summary()["DSC_mean"]

    def __init__(self, *, model: nn.Module, loader, sup_criterion, cur_epoch=0, device="cpu", scaler,
                 accumulate_iter: int) -> None:
        super().__init__(model=model, num_batches=len(loader), cur_epoch=cur_epoch, device=device, scaler=scaler,
                         accumulate_iter=accumulate_iter)
        self._loader = loader
        self._sup_criterion = sup_criterion

    def configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters = super().configure_meters(meters)
        C = self.num_classes
        meters.register_meter("loss", AverageValueMeter())
        meters.register_meter("dice", UniversalDice(C, report_axis=list(range(1, C))))
        return meters

    def _run(self, **kwargs):
        self._model.eval()
        return self._run_implement()

    @torch.no_grad()
    def _run_implement(self):
        for i, eval_data in zip(self.indicator, self._loader):
            eval_img, eval_target, file_path, _, group = self._unzip_data(eval_data, self._device)
            self.batch_update(eval_img=eval_img, eval_target=eval_target, eval_group=group, file_names=file_path)
            self._report(i, i == self.num_batches - 1)

    def _batch_update(self, *, eval_img, eval_target, eval_group, file_names):
        with self.autocast:
            eval_logits = self._model(eval_img)
            eval_loss = _sup_loss(self._sup_criterion, eval_logits, eval_target, self.num_classes)
        self.meters["loss"].add(eval_loss.detach())
        self.meters["dice"].add_logits(eval_logits, eval_target, group_name=eval_group)

    @staticmethod
    def _unzip_data(data, device):
        return preprocess_input_with_single_transformation(data, device)


class SemiSupervisedEpocher(EpocherBase):
    meter_focus = "semi"

    def _assertion(self):
        assert_transform_freedom(self._labeled_loader, False)
        if self._unlabeled_loader is not None:
            assert_transform_freedom(self._unlabeled_loader, False)

    def __init__(self, *, model: nn.Module, optimizer, labeled_loader, unlabeled_loader, sup_criterion,
                 num_batches: int, cur_epoch=0, device="cpu", two_stage: bool = False, disable_bn: bool = False,
                 scaler, accumulate_iter: int = 1, **kwargs) -> None:
        super().__init__(model=model, num_batches=num_batches, cur_epoch=cur_epoch, device=device, scaler=scaler,
                         accumulate_iter=accumulate_iter)
        self._optimizer = optimizer
        self._labeled_loader = labeled_loader
        self._unlabeled_loader = unlabeled_loader
        self._sup_criterion = sup_criterion
        # geometry + intensity augmentation of the unlabeled view (epocher.py:226-238)
        self._affine_transformer = AffineAugment(scale=(0.8, 1.3), rotation=(-45, 45), translation=(-0.1, 0.1),
                                                 mirror_p=0.9, gamma=(0.5, 2))
        self._two_stage = two_stage
        self._disable_bn = disable_bn
        self.cur_batch_num = 0

    def transform_with_seed(self, features, *, mode: str, seed: int):
        assert mode in {"image", "feature"}, f"mode must be either `image` or `feature`, given {mode}"
        return self._affine_transformer(features, mode=mode, seed=seed)

    def configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters = super().configure_meters(meters)
        C = self.num_classes
        meters.register_meter("sup_loss", AverageValueMeter())
        meters.register_meter("sup_dice", UniversalDice(C, report_axis=list(range(1, C))))
        meters.register_meter("reg_loss", AverageValueMeter())
        return meters

    def _run(self, **kwargs):
        self.meters["lr"].add(get_lrs_from_optimizer(self._optimizer))
        self._model.train()
        return self._run_implement()

    def _run_implement(self):
        if len(self._unlabeled_loader) == 0:  # fully supervised setting
            self._unlabeled_loader = self._labeled_loader
        for self.cur_batch_num, labeled_data, unlabeled_data in zip(self.indicator, self._labeled_loader,
                                                                    self._unlabeled_loader):
            seed = random.randint(0, int(1e7))
            (labeled_image, _), labeled_target, labeled_filename, _, label_group = \
                self._unzip_data(labeled_data, self._device)
            (unlabeled_image, unlabeled_image_cf), _, unlabeled_filename, unl_partition, unl_group = \
                self._unzip_data(unlabeled_data, self._device)
            unlabeled_image_tf = self.transform_with_seed(unlabeled_image_cf, seed=seed, mode="image")
            self.batch_update(cur_batch_num=self.cur_batch_num, labeled_image=labeled_image,
                              labeled_target=labeled_target, labeled_filename=labeled_filename,
                              label_group=label_group, unlabeled_image=unlabeled_image,
                              unlabeled_image_tf=unlabeled_image_tf, seed=seed, unl_group=unl_group,
                              unl_partition=unl_partition, unlabeled_filename=unlabeled_filename,
                              retain_graph=self._retain_graph)
            self._report(self.cur_batch_num, self.cur_batch_num == self.num_batches - 1)

    def _batch_update(self, *, cur_batch_num: int, labeled_image, labeled_target, labeled_filename, label_group,
                      unlabeled_image, unlabeled_image_tf, seed, unl_group, unl_partition, unlabeled_filename,
                      retain_graph=False, **kwargs):
        self.optimizer_zero(self._optimizer, cur_iter=cur_batch_num)
        with self.autocast:
            label_logits, unlabeled_logits, unlabeled_tf_logits = self.forward_pass(
                labeled_image=labeled_image, unlabeled_image=unlabeled_image, unlabeled_image_tf=unlabeled_image_tf)
            unlabeled_logits_tf = self.transform_with_seed(unlabeled_logits, seed=seed, mode="feature")
            sup_loss = _sup_loss(self._sup_criterion, label_logits, labeled_target, self.num_classes)
            reg_loss = self.regularization(
                seed=seed, labeled_image=labeled_image, labeled_target=labeled_target,
                unlabeled_image=unlabeled_image, unlabeled_image_tf=unlabeled_image_tf,
                unlabeled_tf_logits=unlabeled_tf_logits, unlabeled_logits_tf=unlabeled_logits_tf,
                label_group=unl_group, partition_group=unl_partition, labeled_filename=labeled_filename,
                unlabeled_filename=unlabeled_filename,
                affine_transformer=partial(self.transform_with_seed, seed=seed, mode="feature"))
        total_loss = sup_loss + reg_loss
        self.scale_loss(total_loss).backward(retain_graph=retain_graph)
        self.optimizer_step(self._optimizer, cur_iter=cur_batch_num)
        if self.on_master:
            with torch.no_grad():
                self.meters["sup_loss"].add(sup_loss.detach())
                self.meters["sup_dice"].add_logits(label_logits, labeled_target, group_name=label_group)
                self.meters["reg_loss"].add(reg_loss.detach() if isinstance(reg_loss, Tensor) else reg_loss)

    def _forward_pass(self, labeled_image, unlabeled_image, unlabeled_image_tf):
        n_l, n_unl = len(labeled_image), len(unlabeled_image)
        if self._two_stage:
            if ops.TWO_STREAM and labeled_image.is_cuda:
                return self._forward_two_streams(labeled_image, unlabeled_image, unlabeled_image_tf)
            label_logits = self._model(labeled_image)
            with self._bn_context(self._model):
                unlabeled_logits, unlabeled_tf_logits = torch.split(
                    self._model(torch.cat([unlabeled_image, unlabeled_image_tf], dim=0)), [n_unl, n_unl], dim=0)
            return label_logits, unlabeled_logits, unlabeled_tf_logits
        logits = self._model(torch.cat([labeled_image, unlabeled_image, unlabeled_image_tf], dim=0))
        return torch.split(logits, [n_l, n_unl, n_unl], dim=0)

    def _forward_two_streams(self, labeled_image, unlabeled_image, unlabeled_image_tf):
        """The two passes of the two-stage forward are independent networks evaluations (epocher.py:
        351-357) and neither fills the GPU at these batch sizes, so the unlabeled pass is enqueued on
        a second stream and overlaps with the labeled pass (autograd replays each pass's backward on
        its own stream, so the backward passes overlap too).  Order-sensitive shared state -- BN
        running statistics and batch counters, accumulated BN gradients -- is serialised in the
        reference's order (labeled pass first) by cyhip.ops.ordered; `CY_TWO_STREAM=0` disables."""
        n_unl = len(unlabeled_image)
        dev = labeled_image.device
        xb = torch.cat([unlabeled_image, unlabeled_image_tf], dim=0)
        if hasattr(self._model, "arch_elements"):
            # both passes, forward and backward, as HIP graphs (the step is host-bound): cyhip/graphed.py
            res = graphed.two_pass(self._model, self._bn_context, labeled_image, xb, self._disable_bn,
                                   self._autocast_dtype if self.use_mixed_train else None)
            if res is not None:
                label_logits, both = res
                unlabeled_logits, unlabeled_tf_logits = torch.split(both, [n_unl, n_unl], dim=0)
                return label_logits, unlabeled_logits, unlabeled_tf_logits
        main = torch.cuda.current_stream(dev)
        side = ops.side_stream(dev, "pass2")
        side.wait_stream(main)  # inputs, zeroed gradients, the previous step's optimizer update
        ops.note_side_work(side)
        label_logits = self._model(labeled_image)
        with torch.cuda.stream(side), self._bn_context(self._model):
            both = self._model(xb)
        main.wait_stream(side)
        both.record_stream(main)
        unlabeled_logits, unlabeled_tf_logits = torch.split(both, [n_unl, n_unl], dim=0)
        return label_logits, unlabeled_logits, unlabeled_tf_logits

    @property
    def _bn_context(self):
        return disable_tracking_bn_stats if self._disable_bn else (lambda model: nullcontext())

    @staticmethod
    def _unzip_data(data, device):
        (image, target), (image_ct, target_ct), filename, partition, group = \
            preprocess_input_with_twice_transformation(data, device)
        return (image, image_ct), target, filename, partition, group

    def _regularization(self, **kwargs):
        if len(self._hooks) > 0:
            return sum(h(**kwargs) for h in self._hooks)
        return torch.tensor(0, device=self.device, dtype=torch.float)


class FineTuneEpocher(SemiSupervisedEpocher):
    meter_focus = "ft"

    def __init__(self, *, model: nn.Module, optimizer, labeled_loader, sup_criterion, num_batches: int, cur_epoch=0,
                 device="cpu", scaler, accumulate_iter: int, **kwargs) -> None:
        kwargs.setdefault("unlabeled_loader", ())
        super().__init__(model=model, optimizer=optimizer, labeled_loader=labeled_loader,
                         sup_criterion=sup_criterion, num_batches=num_batches, cur_epoch=cur_epoch, device=device,
                         scaler=scaler, accumulate_iter=accumulate_iter, **kwargs)

    def _assertion(self):
        assert_transform_freedom(self._labeled_loader, False)

    def configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters = super().configure_meters(meters)
        meters.delete_meter("reg_loss")
        return meters

    def _forward_pass(self, labeled_image, **kwargs):
        return self._model(labeled_image)

    def _batch_update(self, *, cur_batch_num: int, labeled_image, labeled_target, label_group, retain_graph=False,
                      **kwargs):
        self.optimizer_zero(self._optimizer, cur_iter=cur_batch_num)
        with self.autocast:
            label_logits = self.forward_pass(labeled_image=labeled_image)
            sup_loss = _sup_loss(self._sup_criterion, label_logits, labeled_target, self.num_classes)
        self.scale_loss(sup_loss).backward(retain_graph=retain_graph)
        self.optimizer_step(self._optimizer, cur_iter=cur_batch_num)
        if self.on_master:
            with torch.no_grad():
                self.meters["sup_loss"].add(sup_loss.detach())
                self.meters["sup_dice"].add_logits(label_logits, labeled_target, group_name=label_group)
