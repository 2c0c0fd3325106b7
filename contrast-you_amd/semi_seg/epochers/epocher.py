"""The epochers of the hot path.

    SemiSupervisedEpocher   one step = forward of the labeled batch and of the unlabeled batch with its
                            transformed view, supervised KL loss, the hooks' regularisers, backward,
                            optimizer step
    FineTuneEpocher         the supervised part alone
    EvalEpocher             forward + loss + Dice per scan, no gradients

Classes, constructor kwargs, the order in which hooks are called and the kwargs of
`regularization(**kwargs)` are those of semi_seg/epochers/epocher.py:39-418 in the reference.  What is
different is confined to execution:
  * losses and Dice counts are accumulated on the device, meters read them back at summary time;
  * the supervised loss uses KL_div.from_logits (one fused softmax+KL pass) when the criterion has
    it, instead of softmax -> one_hot -> `unique()` assertion -> KL;
  * the two passes of the two-stage forward run on two HIP streams, or as replayed HIP graphs
    (`_forward_two_streams`, cyhip/graphed.py);
  * progress statistics are sampled every `report_every` batches instead of every batch.
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
from contrastyou.utils.utils import (class_name, disable_tracking_bn_stats, get_dataset, get_lrs_from_optimizer,
                                     get_model)
from cyhip import graphed, ops
from semi_seg.augment import AffineAugment
from semi_seg.epochers.helper import (preprocess_input_with_single_transformation,
                                      preprocess_input_with_twice_transformation)


def assert_transform_freedom(dataloader, is_true):
    """the loaders of a semi-supervised run must apply the SAME random spatial transform to image and
    target (`_total_freedom is False`); pre-training on images alone wants the opposite"""
    assert get_dataset(dataloader).transforms._total_freedom is is_true  # noqa


def _sup_loss(criterion, logits: Tensor, target: Tensor, num_classes: int) -> Tensor:
    """criterion(softmax(logits), one_hot(target)); one fused kernel pass if the criterion can"""
    labels = target.squeeze(1)
    fused = getattr(criterion, "from_logits", None)
    if fused is not None:
        return fused(logits, labels)
    return criterion(logits.softmax(1), class2one_hot(labels, num_classes))


def _scalar(loss):
    return loss.detach() if isinstance(loss, Tensor) else loss


class EpocherBase(_EpocherBase):
    """adds the three hooked stages of a step -- `batch_update`, `forward_pass`, `regularization` --
    each of which runs the hooks' before-callbacks, the epocher's `_<stage>` and the after-callbacks
    (the protocol of epocher.py:39-116)"""

    def __init__(self, *, model: nn.Module, num_batches: int, cur_epoch=0, device="cpu", scaler, **kwargs) -> None:
        super().__init__(model=model, num_batches=num_batches, cur_epoch=cur_epoch, device=device, scaler=scaler,
                         **kwargs)
        self.retain_graph = False
        self.report_every = 50

    num_classes = property(lambda self: get_model(self._model).num_classes)

    def init(self, trainer=None) -> None:
        super().init(trainer=trainer)
        self._assertion()

    def _assertion(self):
        """loader sanity checks of the concrete epocher"""

    def run(self, **kwargs):
        if not self._initialized:
            raise RuntimeError(f"Call {class_name(self)}.init() before {class_name(self)}.run()")
        return super().run(**kwargs)

    def _staged(self, stage: str, **kwargs):
        for h in self._hooks:
            getattr(h, "call_before_" + stage)(**kwargs)
        result = getattr(self, "_" + stage)(**kwargs)
        for h in self._hooks:
            getattr(h, "call_after_" + stage)(**kwargs, result_dict=result)
        return result

    def batch_update(self, **kwargs) -> Optional[Dict[str, Any]]:
        return self._staged("batch_update", **kwargs)

    def forward_pass(self, **kwargs):
        return self._staged("forward_pass", **kwargs)

    def regularization(self, **kwargs):
        return self._staged("regularization", **kwargs)

    def _batch_update(self, **kwargs) -> Optional[Dict[str, Any]]:
        ...

    def _forward_pass(self, **kwargs) -> Any:
        ...

    def _regularization(self, **kwargs):
        return torch.zeros((), dtype=torch.float, device=self._device)

    def _report(self, i: int, last: bool):
        if last or (self.verbose and self.report_every and i % self.report_every == 0):
            self.indicator.set_postfix_statics2(dict(self.meters.statistics()), force_update=last)


class SemiSupervisedEpocher(EpocherBase):
    meter_focus = "semi"

    def __init__(self, *, model: nn.Module, optimizer, labeled_loader, unlabeled_loader, sup_criterion,
                 num_batches: int, cur_epoch=0, device="cpu", two_stage: bool = False, disable_bn: bool = False,
                 scaler, accumulate_iter: int = 1, **kwargs) -> None:
        super().__init__(model=model, num_batches=num_batches, cur_epoch=cur_epoch, device=device, scaler=scaler,
                         accumulate_iter=accumulate_iter)
        self._optimizer, self._sup_criterion = optimizer, sup_criterion
        self._labeled_loader, self._unlabeled_loader = labeled_loader, unlabeled_loader
        self._two_stage, self._disable_bn = two_stage, disable_bn
        self.cur_batch_num = 0
        # the in-step augmentation of the unlabeled view: ranges of epocher.py:226-238
        self._affine_transformer = AffineAugment(scale=(0.8, 1.3), rotation=(-45, 45), translation=(-0.1, 0.1),
                                                 mirror_p=0.9, gamma=(0.5, 2))

    def _assertion(self):
        assert_transform_freedom(self._labeled_loader, False)
        if self._unlabeled_loader is not None:
            assert_transform_freedom(self._unlabeled_loader, False)

    def configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters = super().configure_meters(meters)
        classes = self.num_classes
        meters.register_meter("sup_loss", AverageValueMeter())
        meters.register_meter("sup_dice", UniversalDice(classes, report_axis=list(range(1, classes))))
        meters.register_meter("reg_loss", AverageValueMeter())
        return meters

    def transform_with_seed(self, features, *, mode: str, seed: int):
        """same `seed` => same geometry, whatever the tensor's resolution or channel count"""
        assert mode in {"image", "feature"}, f"mode must be either `image` or `feature`, given {mode}"
        return self._affine_transformer(features, mode=mode, seed=seed)

    @staticmethod
    def _unzip_data(data, device):
        (image, target), (image_ct, _), filename, partition, group = \
            preprocess_input_with_twice_transformation(data, device)
        return (image, image_ct), target, filename, partition, group

    @property
    def _bn_context(self):
        """context for the unlabeled pass: `disable_bn` keeps its batches out of the running statistics"""
        return disable_tracking_bn_stats if self._disable_bn else (lambda model: nullcontext())

    # ---- epoch ------------------------------------------------------------------------------------
    def _run(self, **kwargs):
        self.meters["lr"].add(get_lrs_from_optimizer(self._optimizer))
        self._model.train()
        return self._run_implement()

    def _run_implement(self):
        if len(self._unlabeled_loader) == 0:  # fully supervised: the labeled stream doubles as unlabeled
            self._unlabeled_loader = self._labeled_loader
        batches = zip(self.indicator, self._labeled_loader, self._unlabeled_loader)
        for self.cur_batch_num, labeled_data, unlabeled_data in batches:
            seed = random.randint(0, int(1e7))
            (labeled_image, _), labeled_target, labeled_filename, _, label_group = \
                self._unzip_data(labeled_data, self._device)
            (unlabeled_image, unlabeled_image_cf), _, unlabeled_filename, unl_partition, unl_group = \
                self._unzip_data(unlabeled_data, self._device)
            self.batch_update(
                cur_batch_num=self.cur_batch_num, seed=seed, retain_graph=self.retain_graph,
                labeled_image=labeled_image, labeled_target=labeled_target, labeled_filename=labeled_filename,
                label_group=label_group, unlabeled_image=unlabeled_image,
                unlabeled_image_tf=self.transform_with_seed(unlabeled_image_cf, seed=seed, mode="image"),
                unl_group=unl_group, unl_partition=unl_partition, unlabeled_filename=unlabeled_filename)
            self._report(self.cur_batch_num, self.cur_batch_num == self.num_batches - 1)

    # ---- step -------------------------------------------------------------------------------------
    def _batch_update(self, *, cur_batch_num: int, labeled_image, labeled_target, labeled_filename, label_group,
                      unlabeled_image, unlabeled_image_tf, seed, unl_group, unl_partition, unlabeled_filename,
                      retain_graph=False, **kwargs):
        warp = partial(self.transform_with_seed, seed=seed, mode="feature")
        self.optimizer_zero(self._optimizer, cur_iter=cur_batch_num)
        with self.autocast:
            label_logits, unlabeled_logits, unlabeled_tf_logits = self.forward_pass(
                labeled_image=labeled_image, unlabeled_image=unlabeled_image, unlabeled_image_tf=unlabeled_image_tf)
            unlabeled_logits_tf = warp(unlabeled_logits)
            sup_loss = _sup_loss(self._sup_criterion, label_logits, labeled_target, self.num_classes)
            # NB `label_group` handed to the hooks is the UNLABELED batch's scan ids (as in the reference)
            reg_loss = self.regularization(
                seed=seed, affine_transformer=warp, labeled_image=labeled_image, labeled_target=labeled_target,
                labeled_filename=labeled_filename, unlabeled_image=unlabeled_image,
                unlabeled_image_tf=unlabeled_image_tf, unlabeled_filename=unlabeled_filename,
                unlabeled_tf_logits=unlabeled_tf_logits, unlabeled_logits_tf=unlabeled_logits_tf,
                label_group=unl_group, partition_group=unl_partition)
        self.scale_loss(sup_loss + reg_loss).backward(retain_graph=retain_graph)
        self.optimizer_step(self._optimizer, cur_iter=cur_batch_num)
        if self.on_master:
            with torch.no_grad():
                self.meters["sup_loss"].add(sup_loss.detach())
                self.meters["sup_dice"].add_logits(label_logits, labeled_target, group_name=label_group)
                self.meters["reg_loss"].add(_scalar(reg_loss))

    def _regularization(self, **kwargs):
        if not self._hooks:
            return torch.zeros((), device=self.device, dtype=torch.float)
        total = None  # (`sum()` starts from 0: one `0 + loss` launch per step)
        for h in self._hooks:
            v = h(**kwargs)
            total = v if total is None else total + v
        return total

    # ---- forward ----------------------------------------------------------------------------------
    def _forward_pass(self, labeled_image, unlabeled_image, unlabeled_image_tf):
        n_l, n_unl = len(labeled_image), len(unlabeled_image)
        if not self._two_stage:  # one pass, one set of BN statistics over all three parts
            logits = self._model(torch.cat([labeled_image, unlabeled_image, unlabeled_image_tf], dim=0))
            return torch.split(logits, [n_l, n_unl, n_unl], dim=0)
        views = torch.cat([unlabeled_image, unlabeled_image_tf], dim=0)
        if (ops.TWO_STREAM or graphed.GRAPH_STEP) and labeled_image.is_cuda:
            label_logits, both = self._forward_two_streams(labeled_image, views)
        else:
            label_logits = self._model(labeled_image)
            with self._bn_context(self._model):
                both = self._model(views)
        return (label_logits, *torch.split(both, [n_unl, n_unl], dim=0))

    def _forward_two_streams(self, labeled_image, views):
        """The two passes of the two-stage forward are independent network evaluations (epocher.py:
        351-357) and neither fills the GPU at these batch sizes.  Preferred: both passes, forward and
        backward, replayed as HIP graphs (cyhip/graphed.py; the eager step is host-bound).  Otherwise
        the unlabeled pass is enqueued on a second stream and overlaps with the labeled one (autograd
        replays each pass's backward on its own stream, so the backward passes overlap too).  State
        that is updated in place and shared by the passes -- BN running statistics and counters,
        accumulated BN / head gradients -- is serialised in the reference's order (labeled pass first)
        by cyhip.ops.ordered.  `CY_TWO_STREAM=0` disables both."""
        if hasattr(self._model, "arch_elements"):
            replayed = graphed.two_pass(self._model, self._bn_context, labeled_image, views, self._disable_bn,
                                        self._autocast_dtype if self.use_mixed_train else None)
            if replayed is not None:
                return replayed
        if not ops.TWO_STREAM:
            label_logits = self._model(labeled_image)
            with self._bn_context(self._model):
                return label_logits, self._model(views)
        dev = labeled_image.device
        main, side = torch.cuda.current_stream(dev), ops.side_stream(dev, "pass2")
        side.wait_stream(main)  # inputs, zeroed gradients, the previous step's optimizer update
        ops.note_side_work(side)
        label_logits = self._model(labeled_image)
        with torch.cuda.stream(side), self._bn_context(self._model):
            both = self._model(views)
        main.wait_stream(side)
        both.record_stream(main)
        return label_logits, both


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


class EvalEpocher(EpocherBase):
    meter_focus = "eval"

    def __init__(self, *, model: nn.Module, loader, sup_criterion, cur_epoch=0, device="cpu", scaler,
                 accumulate_iter: int) -> None:
        super().__init__(model=model, num_batches=len(loader), cur_epoch=cur_epoch, device=device, scaler=scaler,
                         accumulate_iter=accumulate_iter)
        self._loader, self._sup_criterion = loader, sup_criterion

    def configure_meters(self, meters: MeterInterface) -> MeterInterface:
        meters = super().configure_meters(meters)
        classes = self.num_classes
        meters.register_meter("loss", AverageValueMeter())
        meters.register_meter("dice", UniversalDice(classes, report_axis=list(range(1, classes))))
        return meters

    def get_score(self) -> float:
        return self.meters["dice"].summary()["DSC_mean"]

    @staticmethod
    def _unzip_data(data, device):
        return preprocess_input_with_single_transformation(data, device)

    def _run(self, **kwargs):
        self._model.eval()
        return self._run_implement()

    @torch.no_grad()
    def _run_implement(self):
        for i, batch in zip(self.indicator, self._loader):
            image, target, file_names, _, group = self._unzip_data(batch, self._device)
            self.batch_update(eval_img=image, eval_target=target, eval_group=group, file_names=file_names)
            self._report(i, i == self.num_batches - 1)

    def _batch_update(self, *, eval_img, eval_target, eval_group, file_names):
        with self.autocast:
            logits = self._model(eval_img)
            loss = _sup_loss(self._sup_criterion, logits, eval_target, self.num_classes)
        self.meters["loss"].add(loss.detach())
        self.meters["dice"].add_logits(logits, eval_target, group_name=eval_group)


class InferenceEpocher(EvalEpocher):
    """`InferenceEpocher` (semi_seg/epochers/epocher.py:174-204): the evaluation epoch of `Trainer.inference`,
    with an optional prediction saver.  The reference's saver writes PNGs through its dataset tooling and its
    extra "ASD" surface meter needs the un-vendored `medpy`; here the predictions (class-index maps) are written
    as one `.npy` per input file under `<save_dir>/predictions/` and the Dice / loss meters are the ones of
    `EvalEpocher` (the surface meter is out of scope)."""
    meter_focus = "infer"

    def __init__(self, *, model: nn.Module, loader, sup_criterion, cur_epoch=0, device="cpu", scaler,
                 accumulate_iter: int, enable_prediction_saver: bool = True, save_dir=None) -> None:
        super().__init__(model=model, loader=loader, sup_criterion=sup_criterion, cur_epoch=cur_epoch, device=device,
                         scaler=scaler, accumulate_iter=accumulate_iter)
        self.enable_prediction_saver = enable_prediction_saver
        self._save_dir = save_dir

    def _prediction_dir(self):
        base = self._save_dir
        if base is None and getattr(self, "trainer", None) is not None:
            base = self.trainer.absolute_save_dir
        if base is None:
            return None
        import os
        out = os.path.join(str(base), "predictions")
        os.makedirs(out, exist_ok=True)
        return out

    def _batch_update(self, *, eval_img, eval_target, eval_group, file_names):
        with self.autocast:
            logits = self._model(eval_img)
            loss = _sup_loss(self._sup_criterion, logits, eval_target, self.num_classes)
        self.meters["loss"].add(loss.detach())
        self.meters["dice"].add_logits(logits, eval_target, group_name=eval_group)
        out_dir = self._prediction_dir() if self.enable_prediction_saver else None
        if out_dir is not None:
            import os
            import numpy as np
            pred = logits.argmax(1).to(torch.uint8).cpu().numpy()
            for name, p in zip(file_names, pred):
                np.save(os.path.join(out_dir, f"{os.path.basename(str(name))}.npy"), p)
