"""Trainers for contrastive pre-training (what semi_seg/trainers/pretrain.py:26-129 provides):
every epoch is a `Pretrain*Epocher` over a stream of two-view unlabeled batches with the forward
pass stopped at `forward_until`; nothing is evaluated, only `last.pth` is written.

Data note: the reference rebuilds a scan/partition-grouped contrastive DataLoader from the dataset
classes (`_get_contrastive_dataloader`); datasets are outside this build's scope, so the stream is
passed in (`contrastive_loader=`, optional `monitor_loader=` for `inference()`); without one the
unlabeled loader is used.  `ContrastiveLoaderParams` must still be present in the config, as in the
reference, so that a config written for it is accepted or rejected identically.
"""
from __future__ import annotations

from contextlib import nullcontext
from typing import Optional, Type

from contrastyou.arch import UNet
from semi_seg.epochers.pretrain import (PretrainDecoderEpocher, PretrainDecoderEpocherInference,
                                        PretrainEncoderEpocher)
from semi_seg.trainers.trainer import SemiTrainer

__all__ = ["PretrainEncoderTrainer", "PretrainDecoderTrainer"]

_LAST_BLOCK = tuple(UNet.decoder_names)[-1]


class _PretrainTrainer(SemiTrainer):
    epocher_class: Type = None  # set by the public subclasses

    def __init__(self, *, contrastive_loader=None, monitor_loader=None, **kwargs):
        super().__init__(**kwargs)
        if "ContrastiveLoaderParams" not in self._config:
            raise RuntimeError("`ContrastiveLoaderParams` should be found in config, given \n"
                               f"`{', '.join(self._config.keys())}`")
        stream = self._unlabeled_loader if contrastive_loader is None else contrastive_loader
        self._contrastive_loader = iter(stream)  # consumed across epochs, never restarted
        self._monitor_loader = monitor_loader
        self._inference_until: Optional[str] = None

    train_epocher = property(lambda self: self.epocher_class)

    # ---- how far the forward pass goes ------------------------------------------------------------
    @property
    def forward_until(self) -> str:
        return self._inference_until or _LAST_BLOCK

    @forward_until.setter
    def forward_until(self, block: Optional[str]):
        if block == "all":
            block = None
        if block is not None:
            assert block in self._model.arch_elements, block
        self._inference_until = block

    # ---- epochs -----------------------------------------------------------------------------------
    def _epocher(self, cls, *, loader, num_batches, accumulate_iter):
        epocher = cls(model=self._model, optimizer=self._optimizer, labeled_loader=self._labeled_loader,
                      unlabeled_loader=self._unlabeled_loader, sup_criterion=self._criterion,
                      cur_epoch=self._cur_epoch, device=self._device, two_stage=False, disable_bn=False,
                      scaler=self.scaler, chain_dataloader=loader, inference_until=self._inference_until,
                      num_batches=num_batches, accumulate_iter=accumulate_iter)
        epocher.init(trainer=self)
        return epocher

    def _create_initialized_tra_epoch(self, **kwargs):
        return self._epocher(self.epocher_class, loader=self._contrastive_loader, num_batches=self._num_batches,
                             accumulate_iter=self._accumulate_iter)

    def _start_training(self, **kwargs):
        first = max(self._cur_epoch + 1, self._start_epoch)
        for self._cur_epoch in range(first, self._max_epoch + 1):
            with self._storage:  # csv after every epoch
                metrics = self.tra_epoch()
                if self.on_master:
                    self._storage.add_from_meter_interface(pre_tra=metrics, epoch=self._cur_epoch)
                if self._scheduler is not None:
                    self._scheduler.step()
            if self.on_master:
                self.save_to(save_name="last.pth")

    # ---- monitoring pass: the hooks' loss without augmentation, gradients or updates ----------------
    def inference(self, **kwargs):
        epocher = self._epocher(PretrainDecoderEpocherInference, loader=self._monitor_loader,
                                num_batches=len(self._monitor_loader), accumulate_iter=1)
        hooks = [h() for h in self._hooks] if (self.activate_hooks and len(self._hooks) > 0) else []
        self._model.eval()
        try:
            with epocher.register_hook(*hooks) if hooks else nullcontext():
                epocher.run()
        finally:
            self._model.train()
        return epocher.get_metric()


class PretrainEncoderTrainer(_PretrainTrainer):
    epocher_class = PretrainEncoderEpocher


class PretrainDecoderTrainer(_PretrainTrainer):
    epocher_class = PretrainDecoderEpocher
