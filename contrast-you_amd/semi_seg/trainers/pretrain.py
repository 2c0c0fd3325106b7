"""Pretrain trainers (semi_seg/trainers/pretrain.py:26-129): contrastive pre-training epochs on
a stream of unlabeled two-view batches, forward stopped at `forward_until`; no evaluation, only
`last.pth`.

The reference rebuilds a scan/partition-grouped contrastive DataLoader from the dataset
(`_get_contrastive_dataloader`); the data layer is outside this build's scope, so the loader is
passed in (`contrastive_loader=`, `monitor_loader=`) and defaults to the unlabeled loader.
`ContrastiveLoaderParams` must still be present in the config, as in the reference.
"""
from __future__ import annotations

from contextlib import nullcontext
from typing import Type, Union

from contrastyou.arch import UNet
from semi_seg.epochers.pretrain import (PretrainDecoderEpocher, PretrainDecoderEpocherInference,
                                        PretrainEncoderEpocher)
from semi_seg.trainers.trainer import SemiTrainer

__all__ = ["PretrainEncoderTrainer", "PretrainDecoderTrainer"]


class _PretrainTrainerMixin:

    def __init__(self, *, contrastive_loader=None, monitor_loader=None, **kwargs):
        super().__init__(**kwargs)
        if "ContrastiveLoaderParams" not in self._config:
            raise RuntimeError("`ContrastiveLoaderParams` should be found in config, given \n"
                               f"`{', '.join(self._config.keys())}`")
        self._contrastive_loader = iter(contrastive_loader if contrastive_loader is not None
                                        else self._unlabeled_loader)
        self._monitor_loader = monitor_loader
        self._inference_until = None

    @property
    def forward_until(self) -> str:
        return list(UNet.decoder_names)[-1] if self._inference_until is None else self._inference_until

    @forward_until.setter
    def forward_until(self, forward_until: Union[str, None]):
        if isinstance(forward_until, str):
            if forward_until == "all":
                self._inference_until = None
                return
            assert forward_until in self._model.arch_elements, forward_until
        self._inference_until = forward_until

    def _start_training(self, **kwargs):
        start_epoch = max(self._cur_epoch + 1, self._start_epoch)
        for self._cur_epoch in range(start_epoch, self._max_epoch + 1):
            with self._storage:
                train_metrics = self.tra_epoch()
                if self.on_master:
                    self._storage.add_from_meter_interface(pre_tra=train_metrics, epoch=self._cur_epoch)
                if self._scheduler is not None:
                    self._scheduler.step()
            if self.on_master:
                self.save_to(save_name="last.pth")

    def _epocher_kwargs(self):
        return dict(model=self._model, optimizer=self._optimizer, labeled_loader=self._labeled_loader,
                    unlabeled_loader=self._unlabeled_loader, sup_criterion=self._criterion,
                    cur_epoch=self._cur_epoch, device=self._device, two_stage=False, disable_bn=False,
                    inference_until=self._inference_until, scaler=self.scaler)

    def _create_initialized_tra_epoch(self, **kwargs):
        epocher = self.train_epocher(chain_dataloader=self._contrastive_loader, num_batches=self._num_batches,
                                     accumulate_iter=self._accumulate_iter, **self._epocher_kwargs())
        epocher.init(trainer=self)
        return epocher


class _PretrainInferenceMixin:

    def _inference(self, *, monitor_dataloader, **kwargs):
        epocher = PretrainDecoderEpocherInference(chain_dataloader=monitor_dataloader,
                                                  num_batches=len(monitor_dataloader), accumulate_iter=1,
                                                  **self._epocher_kwargs())
        epocher.init(trainer=self)
        use_hook = self.activate_hooks and len(self._hooks) > 0
        with epocher.register_hook(*[h() for h in self._hooks]) if use_hook else nullcontext():
            epocher.run()
        return epocher.get_metric()

    def inference(self, **kwargs):
        self._model.eval()
        try:
            return self._inference(monitor_dataloader=self._monitor_loader)
        finally:
            self._model.train()


class PretrainEncoderTrainer(_PretrainInferenceMixin, _PretrainTrainerMixin, SemiTrainer):
    @property
    def train_epocher(self) -> Type:
        return PretrainEncoderEpocher


class PretrainDecoderTrainer(_PretrainInferenceMixin, _PretrainTrainerMixin, SemiTrainer):
    @property
    def train_epocher(self) -> Type:
        return PretrainDecoderEpocher
