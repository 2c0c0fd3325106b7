"""`SemiTrainer`, `FineTuneTrainer`, `MTTrainer` (semi_seg/trainers/trainer.py:27-167, 199-204):
which epocher runs a training epoch and how evaluation is wired (teacher model for mean teacher).
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Any, Dict, Type

from torch import nn

from contrastyou.trainer.base import Trainer
from semi_seg.epochers.epocher import (EpocherBase, EvalEpocher, FineTuneEpocher, InferenceEpocher,
                                       SemiSupervisedEpocher)
from semi_seg.hooks import MeanTeacherTrainerHook


class SemiTrainer(Trainer):
    activate_hooks = True

    def __init__(self, *, model: nn.Module, labeled_loader, unlabeled_loader, val_loader, test_loader, criterion,
                 save_dir: str, max_epoch: int = 100, num_batches: int = 100, device="cpu", disable_bn: bool,
                 two_stage: bool, config: Dict[str, Any], enable_scale=True, accumulate_iter: int = 1,
                 **kwargs) -> None:
        super().__init__(model=model, criterion=criterion, tra_loader=None, val_loader=val_loader, save_dir=save_dir,
                         max_epoch=max_epoch, num_batches=num_batches, device=device, config=config,
                         enable_scale=enable_scale, accumulate_iter=accumulate_iter, **kwargs)
        del self._tra_loader
        self._labeled_loader = labeled_loader
        self._unlabeled_loader = unlabeled_loader
        self._val_loader = val_loader
        self._test_loader = test_loader
        self._disable_bn = disable_bn
        self._two_stage = two_stage

    @property
    def train_epocher(self) -> Type[EpocherBase]:
        return SemiSupervisedEpocher

    def _create_initialized_tra_epoch(self, **kwargs) -> EpocherBase:
        epocher = self.train_epocher(
            model=self._model, optimizer=self._optimizer, labeled_loader=self._labeled_loader,
            unlabeled_loader=self._unlabeled_loader, sup_criterion=self._criterion, num_batches=self._num_batches,
            cur_epoch=self._cur_epoch, device=self._device, two_stage=self._two_stage, disable_bn=self._disable_bn,
            scaler=self.scaler, accumulate_iter=self._accumulate_iter)
        epocher.init(trainer=self)
        return epocher

    def _create_initialized_eval_epoch(self, *, model, loader, **kwargs) -> EpocherBase:
        epocher = EvalEpocher(model=model, loader=loader, sup_criterion=self._criterion, cur_epoch=self._cur_epoch,
                              device=self._device, scaler=self.scaler, accumulate_iter=self._accumulate_iter)
        epocher.init(trainer=self)
        return epocher

    def inference(self, checkpoint_path: str = None, checkpoint_name: str = "best.pth", save_dir: str = None,
                  enable_prediction_save=False, **kwargs):
        """load `checkpoint_name`, evaluate the test loader, write inference_result.json
        (trainer.py:71-113; per-scan re-batching of the loader is the data layer's job)"""
        checkpoint_path = checkpoint_path or self.absolute_save_dir
        if not os.path.isabs(checkpoint_path):
            raise ValueError(f"`checkpoint_path` must be an absolute path, given {checkpoint_path}")
        self.resume_from_path(str(checkpoint_path), name=checkpoint_name)
        save_dir = save_dir or self.absolute_save_dir
        # (trainer.py:107-123 `_inference`: an InferenceEpocher over the test loader)
        epocher = InferenceEpocher(model=self._model, loader=self._test_loader, sup_criterion=self._criterion,
                                   cur_epoch=self._cur_epoch, device=self._device, scaler=self.scaler,
                                   accumulate_iter=self._accumulate_iter,
                                   enable_prediction_saver=enable_prediction_save, save_dir=save_dir)
        epocher.init(trainer=self)
        epocher.run()
        metrics, score = epocher.get_metric(), epocher.get_score()
        Path(save_dir).mkdir(exist_ok=True, parents=True)
        with open(os.path.join(save_dir, "inference_result.json"), "w") as f:
            json.dump(metrics, f, indent=4)
        return metrics, score


class FineTuneTrainer(SemiTrainer):
    activate_hooks = False

    @property
    def train_epocher(self) -> Type[EpocherBase]:
        return FineTuneEpocher


class MTTrainer(SemiTrainer):
    """validation/test run on the TEACHER of the (single) mean-teacher hook (trainer.py:125-167)"""

    def eval_epoch(self, *, model, loader, **kwargs):
        mt_hook = [h for h in self._hooks if isinstance(h, MeanTeacherTrainerHook)]
        assert len(mt_hook) == 1, mt_hook
        return super().eval_epoch(model=mt_hook[0].teacher_model, loader=loader, **kwargs)
