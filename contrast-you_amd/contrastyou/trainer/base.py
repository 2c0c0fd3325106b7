"""`Trainer`: owns model, optimizer, scheduler and trainer hooks; runs
`tra_epoch -> eval_epoch(val) -> eval_epoch(test) -> scheduler.step -> save last/best` per epoch.

Interface parity with contrastyou/trainer/base.py:27-191 (+ _hooks.py, _io.py, _amp.py, _ddp.py):
same constructor kwargs, `register_hook` (context manager, must precede `init()`), `init()`,
`start_training()`, `tra_epoch()/eval_epoch()`, `save_to()/load_state_dict_from_path()/
resume_from_path()`, `inference_model`/`switch_inference_model`, and the checkpoint schema
{"module_state": {"_model.*", "_hooks.N.*"}, "buffer_state": {_save_dir, _max_epoch, _num_batches,
config, _cur_epoch, _start_epoch, _best_score}, "other_state": {_optimizer, _scheduler, scaler,
_storage}} so `last.pth`/`best.pth` interchange with the reference.

Build-side differences: the optimizer named in config["Optim"] resolves in `contrastyou.optim`
(RAdam = FusedRAdam over flat buffers); the scaler is `BF16Scaler` when `enable_scale` (bf16
autocast, nothing to scale) and a disabled GradScaler otherwise; the tensorboard writer is
optional (absent in this image) and metrics always go to `Storage` (csv per epoch).
"""
from __future__ import annotations

import os
from abc import abstractmethod
from contextlib import contextmanager, nullcontext
from itertools import chain
from pathlib import Path
from typing import Any, Dict, Optional

import torch
from torch import nn

from .. import optim
from ..amp import BF16Scaler, DDPMixin
from ..epochers.base import EpocherBase
from ..hooks.base import TrainerHook
from ..meters.storage import Storage
from ..nn import Buffer, ModuleBase
from ..optim import GradualWarmupScheduler
from ._utils import safe_save

_OPTIM_SKIP = ("name", "pre_lr", "ft_lr")


class Trainer(DDPMixin, ModuleBase):
    RUN_PATH = os.environ.get("CONTRASTYOU_RUN_PATH", str(Path.cwd() / "runs"))
    activate_hooks = True

    def __init__(self, *, model: nn.Module, criterion, tra_loader, val_loader, save_dir: str, max_epoch: int = 100,
                 num_batches: int = 100, device="cpu", config: Dict[str, Any], enable_scale: bool = False,
                 accumulate_iter: int = 1, **kwargs) -> None:
        super().__init__()
        self._initialized = False
        self._hooks = nn.ModuleList()
        self._model = self._inference_model = model
        self.register_non_trackable_buffer("_criterion", criterion)
        self._tra_loader = tra_loader
        self._val_loader = val_loader
        self._save_dir = Buffer(str(save_dir))
        self._max_epoch = Buffer(int(max_epoch))
        self._num_batches = Buffer(int(num_batches))
        self._device = device
        self.config = Buffer(_plain(config))
        self._config = config

        self._enable_scale = enable_scale
        self._accumulate_iter = accumulate_iter
        self.scaler = BF16Scaler() if enable_scale else torch.amp.GradScaler("cuda", enabled=False)

        if config is not None:
            self.dump_config(self._persist_buffer["config"])
        self._storage = Storage(save_dir=self.save_dir)
        self._writer = None  # tensorboard is optional; see contrastyou.writer in the reference

        self._optimizer = None
        self._scheduler = None
        self._cur_epoch = Buffer(0)
        self._start_epoch = Buffer(0)
        self._best_score = Buffer(0.0)

    # ---- hooks (trainer/_hooks.py:21-42) ----------------------------------------------------
    @contextmanager
    def register_hook(self, *hook: TrainerHook):
        if self._initialized:
            raise RuntimeError("`register_hook must be called before `init()``")
        for h in hook:
            self._hooks.append(h)
            h.to(self.device)
            h.register_trainer(self)
        for h in self._hooks:
            h.after_initialize()
        yield
        for h in hook:
            h.close()

    # ---- optimizer / scheduler (trainer/base.py:59-89) --------------------------------------
    def init(self):
        if self._initialized:
            raise RuntimeError(f"{self.__class__.__name__} has been initialized.")
        self._optimizer = self._init_optimizer()
        self._scheduler = self._init_scheduler(self._optimizer, scheduler_params=self._config.get("Scheduler", None))
        self._initialized = True

    def _init_optimizer(self) -> torch.optim.Optimizer:
        params = self._config["Optim"]
        kw = {k: v for k, v in params.items() if k not in _OPTIM_SKIP}
        optimizer = optim.__dict__[params["name"]](
            params=[p for p in self._model.parameters() if p.requires_grad], **kw)
        hook_params = list(chain(*(x.parameters() for x in self._hooks)))
        if hook_params:
            optimizer.add_param_group({"params": hook_params, **kw})
        return optimizer

    def _init_scheduler(self, optimizer, scheduler_params) -> Optional[GradualWarmupScheduler]:
        if scheduler_params is None:
            return None
        cosine = torch.optim.lr_scheduler.CosineAnnealingLR(
            optimizer, T_max=self._max_epoch - int(scheduler_params["warmup_max"]), eta_min=1e-7)
        return GradualWarmupScheduler(optimizer, scheduler_params["multiplier"],
                                      total_epoch=scheduler_params["warmup_max"], after_scheduler=cosine)

    # ---- the epoch loop (trainer/base.py:91-125) --------------------------------------------
    def start_training(self, **kwargs):
        if not self._initialized:
            raise RuntimeError(f"{self.__class__.__name__} should call `init()` first")
        self.to(self.device)
        self._start_training(**kwargs)
        if self.on_master:
            Path(self.absolute_save_dir, ".success").touch()

    def _start_training(self, **kwargs):
        start_epoch = max(self._cur_epoch + 1, self._start_epoch)
        for self._cur_epoch in range(start_epoch, self._max_epoch + 1):
            cur_score = 0.0
            with self._storage:
                train_metrics = self.tra_epoch()
                if self.on_master:
                    eval_metrics, cur_score = self.eval_epoch(model=self.inference_model, loader=self._val_loader)
                    test_metrics, _ = self.eval_epoch(model=self.inference_model, loader=self._test_loader)
                    self._storage.add_from_meter_interface(tra=train_metrics, val=eval_metrics, test=test_metrics,
                                                           epoch=self._cur_epoch)
                if self._scheduler is not None:
                    self._scheduler.step()
                best_case_sofa = self._best_score < cur_score
                if best_case_sofa:
                    self._best_score = cur_score
            if self.on_master:
                self.save_to(save_name="last.pth")
                if best_case_sofa:
                    self.save_to(save_name="best.pth")

    def tra_epoch(self, **kwargs):
        epocher = self._create_initialized_tra_epoch(**kwargs)
        return self._run_tra_epoch(epocher)

    def _run_tra_epoch(self, epocher: EpocherBase):
        use_hook = self.activate_hooks and len(self._hooks) > 0
        with epocher.register_hook(*[h() for h in self._hooks]) if use_hook else nullcontext():
            epocher.run()
        return epocher.get_metric()

    @abstractmethod
    def _create_initialized_tra_epoch(self, **kwargs) -> EpocherBase:
        ...

    def eval_epoch(self, *, model, loader, **kwargs):
        epocher = self._create_initialized_eval_epoch(model=model, loader=loader, **kwargs)
        return self._run_eval_epoch(epocher)

    @abstractmethod
    def _create_initialized_eval_epoch(self, *, model, loader, **kwargs) -> EpocherBase:
        ...

    def _run_eval_epoch(self, epocher):
        epocher.run()
        return epocher.get_metric(), epocher.get_score()

    # ---- inference model switch (trainer/base.py:155-169) -----------------------------------
    @property
    def inference_model(self):
        return self._inference_model

    def set_model4inference(self, model: nn.Module):
        self._inference_model = model

    @contextmanager
    def switch_inference_model(self, model: nn.Module):
        previous = self.inference_model
        self.set_model4inference(model)
        try:
            yield
        finally:
            self.set_model4inference(previous)

    # ---- io (trainer/_io.py:24-68) ----------------------------------------------------------
    def load_state_dict_from_path(self, path: str, name="last.pth", strict=True) -> None:
        path_ = Path(path)
        assert path_.exists(), path
        if path_.is_dir() and (path_ / name).exists():
            path_ = path_ / name
        elif not (path_.is_file() and path_.suffix in (".pth", ".pt")):
            raise FileNotFoundError(path_)
        state_dict = torch.load(str(path_), map_location="cpu", weights_only=True)
        self.load_state_dict(state_dict, strict)

    def save_to(self, *, save_dir: str = None, save_name: str):
        assert Path(save_name).suffix in (".pth", ".pt"), save_name
        save_dir_ = Path(save_dir or self.save_dir)
        save_dir_.mkdir(parents=True, exist_ok=True)
        safe_save(self.state_dict(), str(save_dir_ / save_name))

    def resume_from_checkpoint(self, checkpoint: Dict[str, Dict], strict=True):
        self.load_state_dict(checkpoint, strict=strict)

    def resume_from_path(self, path: str, name="last.pth", strict=True):
        return self.load_state_dict_from_path(str(path), name, strict)

    def dump_config(self, config, path=None, save_name="config.yaml"):
        import yaml
        path_ = Path(path) if path else Path(self.save_dir)
        if not path_.is_absolute():
            path_ = Path(self.RUN_PATH) / path_
        path_.mkdir(parents=True, exist_ok=True)
        if (path_ / save_name).exists():
            save_name = f"{save_name.split('.')[0]}_{len(sorted(path_.glob('*.yaml')))}.yaml"
        with open(path_ / save_name, "w") as f:
            yaml.safe_dump(_plain(config), f)

    @property
    def save_dir(self) -> str:
        return str(self._save_dir)

    @property
    def absolute_save_dir(self) -> str:
        return self.save_dir

    @property
    def relative_save_dir(self):
        return str(Path(self.absolute_save_dir).relative_to(self.RUN_PATH))

    @property
    def success(self):
        return ".success" in os.listdir(self.absolute_save_dir)

    @property
    def device(self):
        return self._device


def _plain(obj):
    """nested mappings/sequences -> plain dict/list of python scalars (yaml- and weights_only-safe)"""
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if hasattr(obj, "items"):
        return {str(k): _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    if hasattr(obj, "item"):
        return obj.item()
    return str(obj)
