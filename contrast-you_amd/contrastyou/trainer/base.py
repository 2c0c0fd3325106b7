"""`Trainer`: the epoch loop around the epochers.

It owns the model, the optimizer built from `config["Optim"]` (model parameters in group 0, the
registered hooks' parameters in group 1), the warm-up + cosine schedule built from
`config["Scheduler"]`, the trainer hooks, a per-epoch csv `Storage`, and the checkpoint files:

    trainer = SomeTrainer(model=..., criterion=..., config=..., save_dir=..., ...)
    with trainer.register_hook(*hooks):          # must come before init()
        trainer.init()
        trainer.resume_from_path(ckpt_dir)       # optional
        trainer.start_training()                 # per epoch: train, eval val/test, lr step, last/best.pth

Interface and checkpoint layout follow the reference (contrastyou/trainer/base.py:27-191 with its
_hooks / _io / _amp / _ddp mixins): a checkpoint is
`{"module_state": {"_model.*", "_inference_model.*", "_hooks.N.*"}, "buffer_state": {_save_dir,
_max_epoch, _num_batches, config, _cur_epoch, _start_epoch, _best_score}, "other_state":
{_optimizer, _scheduler, scaler, _storage}}`.

Build-side choices: optimizer names resolve in `contrastyou.optim` (RAdam = the fused flat-buffer
RAdam); `enable_scale=True` means bf16 autocast with a pass-through scaler (`BF16Scaler`), False a
disabled GradScaler; there is no tensorboard writer (not installed here) -- metrics always go to
`storage.csv`; checkpoints are read with `torch.load(weights_only=True)`.
"""
from __future__ import annotations

import os
from abc import abstractmethod
from contextlib import contextmanager, nullcontext
from pathlib import Path
from typing import Any, Dict, List, Optional

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

_NOT_OPTIMIZER_ARGS = {"name", "pre_lr", "ft_lr"}  # keys of config["Optim"] that are not constructor kwargs
_CKPT_SUFFIXES = (".pth", ".pt")


def _plain(obj):
    """nested mappings / sequences -> plain dicts / lists of python scalars, so that the stored config
    is yaml-dumpable and loadable with weights_only=True whatever config class produced it"""
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if hasattr(obj, "items"):
        return {str(k): _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    return obj.item() if hasattr(obj, "item") else str(obj)


class Trainer(DDPMixin, ModuleBase):
    RUN_PATH = os.environ.get("CONTRASTYOU_RUN_PATH", str(Path.cwd() / "runs"))
    activate_hooks = True  # FineTune-style trainers switch the hooks off

    def __init__(self, *, model: nn.Module, criterion, tra_loader, val_loader, save_dir: str, max_epoch: int = 100,
                 num_batches: int = 100, device="cpu", config: Dict[str, Any], enable_scale: bool = False,
                 accumulate_iter: int = 1, amp_dtype="bf16", **kwargs) -> None:
        super().__init__()
        # modules (checkpointed under module_state)
        self._model = self._inference_model = model
        self._hooks = nn.ModuleList()
        # plain references
        self.register_non_trackable_buffer("_criterion", criterion)
        self._tra_loader, self._val_loader = tra_loader, val_loader
        self._device, self._config = device, config
        # persistent python state (checkpointed under buffer_state)
        self._save_dir = Buffer(str(save_dir))
        self._max_epoch = Buffer(int(max_epoch))
        self._num_batches = Buffer(int(num_batches))
        self.config = Buffer(_plain(config))
        self._cur_epoch = Buffer(0)
        self._start_epoch = Buffer(0)
        self._best_score = Buffer(0.0)
        # objects with their own state_dict (checkpointed under other_state)
        # enable_scale: mixed precision.  amp_dtype "bf16" (default of this build): bf16 autocast, pass-through
        # scaler; "fp16": the reference's mode, fp16 autocast + torch GradScaler (contrastyou/amp/amp.py:13-45)
        if enable_scale and amp_dtype in ("fp16", "float16", torch.float16):
            self.scaler = torch.amp.GradScaler("cuda", enabled=True)
        else:
            self.scaler = BF16Scaler() if enable_scale else torch.amp.GradScaler("cuda", enabled=False)
        self._storage = Storage(save_dir=self.save_dir)
        self._optimizer: Optional[torch.optim.Optimizer] = None
        self._scheduler: Optional[GradualWarmupScheduler] = None
        self._writer = None
        self._enable_scale, self._accumulate_iter = enable_scale, accumulate_iter
        self._initialized = False
        if config is not None:
            self.dump_config(self._persist_buffer["config"])

    device = property(lambda self: self._device)
    save_dir = property(lambda self: str(self._save_dir))
    absolute_save_dir = property(lambda self: str(self._save_dir))
    relative_save_dir = property(lambda self: str(Path(str(self._save_dir)).relative_to(self.RUN_PATH)))
    success = property(lambda self: ".success" in os.listdir(str(self._save_dir)))
    inference_model = property(lambda self: self._inference_model)

    # ======================================================================== hooks + set-up
    @contextmanager
    def register_hook(self, *hook: TrainerHook):
        """adopt trainer hooks; their parameters join the optimizer built by the following init()"""
        if self._initialized:
            raise RuntimeError("`register_hook must be called before `init()``")
        for h in hook:
            h.to(self.device)
            h.register_trainer(self)
            self._hooks.append(h)
        for h in self._hooks:
            h.after_initialize()
        yield
        for h in hook:
            h.close()

    def init(self):
        if self._initialized:
            raise RuntimeError(f"{self.__class__.__name__} has been initialized.")
        self._optimizer = self._init_optimizer()
        self._scheduler = self._init_scheduler(self._optimizer, scheduler_params=self._config.get("Scheduler", None))
        self._initialized = True

    def _init_optimizer(self) -> torch.optim.Optimizer:
        spec = self._config["Optim"]
        kwargs = {k: v for k, v in spec.items() if k not in _NOT_OPTIMIZER_ARGS}
        factory = getattr(optim, spec["name"])
        optimizer = factory(params=[p for p in self._model.parameters() if p.requires_grad], **kwargs)
        hook_params: List[nn.Parameter] = [p for h in self._hooks for p in h.parameters()]
        if hook_params:
            optimizer.add_param_group(dict(params=hook_params, **kwargs))
        return optimizer

    def _init_scheduler(self, optimizer, scheduler_params) -> Optional[GradualWarmupScheduler]:
        if scheduler_params is None:
            return None
        warmup = int(scheduler_params["warmup_max"])
        cosine = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=self._max_epoch - warmup, eta_min=1e-7)
        return GradualWarmupScheduler(optimizer, scheduler_params["multiplier"], total_epoch=warmup,
                                      after_scheduler=cosine)

    # ======================================================================== the epoch loop
    def start_training(self, **kwargs):
        if not self._initialized:
            raise RuntimeError(f"{self.__class__.__name__} should call `init()` first")
        self.to(self.device)
        if hasattr(self._optimizer, "broadcast_buffers"):  # data parallel: replicas start from rank 0's BN statistics too
            self._optimizer.broadcast_buffers(self._model)
        self._start_training(**kwargs)
        if self.on_master:
            Path(self.absolute_save_dir, ".success").touch()

    def _start_training(self, **kwargs):
        first = max(self._cur_epoch + 1, self._start_epoch)
        for self._cur_epoch in range(first, self._max_epoch + 1):
            score, improved = 0.0, False
            with self._storage:  # writes storage.csv when the block ends
                train_metrics = self.tra_epoch()
                if self.on_master:
                    eval_metrics, score = self.eval_epoch(model=self.inference_model, loader=self._val_loader)
                    test_metrics, _ = self.eval_epoch(model=self.inference_model, loader=self._test_loader)
                    self._storage.add_from_meter_interface(tra=train_metrics, val=eval_metrics, test=test_metrics,
                                                           epoch=self._cur_epoch)
                if self._scheduler is not None:
                    self._scheduler.step()
                if score > self._best_score:
                    self._best_score, improved = score, True
            if self.on_master:
                self.save_to(save_name="last.pth")
                if improved:
                    self.save_to(save_name="best.pth")

    def tra_epoch(self, **kwargs):
        return self._run_tra_epoch(self._create_initialized_tra_epoch(**kwargs))

    def eval_epoch(self, *, model, loader, **kwargs):
        return self._run_eval_epoch(self._create_initialized_eval_epoch(model=model, loader=loader, **kwargs))

    def _run_tra_epoch(self, epocher: EpocherBase):
        epoch_hooks = [h() for h in self._hooks] if (self.activate_hooks and len(self._hooks) > 0) else []
        with epocher.register_hook(*epoch_hooks) if epoch_hooks else nullcontext():
            epocher.run()
        return epocher.get_metric()

    def _run_eval_epoch(self, epocher):
        epocher.run()
        return epocher.get_metric(), epocher.get_score()

    @abstractmethod
    def _create_initialized_tra_epoch(self, **kwargs) -> EpocherBase:
        ...

    @abstractmethod
    def _create_initialized_eval_epoch(self, *, model, loader, **kwargs) -> EpocherBase:
        ...

    # ---- which model is evaluated (mean teacher evaluates the teacher) -----------------------------
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

    # ======================================================================== files
    def save_to(self, *, save_dir: str = None, save_name: str):
        assert Path(save_name).suffix in _CKPT_SUFFIXES, save_name
        target = Path(save_dir or self.save_dir)
        target.mkdir(parents=True, exist_ok=True)
        safe_save(self.state_dict(), str(target / save_name))

    def load_state_dict_from_path(self, path: str, name="last.pth", strict=True) -> None:
        where = Path(path)
        assert where.exists(), path
        if where.is_dir():
            where = where / name
        if not (where.is_file() and where.suffix in _CKPT_SUFFIXES):
            raise FileNotFoundError(where)
        self.load_state_dict(torch.load(str(where), map_location="cpu", weights_only=True), strict)

    def resume_from_path(self, path: str, name="last.pth", strict=True):
        return self.load_state_dict_from_path(str(path), name, strict)

    def resume_from_checkpoint(self, checkpoint: Dict[str, Dict], strict=True):
        self.load_state_dict(checkpoint, strict=strict)

    def dump_config(self, config, path=None, save_name="config.yaml"):
        """write the config next to the checkpoints; a second run in the same directory gets
        config_<k>.yaml instead of overwriting"""
        import yaml
        folder = Path(path) if path else Path(self.save_dir)
        if not folder.is_absolute():
            folder = Path(self.RUN_PATH) / folder
        folder.mkdir(parents=True, exist_ok=True)
        if (folder / save_name).exists():
            save_name = f"{Path(save_name).stem}_{len(list(folder.glob('*.yaml')))}.yaml"
        with open(folder / save_name, "w") as f:
            yaml.safe_dump(_plain(config), f)
