"""Trainer zoo (semi_seg/trainers/__init__.py:7-15).  `dmt` and `mixup` drive comparison-method
epochers that are outside this build's hot-path scope and raise on use."""
from contrastyou.trainer.base import Trainer as _Trainer

from .pretrain import PretrainDecoderTrainer, PretrainEncoderTrainer  # noqa: F401
from .trainer import FineTuneTrainer, MTTrainer, SemiTrainer  # noqa: F401


def _out_of_scope(name):
    class _Unavailable(_Trainer):  # noqa
        def __init__(self, *a, **k):
            raise NotImplementedError(f"Trainer `{name}` (comparison method) is not part of this build")

    return _Unavailable


trainer_zoo = {
    "semi": SemiTrainer,
    "ft": FineTuneTrainer,
    "pretrain": PretrainEncoderTrainer,
    "pretrain_decoder": PretrainDecoderTrainer,
    "mt": MTTrainer,
    "dmt": _out_of_scope("dmt"),
    "mixup": _out_of_scope("mixup"),
}
