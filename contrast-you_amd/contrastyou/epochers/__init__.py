from .base import EpocherBase, TrainerNotSetError  # noqa: F401
