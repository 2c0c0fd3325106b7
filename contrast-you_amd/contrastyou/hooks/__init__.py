from .base import (CombineEpochHook, CombineTrainerHook, EpocherHook, HookNameExistError,  # noqa: F401
                   HookNotInitializedError, TrainerHook)
