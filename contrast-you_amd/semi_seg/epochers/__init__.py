from .epocher import EpocherBase, EvalEpocher, FineTuneEpocher, SemiSupervisedEpocher  # noqa: F401
