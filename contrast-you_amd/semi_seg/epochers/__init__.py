from .epocher import EpocherBase, EvalEpocher, FineTuneEpocher, SemiSupervisedEpocher  # noqa: F401
from .pretrain import (PretrainDecoderEpocher, PretrainDecoderEpocherInference,  # noqa: F401
                       PretrainEncoderEpocher)
