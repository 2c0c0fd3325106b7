from .epocher import (EpocherBase, EvalEpocher, FineTuneEpocher, InferenceEpocher,  # noqa: F401
                      SemiSupervisedEpocher)
from .pretrain import (PretrainDecoderEpocher, PretrainDecoderEpocherInference,  # noqa: F401
                       PretrainEncoderEpocher)
