from .consistency import ConsistencyTrainerHook  # noqa: F401
from .creator import (create_consistency_hook, create_discrete_mi_consistency_hook,  # noqa: F401
                      create_discrete_mi_hooks, create_iid_segmentation_hook, create_infonce_hooks,
                      create_mt_hook, create_sp_infonce_hooks, create_superpixel_hooks, feature_until_from_hooks)
from .discretemi import DiscreteMITrainHook  # noqa: F401
from .infonce import (INFONCEHook, PScheduler, SelfPacedINFONCEHook, SuperPixelInfoNCEHook,  # noqa: F401
                      region_extractor)
from .midl import IIDSegmentationTrainerHook  # noqa: F401
from .mt import EMAUpdater, MeanTeacherTrainerHook  # noqa: F401
