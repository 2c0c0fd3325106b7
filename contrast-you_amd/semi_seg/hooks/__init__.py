from .consistency import ConsistencyTrainerHook  # noqa: F401
from .creator import (create_consistency_hook, create_infonce_hooks, create_mt_hook,  # noqa: F401
                      feature_until_from_hooks)
from .infonce import INFONCEHook  # noqa: F401
from .mt import EMAUpdater, MeanTeacherTrainerHook  # noqa: F401
