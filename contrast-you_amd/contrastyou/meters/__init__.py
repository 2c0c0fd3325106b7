from .averagemeter import AverageValueDictionaryMeter, AverageValueListMeter, AverageValueMeter  # noqa: F401
from .general_dice_meter import UniversalDice  # noqa: F401
from .meter_interface import MeterInterface  # noqa: F401
from .metric import Metric  # noqa: F401
from .storage import Storage  # noqa: F401
