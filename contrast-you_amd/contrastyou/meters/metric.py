"""The meter protocol used by MeterInterface: `add` feeds a meter, `summary` reports it, `reset`
clears it, `join`/`close` are life-cycle no-ops by default (interface of the reference's
contrastyou/meters/metric.py).  Concrete meters fill in `reset`, `_add` and `_summary`."""
from __future__ import annotations


class Metric:

    def __init__(self, **_unused) -> None:
        self.__armed = True  # a subclass that forgets super().__init__() is caught on first use

    # -- to be provided by concrete meters --------------------------------------------------
    def reset(self):
        raise NotImplementedError(type(self).__name__ + ".reset")

    def _add(self, *values, **named):
        raise NotImplementedError(type(self).__name__ + "._add")

    def _summary(self):
        raise NotImplementedError(type(self).__name__ + "._summary")

    # -- public surface ---------------------------------------------------------------------
    def add(self, *values, **named):
        if not getattr(self, "_Metric__armed", False):
            raise AssertionError(f"{type(self).__name__}.__init__ must call Metric.__init__")
        return self._add(*values, **named)

    def summary(self):
        return self._summary()

    def join(self):
        """hook for meters that collect from worker processes; nothing to do here"""

    def close(self):
        """release resources; nothing to do here"""
