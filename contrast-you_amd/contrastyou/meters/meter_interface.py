"""Grouped meter registry of one epoch (contrastyou/meters/meter_interface.py:8-102)."""
from __future__ import annotations

from collections import OrderedDict, defaultdict
from contextlib import contextmanager
from typing import Dict, List

from .metric import Metric


class MeterInterface:

    def __init__(self, default_focus="tra") -> None:
        self._group_bank: Dict[str, Dict[str, Metric]] = defaultdict(OrderedDict)
        self._focus = default_focus

    def register_meter(self, name: str, meter: Metric):
        if not isinstance(meter, Metric):
            raise KeyError(meter)
        group = self._group_bank[self._focus]
        if name in group:
            raise KeyError(f"{name} exists in {self._focus}")
        group[name] = meter

    def delete_meter(self, name: str):
        group = self._meters_of(self._focus)
        if name not in group:
            raise KeyError(name)
        del group[name]
        if not group:
            del self._group_bank[self._focus]

    def delete_meters(self, name_list: List[str]):
        for n in name_list:
            self.delete_meter(n)

    def add(self, meter_name, *args, **kwargs):
        self[meter_name].add(*args, **kwargs)

    def reset(self) -> None:
        for g in self._group_bank.values():
            for m in g.values():
                m.reset()

    def join(self):
        for g in self._group_bank.values():
            for m in g.values():
                m.join()

    def _meters_of(self, group_name: str):
        if group_name not in self._group_bank:
            raise KeyError(f"{group_name} not in {self.__class__.__name__}: ({', '.join(self.groups())})")
        return self._group_bank[group_name]

    def groups(self):
        return list(self._group_bank.keys())

    @property
    def cur_focus(self):
        return self._focus

    @contextmanager
    def focus_on(self, group_name: str):
        prev, self._focus = self._focus, group_name
        try:
            yield
        finally:
            self._focus = prev

    def statistics(self):
        """(group, {meter: summary}) pairs; groups starting with `_` are private"""
        for g in self.groups():
            if not g.startswith("_"):
                yield g, {k: m.summary() for k, m in self._group_bank[g].items()}

    def __enter__(self):
        self.reset()

    def __exit__(self, *args, **kwargs):
        self.join()

    def __getitem__(self, meter_name: str) -> Metric:
        group = self._meters_of(self._focus)
        if meter_name not in group:
            raise KeyError(f"{meter_name} not in {self._focus} group: ({', '.join(group)})")
        return group[meter_name]
