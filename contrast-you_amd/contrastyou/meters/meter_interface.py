"""The meter registry of one epoch (interface of contrastyou/meters/meter_interface.py:8-102).

Meters live in named groups; exactly one group is "in focus" at a time and `interface[name]`,
`register_meter`, `delete_meter` act on it.  The epocher's own meters sit in its `meter_focus`
group, every hook gets the group of its name (the epocher switches focus around each hook call).
`with interface:` resets all meters on entry and joins them on exit; `statistics()` yields
`(group, {meter: summary})` for every public group (a leading underscore hides a group).
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import Dict, Iterator, List, Tuple

from .metric import Metric


class MeterInterface:

    def __init__(self, default_focus="tra") -> None:
        self._focus = default_focus
        self._group_bank: Dict[str, Dict[str, Metric]] = {}

    # ---- focus --------------------------------------------------------------------------------
    @property
    def cur_focus(self) -> str:
        return self._focus

    @contextmanager
    def focus_on(self, group_name: str):
        previous, self._focus = self._focus, group_name
        try:
            yield
        finally:
            self._focus = previous

    def groups(self) -> List[str]:
        return list(self._group_bank)

    def _meters_of(self, group_name: str) -> Dict[str, Metric]:
        try:
            return self._group_bank[group_name]
        except KeyError:
            raise KeyError(f"{group_name} not in {type(self).__name__}: ({', '.join(self.groups())})") from None

    # ---- registration (in the focused group) ----------------------------------------------------
    def register_meter(self, name: str, meter: Metric):
        if not isinstance(meter, Metric):
            raise KeyError(meter)
        group = self._group_bank.setdefault(self._focus, {})
        if name in group:
            raise KeyError(f"{name} exists in {self._focus}")
        group[name] = meter

    def delete_meter(self, name: str):
        group = self._meters_of(self._focus)
        if name not in group:
            raise KeyError(name)
        group.pop(name)
        if not group:  # an emptied group disappears from the statistics
            self._group_bank.pop(self._focus)

    def delete_meters(self, name_list: List[str]):
        for name in name_list:
            self.delete_meter(name)

    def __getitem__(self, meter_name: str) -> Metric:
        group = self._meters_of(self._focus)
        try:
            return group[meter_name]
        except KeyError:
            raise KeyError(f"{meter_name} not in {self._focus} group: ({', '.join(group)})") from None

    def add(self, meter_name, *args, **kwargs):
        self[meter_name].add(*args, **kwargs)

    # ---- whole-registry operations ---------------------------------------------------------------
    def _all_meters(self) -> Iterator[Metric]:
        for group in self._group_bank.values():
            yield from group.values()

    def reset(self) -> None:
        for meter in self._all_meters():
            meter.reset()

    def join(self):
        for meter in self._all_meters():
            meter.join()

    def statistics(self) -> Iterator[Tuple[str, Dict[str, object]]]:
        for name, group in self._group_bank.items():
            if not name.startswith("_"):
                yield name, {key: meter.summary() for key, meter in group.items()}

    def __enter__(self):
        self.reset()

    def __exit__(self, *exc):
        self.join()
