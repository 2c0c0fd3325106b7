"""`Storage`: per-epoch history of every meter summary, written to `storage.csv` after each epoch
(contrastyou/meters/storage_interface.py:19-101).  Same `put/put_group/add_from_meter_interface/
get/summary/to_csv` API; the state is a plain nested dict {name: {epoch: {key: value}}} so that
checkpoints load with `torch.load(weights_only=True)` (the reference pickles container objects).
"""
from __future__ import annotations

import csv
from collections import OrderedDict
from pathlib import Path
from typing import Dict, List

__all__ = ["Storage"]


def _scalar(v):
    if hasattr(v, "item"):
        return v.item()
    return v


class Storage:

    def __init__(self, save_dir, csv_name="storage.csv") -> None:
        self._storage: "OrderedDict[str, OrderedDict[int, Dict[str, float]]]" = OrderedDict()
        self._next_epoch: Dict[str, int] = {}
        self._csv_name = csv_name
        self._save_dir = str(save_dir)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.to_csv()

    def put(self, name: str, value, epoch=None, prefix="", postfix=""):
        key = prefix + name + postfix
        hist = self._storage.setdefault(key, OrderedDict())
        cur = epoch if epoch else self._next_epoch.get(key, 0)
        if not isinstance(value, dict):
            value = {"": value}
        hist[int(cur)] = {str(k): _flat(v) for k, v in value.items()}
        self._next_epoch[key] = int(cur) + 1

    def put_group(self, group_name: str, epoch_result: Dict, epoch=None, sep="/"):
        assert isinstance(group_name, str), group_name
        for k, v in (epoch_result or {}).items():
            self.put(group_name + sep + k, v, epoch)

    def add_from_meter_interface(self, *, epoch: int, **kwargs):
        for k, groups in kwargs.items():
            for g, group_result in dict(groups).items():
                self.put_group(group_name=k + "/" + g, epoch_result=group_result, epoch=epoch)

    def get(self, name, epoch=None):
        assert name in self._storage, name
        return self._storage[name] if epoch is None else self._storage[name][epoch]

    def summary(self):
        """(sorted epochs, column names, rows)"""
        cols: List[str] = []
        for name, hist in self._storage.items():
            for rec in hist.values():
                for k in rec:
                    c = name + ("/" + k if k else "")
                    if c not in cols:
                        cols.append(c)
        epochs = sorted({e for hist in self._storage.values() for e in hist})
        rows = []
        for e in epochs:
            row = {}
            for name, hist in self._storage.items():
                for k, v in hist.get(e, {}).items():
                    row[name + ("/" + k if k else "")] = v
            rows.append(row)
        return epochs, cols, rows

    @property
    def meter_names(self) -> List[str]:
        return list(self._storage.keys())

    @property
    def storage(self):
        return self._storage

    def state_dict(self):
        return {k: {int(e): dict(r) for e, r in h.items()} for k, h in self._storage.items()}

    def load_state_dict(self, state_dict):
        if not isinstance(state_dict, dict):
            return
        self._storage = OrderedDict()
        for k, h in state_dict.items():
            if isinstance(h, dict):
                self._storage[k] = OrderedDict((int(e), dict(r)) for e, r in h.items())
                self._next_epoch[k] = max(self._storage[k], default=-1) + 1

    def to_csv(self):
        path = Path(self._save_dir)
        path.mkdir(exist_ok=True, parents=True)
        epochs, cols, rows = self.summary()
        with open(path / self._csv_name, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow([""] + cols)
            for e, row in zip(epochs, rows):
                w.writerow([e] + [row.get(c, "") for c in cols])


def _flat(v):
    v = _scalar(v)
    if isinstance(v, (list, tuple)):
        return [_scalar(x) for x in v]
    return v
