"""Name -> factory registries: the plugin surface neosr's TOML files address.

Behavioural contract mirrored from the reference (neosr/utils/registry.py:8-107):
``register()`` works bare or as a decorator and keys on ``__name__`` (+ ``_suffix``); duplicate
names assert; ``get(name, suffix="neosr")`` falls back to ``name_suffix`` and raises ``KeyError``
("No object named ...") on a miss; ``in``, iteration and ``keys()`` are supported.
"""

from __future__ import annotations

from collections.abc import Callable, Iterator
from typing import Any


class Registry:
    def __init__(self, name: str) -> None:
        self._name = name
        self._obj_map: dict[str, Any] = {}

    # -- registration -------------------------------------------------------------------
    def _add(self, key: str, obj: Any, suffix: str | None) -> None:
        if isinstance(suffix, str):
            key = f"{key}_{suffix}"
        assert key not in self._obj_map, (
            f"An object named '{key}' was already registered in '{self._name}' registry!"
        )
        self._obj_map[key] = obj

    def register(self, obj: Any = None, suffix: str | None = None) -> Any:
        if obj is not None:  # plain call: REGISTRY.register(thing)
            self._add(obj if isinstance(obj, str) else obj.__name__, obj, suffix)
            return None

        def decorate(target: Callable[..., Any]) -> Callable[..., Any]:
            self._add(target.__name__, target, suffix)
            return target

        return decorate

    # -- lookup -------------------------------------------------------------------------
    def get(self, name: str, suffix: str = "neosr") -> Any:
        for key in (name, f"{name}_{suffix}"):
            found = self._obj_map.get(key)
            if found is not None:
                return found
        msg = f"No object named '{name}' found in '{self._name}' registry!"
        raise KeyError(msg)

    def __contains__(self, name: str) -> bool:
        return name in self._obj_map

    def __iter__(self) -> Iterator[tuple[str, Any]]:
        return iter(self._obj_map.items())

    def keys(self):
        return self._obj_map.keys()


DATASET_REGISTRY = Registry("dataset")
ARCH_REGISTRY = Registry("arch")
MODEL_REGISTRY = Registry("model")
LOSS_REGISTRY = Registry("loss")
METRIC_REGISTRY = Registry("metric")
