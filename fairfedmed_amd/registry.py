"""Name -> class registry with the semantics of the reference's Dassl
``Registry`` (Dassl/dassl/utils/registry.py:32-68) and ``build_trainer``
(Dassl/dassl/engine/build.py:14-20; check_availability:
Dassl/dassl/utils/tools.py:170-185): registering a duplicate name raises
KeyError, getting an unknown name raises KeyError, building an unavailable
trainer raises ValueError with a nearest-match hint."""
from __future__ import annotations

import difflib
from typing import Callable, Dict, List, Optional


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map: Dict[str, object] = {}

    def _do_register(self, name: str, obj, force: bool = False) -> None:
        if name in self._obj_map and not force:
            raise KeyError(f'An object named "{name}" was already registered in "{self._name}" registry')
        self._obj_map[name] = obj

    def register(self, obj=None, force: bool = False):
        if obj is None:                       # decorator form
            def wrapper(fn_or_class):
                self._do_register(fn_or_class.__name__, fn_or_class, force=force)
                return fn_or_class
            return wrapper
        self._do_register(obj.__name__, obj, force=force)
        return obj

    def get(self, name: str):
        if name not in self._obj_map:
            raise KeyError(f'Object name "{name}" does not exist in "{self._name}" registry')
        return self._obj_map[name]

    def registered_names(self) -> List[str]:
        return list(self._obj_map.keys())


TRAINER_REGISTRY = Registry("TRAINER")


def check_availability(requested: str, available: List[str]) -> None:
    if requested not in available:
        close = difflib.get_close_matches(requested, available, n=1)
        hint = f" (do you mean '{close[0]}'?)" if close else ""
        raise ValueError(f"The requested one is expected to belong to {available}, but got [{requested}]{hint}")


def build_trainer(cfg):
    """cfg.TRAINER.NAME -> TRAINER_REGISTRY.get(name)(cfg)."""
    avai = TRAINER_REGISTRY.registered_names()
    check_availability(cfg.TRAINER.NAME, avai)
    if getattr(cfg, "VERBOSE", False):
        print(f"Loading trainer: {cfg.TRAINER.NAME}")
    return TRAINER_REGISTRY.get(cfg.TRAINER.NAME)(cfg)
