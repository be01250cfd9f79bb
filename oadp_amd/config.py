"""Minimal loader for the reference's python-dict config files (``todd.Config.load``):
``_base_`` inheritance with recursive dict merge, attribute access, and ``--override
.a.b.c:value`` (reference README.md:216, oadp/oake/base.py:66-72,119-120)."""
from __future__ import annotations

import ast
import pathlib
from typing import Any


class Config(dict):

    def __getattr__(self, name: str) -> Any:
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = value

    @staticmethod
    def _wrap(obj: Any) -> Any:
        if isinstance(obj, dict):
            return Config({k: Config._wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return type(obj)(Config._wrap(v) for v in obj)
        return obj

    @staticmethod
    def _merge(base: dict, new: dict) -> dict:
        out = dict(base)
        for k, v in new.items():
            if isinstance(v, dict) and isinstance(out.get(k), dict):
                out[k] = Config._merge(out[k], v)
            else:
                out[k] = v
        return out

    @classmethod
    def _load_raw(cls, path: pathlib.Path) -> dict:
        scope: dict[str, Any] = {}
        exec(compile(path.read_text(), str(path), 'exec'), scope)
        # config keys are the public data names of the file: helpers (``_COCO``, ``def _split``), imported
        # modules and functions are not — they would otherwise travel into Validator(**config)
        import types
        cfg = {k: v for k, v in scope.items()
               if (k == '_base_' or not k.startswith('_'))
               and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
        bases = cfg.pop('_base_', [])
        if isinstance(bases, str):
            bases = [bases]
        merged: dict = {}
        for b in bases:
            merged = cls._merge(merged, cls._load_raw(path.parent / b))
        return cls._merge(merged, cfg)

    @classmethod
    def load(cls, path: str | pathlib.Path) -> 'Config':
        return cls._wrap(cls._load_raw(pathlib.Path(path)))

    def override(self, items: dict[str, Any]) -> None:
        """``{'.train.dataloader.dataset.auto_fix': True}`` style keys."""
        for key, value in items.items():
            node: Any = self
            parts = [p for p in key.split('.') if p]
            for p in parts[:-1]:
                node = node.setdefault(p, Config())
            node[parts[-1]] = Config._wrap(value)


def parse_override(pairs: list[str] | None) -> dict[str, Any] | None:
    """``--override .a.b:True .c:3`` -> {'.a.b': True, '.c': 3} (DictAction)."""
    if not pairs:
        return None
    out: dict[str, Any] = {}
    for p in pairs:
        key, _, raw = p.partition(':')
        try:
            out[key] = ast.literal_eval(raw)
        except (ValueError, SyntaxError):
            out[key] = raw
    return out
