"""A minimal stand-in for gin-config, used only when the real `gin` package is not importable.

The reference drives `train_rqvae.train` with gin (train_rqvae.py:24, modules/utils.py:18-22).  Its four
config files use a small subset of gin's grammar, which is all this module understands:

    import a.b.c                      -> importlib.import_module (registers enum constants as a side effect)
    include 'other.gin'               -> gin's include: parse that file first (path relative to the working directory, as
                                         gin does, else to the including file's directory or its parent)
    # comment
    scope.name = <python literal>     -> int / float / str / bool / None / list literals
    scope.name = %mod.path.Enum.MEMBER -> constants registered by @constants_from_enum

`configurable` functions look up bindings by their own __name__ (`train.iterations=...`).  Explicit call
arguments win over bindings, as in gin.
"""
from __future__ import annotations

import ast
import functools
import importlib
import inspect
from typing import Any, Callable, Dict

_BINDINGS: Dict[str, Dict[str, Any]] = {}
_CONSTANTS: Dict[str, Any] = {}


def clear_config() -> None:
    _BINDINGS.clear()


def constants_from_enum(cls=None, *, module: str | None = None):
    """Register `module.Class.MEMBER` names so that `%module.Class.MEMBER` resolves in config files."""
    def register(c):
        mod = module or c.__module__
        for member in c:
            _CONSTANTS[f"{mod}.{c.__name__}.{member.name}"] = member
        return c
    return register(cls) if cls is not None else register


def configurable(fn: Callable | None = None, **_kw):
    def wrap(f):
        params = inspect.signature(f).parameters

        @functools.wraps(f)
        def inner(*args, **kwargs):
            bound = dict(_BINDINGS.get(f.__name__, {}))
            unknown = [k for k in bound if k not in params]
            if unknown:
                raise ValueError(f"No parameter(s) {unknown} in configurable '{f.__name__}'")
            positional = list(params)[: len(args)]
            for k in positional:
                bound.pop(k, None)
            bound.update(kwargs)
            return f(*args, **bound)
        return inner
    return wrap(fn) if callable(fn) else wrap


def _resolve_constant(name: str) -> Any:
    if name in _CONSTANTS:
        return _CONSTANTS[name]
    # allow the suffix form (gin matches on the shortest unambiguous suffix)
    hits = [v for k, v in _CONSTANTS.items() if k.endswith("." + name)]
    if len(hits) == 1:
        return hits[0]
    raise ValueError(f"Unknown or ambiguous gin constant %{name}")


def _parse_value(text: str) -> Any:
    text = text.strip()
    if text.startswith("%"):
        return _resolve_constant(text[1:].strip())
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError) as exc:
        raise ValueError(f"ginlite cannot parse value {text!r} (supported: python literals and %constants)") from exc


def parse_config(text: str, base_dir: str | None = None) -> None:
    for lineno, raw in enumerate(text.splitlines(), 1):
        line = raw.split("#", 1)[0].strip() if not ('"' in raw or "'" in raw) else _strip_comment(raw).strip()
        if not line:
            continue
        if line.startswith("import "):
            importlib.import_module(line[len("import "):].strip())
            continue
        if line.startswith("include "):
            import os
            inc = ast.literal_eval(line[len("include "):].strip())
            # gin resolves an include relative to the working directory; this parser also tries the including file's own
            # directory and its parent (so `include 'configs/x.gin'` written for a launch from the package root resolves from
            # anywhere) before it gives up with the list of what it tried
            tried = [inc] if (os.path.isabs(inc) or base_dir is None) else \
                [inc, os.path.join(base_dir, inc), os.path.join(os.path.dirname(base_dir), inc)]
            found = next((t for t in tried if os.path.exists(t)), None)
            if found is None:
                raise FileNotFoundError(f"ginlite: line {lineno}: include {inc!r} not found (tried {tried})")
            parse_config_file(found)
            continue
        if "=" not in line:
            raise ValueError(f"ginlite: line {lineno}: expected 'scope.name = value', got {raw!r}")
        key, value = line.split("=", 1)
        key = key.strip()
        if "." not in key:
            raise ValueError(f"ginlite: line {lineno}: binding key {key!r} has no configurable name")
        scope, name = key.rsplit(".", 1)
        scope = scope.rsplit(".", 1)[-1].rsplit("/", 1)[-1]
        _BINDINGS.setdefault(scope, {})[name] = _parse_value(value)


def _strip_comment(raw: str) -> str:
    out, quote = [], None
    for ch in raw:
        if quote:
            if ch == quote:
                quote = None
        elif ch in "\"'":
            quote = ch
        elif ch == "#":
            break
        out.append(ch)
    return "".join(out)


def parse_config_file(path: str) -> None:
    import os
    with open(path, "r", encoding="utf-8") as fh:
        parse_config(fh.read(), base_dir=os.path.dirname(os.path.abspath(path)))


def query_parameter(key: str) -> Any:
    scope, name = key.rsplit(".", 1)
    return _BINDINGS[scope][name]
