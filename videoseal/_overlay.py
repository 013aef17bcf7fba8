"""Overlay of the `videoseal` import path over a checkout of facebookresearch/videoseal.

This shim binds the reference's import paths of the embed -> augment -> extract path to the MI355X implementation and defines
nothing else.  train.py:55-72 also imports `videoseal.utils.{optim,dist,logger}`, `videoseal.losses`, `videoseal.data`, ... --
code outside the path (optimizer factories, datasets, the discriminator / perceptual loss networks) that this repo does not
rebuild.  With

    export VIDEOSEAL_REFERENCE_ROOT=/path/to/facebookresearch-videoseal        # the directory that holds train.py and videoseal/

every (sub)package of the shim appends the matching directory of that checkout to its `__path__`, so a module the shim does
NOT define resolves to the checkout's file, while everything the shim defines (models, modules.jnd, augmentation, evals.metrics,
utils.cfg) stays the HIP implementation -- the shim's directory comes first on `__path__`.  Attributes of the reference's
package `__init__` files that callers use directly (`videoseal.utils.bool_inst`, `get_sha`, ...) resolve through
`fallback_getattr`.  Without the variable the shim stays closed: a missing module is an ImportError, as before.
"""
from __future__ import annotations

import importlib.util
import os
import sys
from typing import List, Optional

ENV = "VIDEOSEAL_REFERENCE_ROOT"


def reference_dir(sub: str = "") -> Optional[str]:
    root = os.environ.get(ENV)
    if not root:
        return None
    d = os.path.join(root, "videoseal", *[p for p in sub.split(".") if p])
    return d if os.path.isdir(d) else None


def extend(path: List[str], sub: str = "") -> None:
    """append the checkout's directory for sub-package `sub` ('' = the top-level package) to a package's __path__"""
    d = reference_dir(sub)
    if d and d not in path:
        path.append(d)


def fallback_getattr(pkg_name: str, sub: str):
    """module-level __getattr__ for a shim package: names it does not define come from the checkout's own __init__.py of that package,
    loaded once under a private module name (its relative imports resolve inside the overlaid package)"""
    def __getattr__(name: str):
        if name.startswith("__"):
            raise AttributeError(name)
        d = reference_dir(sub)
        init = os.path.join(d, "__init__.py") if d else None
        if not init or not os.path.isfile(init):
            raise AttributeError(f"module '{pkg_name}' has no attribute '{name}' (the MI355X shim defines the embed / extract path only; "
                                 f"set {ENV} to a reference checkout for the rest)")
        priv = pkg_name + "._reference_init"
        mod = sys.modules.get(priv)
        if mod is None:
            spec = importlib.util.spec_from_file_location(priv, init, submodule_search_locations=None)
            mod = importlib.util.module_from_spec(spec)
            mod.__package__ = pkg_name            # relative imports of the reference's __init__ resolve inside the overlaid package
            sys.modules[priv] = mod
            try:
                spec.loader.exec_module(mod)
            except BaseException:
                sys.modules.pop(priv, None)
                raise
        try:
            return getattr(mod, name)
        except AttributeError:
            raise AttributeError(f"module '{pkg_name}' has no attribute '{name}'") from None
    return __getattr__


def _load_private(priv: str, path: str, package: str, stubs=None):
    """execute the checkout's file `path` once as module `priv` inside package `package`; a third-party module the file needs and this
    environment lacks surfaces as an ImportError that names it (`stubs`: {module name: factory} installed first when absent)"""
    mod = sys.modules.get(priv)
    if mod is not None:
        return mod
    for name, factory in (stubs or {}).items():
        if name not in sys.modules and importlib.util.find_spec(name) is None:
            sys.modules[name] = factory()
    spec = importlib.util.spec_from_file_location(priv, path, submodule_search_locations=None)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = package
    sys.modules[priv] = mod
    try:
        spec.loader.exec_module(mod)
    except ModuleNotFoundError as e:
        sys.modules.pop(priv, None)
        raise ImportError(f"{path} needs the third-party module '{e.name}', which is not installed here") from e
    except BaseException:
        sys.modules.pop(priv, None)
        raise
    return mod


def fallback_module_getattr(mod_name: str, rel_file: str, stubs=None):
    """module-level __getattr__ for a shim MODULE that shadows a file of the checkout (e.g. videoseal/evals/metrics.py): a name the shim
    does not define is taken from the checkout's own file, executed once under a private name; names the shim defines never get here,
    so the path's functions stay this package's"""
    package = mod_name.rsplit(".", 1)[0]

    def __getattr__(name: str):
        if name.startswith("__"):
            raise AttributeError(name)
        root = os.environ.get(ENV)
        path = os.path.join(root, "videoseal", *rel_file.split("/")) if root else None
        if not path or not os.path.isfile(path):
            raise AttributeError(f"module '{mod_name}' has no attribute '{name}' (the MI355X shim defines the embed / extract path only; "
                                 f"set {ENV} to a reference checkout for the rest)")
        mod = _load_private(mod_name + "__reference", path, package, stubs)
        try:
            return getattr(mod, name)
        except AttributeError:
            raise AttributeError(f"module '{mod_name}' has no attribute '{name}'") from None
    return __getattr__
