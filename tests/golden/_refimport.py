"""Bind the REAL reference's top-level packages for the golden generators (build container only).

The repo carries import-path shims with the same top-level names as the reference (``utils/``, ``models/``, ``Train/``,
``Test/``).  ``/root/reference/utils`` has no ``__init__.py`` (namespace package), and a regular package anywhere on
``sys.path`` beats a namespace package, so ``sys.path`` order alone cannot make ``import utils`` mean the reference.
``bind_reference()`` therefore registers every reference top-level package in ``sys.modules`` by explicit file
location before anything is imported, and ``assert_reference()`` proves where a module came from.
"""
import importlib.util
import os
import sys
import types

REF = "/root/reference"
PACKAGES = ("utils", "models", "Train", "Test", "data")


def bind_reference(ref=REF):
    if not os.path.isdir(ref):
        raise SystemExit(f"{ref} is absent: golden fixtures can only be generated in the build container")
    sys.dont_write_bytecode = True          # /root/reference is read-only by policy
    for pkg in PACKAGES:
        for k in [k for k in sys.modules if k == pkg or k.startswith(pkg + ".")]:
            del sys.modules[k]
        pdir = os.path.join(ref, pkg)
        init = os.path.join(pdir, "__init__.py")
        if os.path.exists(init):
            spec = importlib.util.spec_from_file_location(pkg, init, submodule_search_locations=[pdir])
            mod = importlib.util.module_from_spec(spec)
            sys.modules[pkg] = mod
            spec.loader.exec_module(mod)
        else:                               # namespace package upstream (utils/): a plain module with a fixed __path__
            mod = types.ModuleType(pkg)
            mod.__path__ = [pdir]
            mod.__file__ = os.path.join(pdir, "")        # marks the origin for assert_reference
            sys.modules[pkg] = mod
    if ref not in sys.path:
        sys.path.insert(0, ref)


def assert_reference(*modules, ref=REF):
    """Every given module (or class / function) must have been loaded from the reference tree."""
    for m in modules:
        mod = sys.modules[m.__module__] if not isinstance(m, types.ModuleType) else m
        f = os.path.realpath(getattr(mod, "__file__", "") or "")
        if not f.startswith(os.path.realpath(ref) + os.sep):
            raise AssertionError(f"{mod.__name__} was loaded from {f!r}, not from {ref}")


def unbind_reference():
    """Drop the reference's packages again (used by the regeneration test so later tests see the repo's shims)."""
    for pkg in PACKAGES:
        for k in [k for k in sys.modules if k == pkg or k.startswith(pkg + ".")]:
            del sys.modules[k]
    while REF in sys.path:
        sys.path.remove(REF)
