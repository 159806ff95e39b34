"""Import-path compatibility: `from PhysDock import PhysDock, PhysDockConfig` resolves to this package.

The reference's drivers bind the model by import (reference `redocking.py:9-12`, `screening.py:9-12`):

    from PhysDock.utils.import_weights import import_state_dict
    from PhysDock import PhysDock, PhysDockConfig
    from PhysDock.utils.tensor_utils import weighted_rigid_align

`install()` makes those statements hand out the MI355X classes WITHOUT editing the driver:

* the reference package is importable (the normal deployment: its data pipeline `PhysDock.data.*` is still needed) ->
  the real modules stay, and the four names of the sampling path are re-bound on them
  (`PhysDock.PhysDock`, `PhysDock.models.model.PhysDock`, `PhysDock.PhysDockConfig`, `PhysDock.configs.PhysDockConfig`,
  `PhysDock.utils.tensor_utils.weighted_rigid_align`, `PhysDock.utils.import_weights.import_state_dict`);
* it is not -> synthetic modules of those names are registered in `sys.modules` (enough for a caller that only samples).

Run a reference driver unchanged:   python -m physdock_amd.compat redocking.py -i ... -f ... --enable_physics_correction
(or `import physdock_amd.compat; physdock_amd.compat.install()` at the top of any launcher / sitecustomize).
Nothing here touches the HIP library: importing stays CPU-safe, the first kernel launch loads it.
"""
import importlib
import importlib.util
import os
import runpy
import sys
import types

#: dotted module -> {attribute: getter of the replacement}
_BINDINGS = {
    "PhysDock": ("PhysDock", "PhysDockConfig"),
    "PhysDock.models.model": ("PhysDock",),
    "PhysDock.configs": ("PhysDockConfig",),
    "PhysDock.utils.tensor_utils": ("weighted_rigid_align",),
    "PhysDock.utils.import_weights": ("import_state_dict", "import_unicore_ckpt"),
}

_installed = None


def _replacements():
    import physdock_amd as pa
    return {"PhysDock": pa.PhysDock, "PhysDockConfig": pa.PhysDockConfig, "weighted_rigid_align": pa.weighted_rigid_align,
            "import_state_dict": pa.import_state_dict, "import_unicore_ckpt": pa.import_unicore_ckpt}


def _reference_present():
    if "PhysDock" in sys.modules and getattr(sys.modules["PhysDock"], "__physdock_amd_synthetic__", False):
        return False
    try:
        return importlib.util.find_spec("PhysDock") is not None
    except (ImportError, ValueError):
        return False


def install(force_synthetic=False):
    """Bind the reference's import paths of the sampling hot path to physdock_amd.  Idempotent.
    Returns "overlay" (reference package present: names re-bound on its modules) or "synthetic"."""
    global _installed
    if _installed:
        return _installed
    new = _replacements()
    if not force_synthetic and _reference_present():
        for modname, names in _BINDINGS.items():
            mod = importlib.import_module(modname)           # a broken reference install should fail loudly, not half-bind
            for n in names:
                if n in new and (hasattr(mod, n) or modname == "PhysDock"):
                    setattr(mod, n, new[n])
        _installed = "overlay"
        return _installed
    for modname, names in _BINDINGS.items():
        parts = modname.split(".")
        for i in range(1, len(parts) + 1):                   # parents first: PhysDock, PhysDock.models, PhysDock.models.model
            name = ".".join(parts[:i])
            if name not in sys.modules:
                m = types.ModuleType(name)
                m.__path__ = []                              # a package: sub-imports resolve through sys.modules
                m.__physdock_amd_synthetic__ = True
                sys.modules[name] = m
                if i > 1:
                    setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], m)
        for n in names:
            setattr(sys.modules[modname], n, new[n])
    _installed = "synthetic"
    return _installed


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        sys.stderr.write("usage: python -m physdock_amd.compat <reference driver .py> [its arguments]\n")
        return 2
    # the driver's own directory first, as `python redocking.py` would have it (runpy.run_path does not add it for a plain file):
    # the reference checkout beside the script must be found BEFORE install() decides between patching it and synthesising modules
    sys.path.insert(0, os.path.dirname(os.path.abspath(argv[0])))
    mode = install()
    sys.stderr.write(f"physdock_amd.compat: PhysDock import paths bound ({mode})\n")
    if mode == "synthetic":
        sys.stderr.write("physdock_amd.compat: no PhysDock checkout beside the script or on sys.path - only PhysDock.models.model, "
                         "PhysDock.utils.* and the top-level names are bound; a driver that imports PhysDock.data will fail\n")
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
