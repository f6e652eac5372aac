"""Import helper: the package directory is named `pumi-pic_amd` (not a valid Python identifier),
so it is loaded by path and registered as `pumipic_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "pumi-pic_amd")


def load():
    if "pumipic_amd" in sys.modules:
        return sys.modules["pumipic_amd"]
    spec = importlib.util.spec_from_file_location(
        "pumipic_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["pumipic_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle():
    """ORACLE -- tests / smoke / bench cpu_baseline only."""
    if "ppo" in sys.modules:
        return sys.modules["ppo"]
    spec = importlib.util.spec_from_file_location("ppo", os.path.join(ROOT, "oracle", "ppo.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ppo"] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle_picpart():
    """ORACLE (PICpart construction / comm arrays) -- tests only."""
    if "ppo_picpart" in sys.modules:
        return sys.modules["ppo_picpart"]
    spec = importlib.util.spec_from_file_location("ppo_picpart", os.path.join(ROOT, "oracle", "ppo_picpart.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ppo_picpart"] = mod
    spec.loader.exec_module(mod)
    return mod
