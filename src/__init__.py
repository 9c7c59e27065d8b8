"""`src` -- the reference's package name (ref:setup.py:3-7, ref:tests/conftest.py:1-6) as an alias of
`two_tower_models_amd`: code written against the reference's layout,

    from src.two_tower_base_retrieval import TwoTowerBaseRetrieval
    from src.baseline_mips_module import BaselineMIPSModule

imports the MI355X implementation unchanged.  Nothing is defined here: each `src.<module>` IS the package's module of
the same name (one object, registered under both names)."""
import importlib
import sys

_MODULES = ("baseline_mips_module", "two_tower_base_retrieval", "two_tower_with_user_history_encoder",
            "two_tower_with_debiasing", "two_tower_with_position_debiased_weights", "two_tower_with_user_debiased_weights",
            "user_history_encoder")
for _name in _MODULES:
    _mod = importlib.import_module(f"two_tower_models_amd.{_name}")
    sys.modules[f"{__name__}.{_name}"] = _mod
    setattr(sys.modules[__name__], _name, _mod)
