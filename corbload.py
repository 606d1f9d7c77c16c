"""Imports the hyphenated product directory `corb-slam_amd/` as the python package `corb_slam_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def load_pkg():
    name = "corb_slam_amd"
    if name in sys.modules:
        return sys.modules[name]
    d = os.path.join(ROOT, "corb-slam_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
