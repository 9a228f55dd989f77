"""Stand-in for the reference's plugin factories -- TEST HARNESS ONLY.

The reference (branch `panopticnerf`, not in /root/reference -- README.md:13 only points to it) resolves every plugin
from config strings:  make_network(cfg) -> imp.load_source(cfg.network_module, cfg.network_path).Network()  (zero
arguments),  make_renderer(cfg, net) -> ....Renderer(net),  the trainer wraps the network as NetworkWrapper(net), the
evaluator is Evaluator().  The plugins read the reference's global config with `from lib.config import cfg`.  This
module provides exactly that much of the reference: a stub `lib.config` and the four factories, so that the files under
integration/ are loaded BY PATH the way a checkout of the reference would load them."""
import importlib.util
import os
import sys
import types
from types import SimpleNamespace as NS

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTEGRATION = os.path.join(ROOT, "integration")


def load_cfg(**overrides):
    """integration/configs/panopticnerf_amd.yaml (+ overrides) as an attribute-style config, installed as lib.config.cfg."""
    with open(os.path.join(INTEGRATION, "configs", "panopticnerf_amd.yaml")) as f:
        d = yaml.safe_load(f)
    d.update(overrides)
    cfg = NS(**d)
    lib = types.ModuleType("lib")
    lib.__path__ = []                      # a package, so that `from lib.config import cfg` resolves
    config = types.ModuleType("lib.config")
    config.cfg = cfg
    lib.config = config
    sys.modules["lib"], sys.modules["lib.config"] = lib, config
    return cfg


def _load_source(module, path):
    """imp.load_source(module, path) (what the reference calls; `imp` is gone from newer Pythons, same semantics)."""
    path = os.path.join(INTEGRATION, path)
    import warnings
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            import imp                      # Python <= 3.11
            return imp.load_source(module, path)
    except ImportError:
        spec = importlib.util.spec_from_file_location(module, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[module] = mod
        spec.loader.exec_module(mod)
        return mod


def make_network(cfg):
    return _load_source(cfg.network_module, cfg.network_path).Network()


def make_renderer(cfg, network):
    return _load_source(cfg.renderer_module, cfg.renderer_path).Renderer(network)


def make_network_wrapper(cfg, network):
    return _load_source(cfg.trainer_module, cfg.trainer_path).NetworkWrapper(network)


def make_evaluator(cfg):
    return _load_source(cfg.evaluator_module, cfg.evaluator_path).Evaluator()
