"""Reference-side evaluator plugin: resolved by make_evaluator through cfg.evaluator_module /
cfg.evaluator_path and instantiated as `Evaluator()`."""
from lib.config import cfg

from panopticnerf_amd.evaluate import Evaluator as _Evaluator


class Evaluator(_Evaluator):
    def __init__(self):
        super().__init__(cfg, is_thing=getattr(cfg, "thing_classes", None))
