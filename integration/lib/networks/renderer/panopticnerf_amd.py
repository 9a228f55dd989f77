"""Reference-side renderer plugin: resolved by make_renderer(cfg, network) through
cfg.renderer_module / cfg.renderer_path and instantiated as `Renderer(network)`."""
from lib.config import cfg

from panopticnerf_amd.renderer import Renderer as _Renderer


class Renderer(_Renderer):
    def __init__(self, net):
        super().__init__(net, cfg)      # render(batch) -> dict, the same call as the reference's renderer
