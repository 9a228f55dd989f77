"""Reference-side network plugin: resolved by make_network(cfg) through cfg.network_module /
cfg.network_path and instantiated as `Network()` with NO arguments, so it reads the
reference's global config itself -- exactly like the renderer adapter does."""
from lib.config import cfg

from panopticnerf_amd.network import Network as _Network


class Network(_Network):
    def __init__(self):
        super().__init__(cfg)           # D, W, skips, xyz_res, view_res, num_classes, num_instances,
                                        # N_importance | cascade_samples, precision
