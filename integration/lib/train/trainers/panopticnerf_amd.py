"""Reference-side trainer plugin: resolved by make_trainer through cfg.trainer_module / cfg.trainer_path;
the trainer wraps the network as `NetworkWrapper(network)` and calls
`output, loss, loss_stats, image_stats = wrapper(batch)`.

The wrapper is an ordinary nn.Module whose parameters reach the HIP kernels as autograd inputs, so the
reference trainer's `DistributedDataParallel(wrapper, device_ids=[local_rank])` works unchanged
(tests/test_integration.py::test_ddp_wraps_the_wrapper_unchanged)."""
from lib.config import cfg

from panopticnerf_amd.losses import NetworkWrapper as _Wrapper


class NetworkWrapper(_Wrapper):
    def __init__(self, net):
        super().__init__(net, cfg)
