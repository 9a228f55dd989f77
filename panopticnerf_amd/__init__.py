"""panopticnerf_amd -- MI355X (gfx950) implementation of PanopticNeRF's render_rays hot path
behind the reference's make_network / make_renderer plugin surface (BASELINE.json north_star).

Only what the path needs lives here: csrc/ (HIP kernels + C-ABI, built into libpnr.so), the
ctypes binding, the torch-tensor op front ends, and the host-side mirrors of the reference's
Network / Renderer interfaces.  Importing the package does not load the library; the first op
does, and raises if it is missing (no CPU fallback)."""
from .network import NeRF, Network, make_network  # noqa: F401
from .renderer import Renderer, make_renderer  # noqa: F401

from .losses import NetworkWrapper  # noqa: F401,E402

__all__ = ["NeRF", "Network", "make_network", "Renderer", "make_renderer", "NetworkWrapper"]
