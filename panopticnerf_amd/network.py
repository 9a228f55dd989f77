"""Network / make_network -- the host-side mirror of the reference's network plugin
(lib/networks/<name>/network.py + make_network; the reference
source is not in the mount, so names follow BASELINE.json's north_star and canonical NeRF).

The module OWNS the parameters (an ordinary nn.Module with the canonical NeRF state_dict
names, so reference checkpoints map key for key -- SURVEY.md 8f-3) but does not evaluate
them with torch: Renderer hands the packed bf16/fp32 MFMA-fragment image of each NeRF to the
fused HIP kernel.  `packed(level, device)` (re)builds that image when parameters change.
"""
import torch
import torch.nn as nn

from . import ops


def _get(cfg, name, default):
    return getattr(cfg, name, default) if cfg is not None else default


class NeRF(nn.Module):
    """Parameter container of one NeRF MLP (coarse or fine).

    pts_linears[i] (i<D): gamma(x)->W, W->W, with [gamma(x), h] concatenated after layer `skip`;
    alpha_linear W->1; feature_linear W->W; views_linears[0] (W+gamma(d))->W/2; rgb_linear W/2->3;
    semantic_linears / instance_linears: W -> W/2 -> n_sem / n_inst (head_depth 2) or W -> n (head_depth 1), reading the trunk
    output (head_tap 'trunk') or the feature_linear output ('feature')."""

    def __init__(self, D=8, W=256, skip=4, xyz_L=10, dir_L=4, n_sem=0, n_inst=0, head_tap="trunk", head_depth=2):
        super().__init__()
        self.D, self.W, self.skip, self.xyz_L, self.dir_L = D, W, skip, xyz_L, dir_L
        if head_tap not in ("trunk", "feature") or int(head_depth) not in (1, 2):
            raise ValueError("head_tap must be 'trunk' or 'feature' and head_depth 1 or 2 (got %r, %r)" % (head_tap, head_depth))
        self.head_tap, self.head_depth = head_tap, int(head_depth)
        self.n_sem, self.n_inst, self.head_W = n_sem, n_inst, W // 2
        ex, ed = 3 + 6 * xyz_L, 3 + 6 * dir_L
        self.pts_linears = nn.ModuleList(
            [nn.Linear(ex, W)] + [nn.Linear(W + ex if (i - 1) == skip else W, W) for i in range(1, D)])
        self.alpha_linear = nn.Linear(W, 1)
        self.feature_linear = nn.Linear(W, W)
        self.views_linears = nn.ModuleList([nn.Linear(W + ed, W // 2)])
        self.rgb_linear = nn.Linear(W // 2, 3)
        head = (lambda n: [nn.Linear(W, n)]) if self.head_depth == 1 else (lambda n: [nn.Linear(W, W // 2), nn.Linear(W // 2, n)])
        if n_sem:
            self.semantic_linears = nn.ModuleList(head(n_sem))
        if n_inst:
            self.instance_linears = nn.ModuleList(head(n_inst))

    def desc(self, precision):
        return ops.make_desc(self.D, self.W, self.skip, self.xyz_L, self.dir_L, self.n_sem, self.n_inst,
                             self.head_W, precision, self.head_tap, self.head_depth)


class Network(nn.Module):
    """Coarse (`nerf_0`) and fine (`nerf_1`) NeRFs, as the reference's Network holds them."""

    def __init__(self, cfg=None):
        super().__init__()
        D, W = _get(cfg, "D", 8), _get(cfg, "W", 256)
        skips = _get(cfg, "skips", [4])
        skip = skips[0] if len(skips) else -1
        if skip >= D - 1:
            skip = -1
        kw = dict(D=D, W=W, skip=skip, xyz_L=_get(cfg, "xyz_res", 10), dir_L=_get(cfg, "view_res", 4),
                  n_sem=_get(cfg, "num_classes", 0), n_inst=_get(cfg, "num_instances", 0),
                  # SURVEY.md 9 item 4 as config switches: where the heads tap the network, and how deep they are
                  head_tap=_get(cfg, "head_tap", "trunk"), head_depth=_get(cfg, "head_depth", 2))
        self.precision = _get(cfg, "precision", "bf16")
        # the same key fallback as Renderer: N_importance, else cascade_samples
        self.N_importance = _get(cfg, "N_importance", _get(cfg, "cascade_samples", 0))
        # share_coarse_fine=True: ONE NeRF evaluated at both levels (must be asked for; never a silent fallback)
        self.share_coarse_fine = bool(_get(cfg, "share_coarse_fine", False))
        self.nerf_0 = NeRF(**kw)
        self.nerf_1 = NeRF(**kw) if (self.N_importance > 0 and not self.share_coarse_fine) else None
        self._packed = {}

    def nerf(self, level):
        if level == 0:
            return self.nerf_0
        if self.nerf_1 is not None:
            return self.nerf_1
        if self.share_coarse_fine:
            return self.nerf_0
        raise RuntimeError("Network has no fine NeRF (built with N_importance / cascade_samples = 0): a fine pass would "
                           "silently evaluate and train the coarse weights.  Build the network from the same cfg as the "
                           "renderer, or set cfg.share_coarse_fine = True to share one NeRF on purpose.")

    def invalidate_packed(self):
        """Drop the packed MFMA images so that the next render re-packs from the parameters.  Call after anything that
        changes parameter VALUES without bumping tensor versions: `p.data.copy_()` / EMA / legacy loaders, and HIP-graph
        replays of a captured optimiser step (the replay updates the parameters and the captured images, but an image
        packed eagerly for another (level, precision, direction) key is stale afterwards)."""
        for k, hit in list(self._packed.items()):
            self._packed[k] = (None,) + tuple(hit[1:])        # keep the buffers (graph-captured pointers stay valid)

    def train(self, mode=True):
        """Every train <-> eval switch drops the packed images (the buffers are kept: one pack kernel per image on the next
        render).  An eval-mode network serves them from the cache keyed on tensor VERSIONS, and versions do not see what a
        training phase may have done to the values: HIP-graph replays of the optimiser step, `.data` writes, EMA."""
        if bool(mode) != self.training:
            self.invalidate_packed()
        return super().train(mode)

    def _version(self, level):
        return tuple(p._version for p in self.nerf(level).parameters())

    def _pack(self, level, device, precision, backward, fused=False):
        """(desc, packed image on `device`) of level's NeRF, rebuilt when any parameter changed.  When the
        parameters live on that GPU the image is packed there (pnr_mlp_pack_device, buffers reused); otherwise
        on the host and uploaded."""
        net = self.nerf(level)                   # raises when a fine level is asked of a coarse-only network
        desc = net.desc(precision)
        if fused and not backward:
            # image for pnr_mlp_forward_composite only: the best fused-inference chunk order the geometry has (fused = 1 / 2: at most
            # that plan -- tests compare the kernels)
            if fused == "softmax":      # the best plan that has a softmax-compositing kernel (ops.fused_image(1))
                desc.plan = ops.fused_plan(ops.desc_for_mode(desc, 1), None)
            else:
                desc.plan = ops.fused_plan(desc, None if fused is True else int(fused))
        key = ("bwd" if backward else "fwd", level if self.nerf_1 is not None else 0, str(device), precision, int(desc.plan))
        ver = self._version(level)
        hit = self._packed.get(key)
        sd = dict(net.named_parameters())
        on_dev = torch.device(device).type == "cuda" and all(p.device == torch.device(device) for p in sd.values())
        # cached until a parameter version changes (or invalidate_packed()).  Exception: a TRAINING network whose
        # parameters live on the GPU is always re-packed -- the device packer is one small kernel, and tensor versions
        # miss .data writes and HIP-graph replays of the optimiser step
        if hit is not None and hit[0] == ver and not (self.training and on_dev):
            return hit[1], hit[2]
        ptrs = tuple(p.data_ptr() for p in sd.values())
        if on_dev:
            # same parameter storage as last time (an in-place optimiser step): the descriptors already on the device
            # are still right, only the packing kernel has to run again -- no host copies, graph-capture safe
            same = hit is not None and hit[3] is not None and len(hit) > 4 and hit[4] == ptrs
            img, ws = ops.pack_mlp_device(desc, sd, backward, hit[2] if hit else None, hit[3] if hit else None, repack=same)
        else:
            sd = {k: v.detach().float().cpu() for k, v in sd.items()}
            img, ws = (ops.pack_mlp_bwd if backward else ops.pack_mlp)(desc, sd).to(device), None
        self._packed[key] = (ver, desc, img, ws, ptrs)
        return desc, img

    def packed(self, level, device, precision=None, fused=False):
        """(desc, packed image).  fused=True: the image only ops.mlp_forward_composite consumes (desc.plan as
        pnr_mlp_fused_plan says; fused = 1 | 2 caps the plan, "softmax" asks for the best plan with a softmax kernel); every other op
        takes the classic image (fused=False)."""
        return self._pack(level, device, precision or self.precision, False, fused)

    def packed_bwd(self, level, device):
        return self._pack(level, device, "bf16", True)

    # --- checkpoint interop: reference-trained weights -> this module
    DEFAULT_KEY_MAP = (
        (r"^(module\.|net\.|network\.)+", ""),                       # DataParallel / wrapper prefixes
        (r"^(coarse|nerf_coarse|model_coarse|cascade\.0)\.", "nerf_0."),
        (r"^(fine|nerf_fine|model_fine|cascade\.1)\.", "nerf_1."),
        (r"\.sigma_linear\.", ".alpha_linear."),
        (r"\.(semantic_linear|semantic_head)\.(\d+)\.", r".semantic_linears.\2."),
        (r"\.(instance_linear|instance_head)\.(\d+)\.", r".instance_linears.\2."),
    )

    def load_reference_state_dict(self, sd, key_map=None, strict=True, skip_concat="input_first", views_concat="feature_first"):
        """Load a reference checkpoint's network state_dict.  skip_concat / views_concat: the order in which the REFERENCE
        concatenates the inputs of the skip layer and of the view layer (`cat(gamma(x), h)` vs
        `cat(h, gamma(x))`, `cat(feature, gamma(d))` vs `cat(gamma(d), feature)` -- unverifiable here).  This network is
        "input_first" / "feature_first" (canonical NeRF); for "hidden_first" / "dir_first" the columns of those two weights
        are rotated on load, which is the whole difference.  key_map: sequence of (regex, replacement) applied in order
        to every key (default: DEFAULT_KEY_MAP -- guesses at the reference's naming; the real names
        cannot be checked here, pass the right table once they can).  Shapes must match exactly: a (out,in) weight of
        the wrong size is an error, never a silent reshape.  Returns {'loaded': [...], 'missing': [...],
        'unexpected': [...]}; with strict=True anything missing or unexpected raises."""
        import re
        own = self.state_dict()
        mapped, unexpected = {}, []
        for k, v in sd.items():
            nk = k
            for pat, rep in (key_map if key_map is not None else self.DEFAULT_KEY_MAP):
                nk = re.sub(pat, rep, nk)
            if nk in own:
                if tuple(v.shape) != tuple(own[nk].shape):
                    raise ValueError(f"{k} -> {nk}: shape {tuple(v.shape)} != {tuple(own[nk].shape)}")
                mapped[nk] = self._concat_order(nk, v, skip_concat, views_concat)
            else:
                unexpected.append(k)
        missing = [k for k in own if k not in mapped]
        if strict and (missing or unexpected):
            raise KeyError(f"load_reference_state_dict: missing {missing[:6]}{'...' if len(missing) > 6 else ''}, "
                           f"unexpected {unexpected[:6]}{'...' if len(unexpected) > 6 else ''}")
        self.load_state_dict(mapped, strict=False)
        self.invalidate_packed()             # the packed images are rebuilt from the new parameters
        return {"loaded": sorted(mapped), "missing": missing, "unexpected": unexpected}

    def _concat_order(self, name, w, skip_concat, views_concat):
        """Columns of a reference weight in this network's order (see load_reference_state_dict)."""
        if skip_concat not in ("input_first", "hidden_first") or views_concat not in ("feature_first", "dir_first"):
            raise ValueError("skip_concat: 'input_first' | 'hidden_first'; views_concat: 'feature_first' | 'dir_first'")
        m = name.split(".")
        if len(m) < 3 or m[-1] != "weight":
            return w
        nerf = getattr(self, m[0], None)
        if nerf is None:
            return w
        if skip_concat == "hidden_first" and m[1] == "pts_linears" and nerf.skip >= 0 and int(m[2]) == nerf.skip + 1:
            return torch.cat([w[:, nerf.W:], w[:, :nerf.W]], 1)          # [h | gamma(x)] -> [gamma(x) | h]
        if views_concat == "dir_first" and m[1] == "views_linears":
            ed = w.shape[1] - nerf.W
            return torch.cat([w[:, ed:], w[:, :ed]], 1)                  # [gamma(d) | feature] -> [feature | gamma(d)]
        return w

    def forward(self, *a, **k):
        raise RuntimeError("Network is evaluated by Renderer.render() through the fused HIP kernel; "
                           "there is no torch forward (and no CPU fallback).")


def make_network(cfg):
    """Reference plugin surface: make_network(cfg) -> nn.Module."""
    return Network(cfg)
