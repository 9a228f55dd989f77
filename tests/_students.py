"""Students of the benched-geometry training-fidelity experiment.

One synthetic scene (a teacher network rendered by the fp32 oracle), one initialisation, one list of ray batches; students
that differ ONLY in the arithmetic of their training step:

  "fp32"      torch autograd through oracle/torch_oracle.py, fp32 everywhere (the reference's arithmetic)
  "bf16_fwd"  bf16-rounded Linear inputs / weights in the forward, fp32 backward through the rounded graph
  "bf16_bwd"  the HIP training path's arithmetic restated on the CPU: bf16 forward AND every dY rounded to bf16 before the
              data-gradient / weight-gradient products (torch_oracle._LinearBf16)
  "bf16_hilo" the same with every dY kept as a bf16 hi + lo pair (16 mantissa bits: what a higher-precision dY would buy)
  "hip"       Renderer.render under autograd through NetworkWrapper on the MI355X (needs the GPU), bf16 or the fp32 parity mode

plus, for the spread an fp32 student has by itself, fp32 students on other batch ORDERS (order_seed) and on an initialisation
perturbed in its last bits (init_jitter).  Test infrastructure: imported by tests/test_gpu_convergence.py and
tools/train_fidelity.py only.
"""
from types import SimpleNamespace as NS

import numpy as np
import torch

from oracle import torch_oracle as to

UNIT = {"rgb": 1.0, "depth": 0.1, "semantic": 1.0, "fix_semantic": 1.0, "instance": 1.0, "fix_instance": 1.0}
# loss weights that keep the image term in charge (the round-3 test's)
IMAGE = {"rgb": 20.0, "depth": 0.2, "semantic": 0.1, "fix_semantic": 0.1, "instance": 0.1, "fix_instance": 0.1}
WEIGHTS = {"unit": (UNIT, 0.1), "image": (IMAGE, 0.02)}


def psnr(a, b):
    return -10.0 * torch.log10(torch.mean((a - b) ** 2)).item()


def scene(Cc=45, Kk=32, Nc=64, Nf=128, n_pool=1536, n_held=384, steps=150, batch=192, threads=None):
    """The benched geometry: 8x256 NeRFs with skip, semantic Cc + instance Kk heads, Nc + Nf samples, the 3D bbox prior."""
    from panopticnerf_amd import synthetic
    if threads:
        torch.set_num_threads(threads)
    oc = to.mlp_config(n_sem=Cc, n_inst=Kk)
    teacher = {"coarse": to.init_params(oc, 51, sigma_bias=0.05), "fine": to.init_params(oc, 52, sigma_bias=0.05)}
    for p in teacher.values():
        p["rgb_linear.weight"] *= 6.0
        p["semantic_linears.1.weight"] *= 4.0
        p["instance_linears.1.weight"] *= 4.0
    frame = synthetic.camera_rays()
    g = torch.Generator().manual_seed(11)
    pool = frame[torch.randint(0, frame.shape[0], (n_pool,), generator=g)].contiguous()
    held = frame[torch.randint(0, frame.shape[0], (n_held,), generator=g)].contiguous()
    box, ids = synthetic.random_boxes(48, Cc, Kk, seed=5)
    with torch.no_grad():
        t_pool = to.render_rays(teacher, oc, pool, Nc, Nf, box=box, box_ids=ids)
        t_held = to.render_rays(teacher, oc, held, Nc, Nf, box=box, box_ids=ids)
    tgt = {"rgb": t_pool["rgb_1"], "depth": t_pool["depth_1"], "semantic": t_pool["semantic_1"].argmax(-1).int(),
           "instance": t_pool["instance_1"].argmax(-1).int()}
    init = {"coarse": to.init_params(oc, 61, sigma_bias=0.03), "fine": to.init_params(oc, 62, sigma_bias=0.03)}
    batches = [torch.randint(0, pool.shape[0], (batch,), generator=g) for _ in range(steps)]
    return NS(oc=oc, Cc=Cc, Kk=Kk, Nc=Nc, Nf=Nf, pool=pool, held=held, box=box, ids=ids, tgt=tgt, t_held=t_held, init=init,
              batches=batches, steps=steps, batch=batch)


def reorder(sc, order_seed):
    """Another batch order over the same pool (the fp32 student's seed-to-seed spread)."""
    g = torch.Generator().manual_seed(1000 + order_seed)
    return [torch.randint(0, sc.pool.shape[0], (sc.batch,), generator=g) for _ in range(sc.steps)]


def oracle_student(sc, mode="fp32", W=UNIT, w3d=0.1, lr=5e-4, batches=None, init_jitter=0.0, log=None, snap=()):
    """Train one CPU student; returns dict(losses, rgb (fine-level colour term per step), psnr, eval maps, params, snaps =
    {k: parameters after k Adam steps, for k in `snap`})."""
    emu = {"fp32": False, "bf16_fwd": True, "bf16_bwd": "bwd", "bf16_hilo": "bwd_hilo"}[mode]
    batches = sc.batches if batches is None else batches
    prm = {lv: {k: v.clone() for k, v in sc.init[lv].items()} for lv in ("coarse", "fine")}
    if init_jitter:
        g = torch.Generator().manual_seed(77)
        for d in prm.values():
            for v in d.values():
                v.mul_(1.0 + init_jitter * (2 * torch.rand(v.shape, generator=g) - 1))
    for d in prm.values():
        for v in d.values():
            v.requires_grad_(True)
    opt = torch.optim.Adam([p for d in prm.values() for p in d.values()], lr=lr)
    losses, rgb, snaps = [], [], {}
    for it, idx in enumerate(batches):
        out = to.render_rays(prm, sc.oc, sc.pool[idx], sc.Nc, sc.Nf, box=sc.box, box_ids=sc.ids, keep_raw=True, emulate_bf16=emu)
        hits = to.bbox_hits(sc.pool[idx], sc.box, 8)
        loss = 0
        for lv in (0, 1):
            maps = {k: out[f"{k}_{lv}"] for k in ("rgb", "depth", "semantic", "fix_semantic", "instance", "fix_instance")}
            terms, total = to.losses(maps, {k: v[idx] for k, v in sc.tgt.items()}, W, sc.Cc, sc.Kk)
            if lv == 1:
                rgb.append(float(terms["rgb"].detach()))
            ls, li = to.sample_labels(out[f"z_vals_{lv}"].detach(), hits[0], hits[1], hits[2], sc.ids)
            raw = out[f"raw_{lv}"].reshape(-1, 4 + sc.Cc + sc.Kk)
            ce_s, _ = to.ce3d(raw[:, 4:4 + sc.Cc], ls.reshape(-1))
            ce_i, _ = to.ce3d(raw[:, 4 + sc.Cc:], li.reshape(-1))
            loss = loss + total + w3d * ce_s + w3d * ce_i
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if it + 1 in snap:
            snaps[it + 1] = {f"{lv}.{k}": v.detach().clone() for lv, d in prm.items() for k, v in d.items()}
        if log and (it % 25 == 0 or it == len(batches) - 1):
            log(f"[{mode}] step {it}: loss {losses[-1]:.4f} rgb {rgb[-1]:.5f}")
    sd = {lv: {k: v.detach() for k, v in d.items()} for lv, d in prm.items()}
    with torch.no_grad():
        ev = to.render_rays(sd, sc.oc, sc.held, sc.Nc, sc.Nf, box=sc.box, box_ids=sc.ids)
    return {"mode": mode, "losses": losses, "rgb": rgb, "psnr": psnr(ev["rgb_1"], sc.t_held["rgb_1"]), "eval": ev, "params": sd,
            "snaps": snaps}


def hip_student(sc, dev, W=UNIT, w3d=0.1, lr=5e-4, precision="bf16", batches=None, snap=(), evaluate=True, graphed=False, fused_adam=False):
    """The HIP student through the trainer's wrapper (NetworkWrapper), same init / batches / terms.  precision "bf16": the
    training path; "fp32": its parity mode (pnr_mlp_forward_train_fp32 / pnr_mlp_backward_fp32).  snap: step counts after which
    the parameters are copied out (`snaps`); graphed: the steps run through train.GraphedStep (one HIP graph per step; implies
    fused_adam); fused_adam: torch's fused capturable Adam (one kernel; its arithmetic differs from the default Adam's in the last bits)."""
    from panopticnerf_amd import NetworkWrapper, make_network, make_renderer
    batches = sc.batches if batches is None else batches
    cfg = NS(N_samples=sc.Nc, N_importance=sc.Nf, num_classes=sc.Cc, num_instances=sc.Kk, precision=precision, chunk_size=4096,
             w_rgb=W["rgb"], w_depth=W["depth"], w_sem=W["semantic"], w_fix_sem=W["fix_semantic"], w_inst=W["instance"],
             w_fix_inst=W["fix_instance"], w_sem3d=w3d, w_inst3d=w3d)
    net = make_network(cfg)
    net.nerf_0.load_state_dict(sc.init["coarse"])
    net.nerf_1.load_state_dict(sc.init["fine"])
    net = net.to(dev).train()
    wrap = NetworkWrapper(net, cfg)
    opt = torch.optim.Adam(net.parameters(), lr=lr, **({"capturable": True, "fused": True} if (graphed or fused_adam) else {}))
    bx, bi = sc.box.to(dev), sc.ids.to(dev)
    losses, rgb, snaps, step = [], [], {}, None

    def mk(idx):
        return {"rays": sc.pool[idx][None].to(dev), "bbox": bx, "bbox_ids": bi, "rgb": sc.tgt["rgb"][idx][None].to(dev),
                "depth": sc.tgt["depth"][idx][None].to(dev), "pseudo_label": sc.tgt["semantic"][idx][None].to(dev),
                "instance_label": sc.tgt["instance"][idx][None].to(dev)}
    if graphed:
        from panopticnerf_amd import train as pnr_train
        step = pnr_train.GraphedStep(wrap, opt, mk(batches[0]))
    for it, idx in enumerate(batches):
        if step is not None:
            _, loss, st = step(mk(idx))
        else:
            _, loss, st, _ = wrap(mk(idx))
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        losses.append(loss.item())
        rgb.append(float(st["rgb_loss_1"]))
        if it + 1 in snap:
            snaps[it + 1] = {f"{lv}.{k}": v.detach().cpu().clone() for lv, n in (("coarse", net.nerf_0), ("fine", net.nerf_1))
                             for k, v in n.state_dict().items()}
    sd = {"coarse": {k: v.detach().cpu() for k, v in net.nerf_0.state_dict().items()},
          "fine": {k: v.detach().cpu() for k, v in net.nerf_1.state_dict().items()}}
    if not evaluate:
        return {"mode": "hip:" + precision, "losses": losses, "rgb": rgb, "params": sd, "snaps": snaps}
    with torch.no_grad():
        ev = make_renderer(cfg, net.eval()).render({"rays": sc.held[None].to(dev), "bbox": bx, "bbox_ids": bi})
    ev = {k: v[0].cpu() for k, v in ev.items()}
    return {"mode": "hip:" + precision, "losses": losses, "rgb": rgb, "psnr": psnr(ev["rgb_1"], sc.t_held["rgb_1"]),
            "eval": ev, "params": sd, "snaps": snaps}


def _loss_of(sc, out, hits, idx, W, w3d):
    loss = 0
    for lv in (0, 1):
        maps = {k: out[f"{k}_{lv}"] for k in ("rgb", "depth", "semantic", "fix_semantic", "instance", "fix_instance")}
        _, total = to.losses(maps, {k: v[idx] for k, v in sc.tgt.items()}, W, sc.Cc, sc.Kk)
        ls, li = to.sample_labels(out[f"z_vals_{lv}"].detach(), hits[0], hits[1], hits[2], sc.ids)
        raw = out[f"raw_{lv}"].reshape(-1, 4 + sc.Cc + sc.Kk)
        ce_s, _ = to.ce3d(raw[:, 4:4 + sc.Cc], ls.reshape(-1))
        ce_i, _ = to.ce3d(raw[:, 4 + sc.Cc:], li.reshape(-1))
        loss = loss + total + w3d * ce_s + w3d * ce_i
    return loss


def oracle_gradient(sc, idx, mode, W=UNIT, w3d=0.1):
    """Gradient of ONE step's loss at the initial weights, {"coarse.<name>" | "fine.<name>": tensor}, in `mode`'s arithmetic."""
    emu = {"fp32": False, "bf16_fwd": True, "bf16_bwd": "bwd", "bf16_hilo": "bwd_hilo"}[mode]
    prm = {lv: {k: v.clone().requires_grad_(True) for k, v in sc.init[lv].items()} for lv in ("coarse", "fine")}
    out = to.render_rays(prm, sc.oc, sc.pool[idx], sc.Nc, sc.Nf, box=sc.box, box_ids=sc.ids, keep_raw=True, emulate_bf16=emu)
    _loss_of(sc, out, to.bbox_hits(sc.pool[idx], sc.box, 8), idx, W, w3d).backward()
    return {f"{lv}.{k}": v.grad.clone() for lv, d in prm.items() for k, v in d.items()}


def hip_gradient(sc, idx, dev, W=UNIT, w3d=0.1, precision="bf16"):
    """The same gradient through the HIP path (NetworkWrapper on `dev`); precision "fp32" = the parity mode."""
    from panopticnerf_amd import NetworkWrapper, make_network
    cfg = NS(N_samples=sc.Nc, N_importance=sc.Nf, num_classes=sc.Cc, num_instances=sc.Kk, precision=precision, chunk_size=4096,
             w_rgb=W["rgb"], w_depth=W["depth"], w_sem=W["semantic"], w_fix_sem=W["fix_semantic"], w_inst=W["instance"],
             w_fix_inst=W["fix_instance"], w_sem3d=w3d, w_inst3d=w3d)
    net = make_network(cfg)
    net.nerf_0.load_state_dict(sc.init["coarse"])
    net.nerf_1.load_state_dict(sc.init["fine"])
    net = net.to(dev).train()
    b = {"rays": sc.pool[idx][None].to(dev), "bbox": sc.box.to(dev), "bbox_ids": sc.ids.to(dev), "rgb": sc.tgt["rgb"][idx][None].to(dev),
         "depth": sc.tgt["depth"][idx][None].to(dev), "pseudo_label": sc.tgt["semantic"][idx][None].to(dev),
         "instance_label": sc.tgt["instance"][idx][None].to(dev)}
    _, loss, _, _ = NetworkWrapper(net, cfg)(b)
    loss.backward()
    out = {}
    for lv, nerf in (("coarse", net.nerf_0), ("fine", net.nerf_1)):
        for k, p in nerf.named_parameters():
            out[f"{lv}.{k}"] = p.grad.detach().cpu().clone()
    return out


def run_job(job):
    """One CPU student in its own process: job = (name, steps, threads, weights); name = mode[:orderN | :jitter].  Returns the
    summary row plus the held-out semantic argmax map (for agreement checks)."""
    import time
    name, steps, threads, weights = job
    torch.set_num_threads(threads)
    sc = scene(steps=steps)
    W, w3d = WEIGHTS[weights]
    mode, _, var = name.partition(":")
    kw = {}
    if var.startswith("order"):
        kw["batches"] = reorder(sc, int(var[5:]))
    elif var == "jitter":
        kw["init_jitter"] = 1e-6
    t0 = time.time()
    r = oracle_student(sc, mode, W, w3d, **kw)
    row = summary(r)
    row.update(name=name, weights=weights, seconds=round(time.time() - t0, 1), sem_argmax=r["eval"]["semantic_1"].argmax(-1).tolist(),
               rgb_curve=[round(float(np.mean(r["rgb"][i:i + 10])), 6) for i in range(0, len(r["rgb"]), 10)])
    return row


def summary(r):
    return {"mode": r["mode"], "psnr": round(r["psnr"], 4), "loss_last5": float(np.mean(r["losses"][-5:])),
            "rgb_last10": float(np.mean(r["rgb"][-10:])), "rgb_first10": float(np.mean(r["rgb"][:10]))}
