"""Package power and clocks of the FUSED fine-level MLP launch for one or more libpnr builds (is the `nt` policy's
extra fabric traffic what costs clock?).  Per build, in its own process: ~3 s of back-to-back launches while a thread samples the
GPU's hwmon power / sclk / mclk / fclk at ~50 Hz (sysfs; `rocm-smi --json` every 0.25 s as a fallback); prints the launch time,
the in-kernel shader clock (s_memtime / s_memrealtime of workgroup 0) and the mean / max of the samples taken while the GPU was busy.
usage: python tools/power_ab.py default <name under build/ab> ..."""
import glob, json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from types import SimpleNamespace as NS
    from panopticnerf_amd import benchlib, make_network, ops, synthetic
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = make_network(NS(N_importance=128, num_classes=45, num_instances=32)).eval()
    synthetic.trained_like_(net)
    net = net.to(dev)
    rays = synthetic.camera_rays()[:65536].to(dev)
    z = ops.stratified(rays, 192)
    desc, img = net.packed(1, dev, fused=True)
    benchlib.time_mlp_forward_tiles(desc, img, rays, z, 3)

    hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    samples, stop = [], threading.Event()

    def read(path):
        try:
            return float(open(path).read().split()[0])
        except (OSError, ValueError, IndexError):
            return None

    def cur_clock(path):          # pp_dpm_* lists the levels, the current one is starred
        try:
            for l in open(path):
                if "*" in l:
                    return float(l.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError):
            pass
        return None

    def sampler():
        while not stop.is_set():
            row = {}
            for h in hw[:1]:
                for key, f in (("power_w", "power1_average"), ("power_in_w", "power1_input")):
                    v = read(os.path.join(h, f))
                    if v is not None:
                        row[key] = v / 1e6
                dev_dir = os.path.dirname(os.path.dirname(h))
                for key, f in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk"), ("fclk", "pp_dpm_fclk")):
                    v = cur_clock(os.path.join(dev_dir, f))
                    if v is not None:
                        row[key] = v
            if not row:
                try:
                    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
                    row = {"rocm_smi": json.loads(out)}
                except Exception:      # noqa: BLE001
                    row = {}
                time.sleep(0.25)
            if row:
                samples.append(row)
            time.sleep(0.02)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    res = [benchlib.time_mlp_forward_tiles(desc, img, rays, z, 40) for _ in range(7)]      # ~3 s busy
    stop.set()
    th.join(timeout=2)
    ms, mhz = min(res)
    agg = {}
    for k in ("power_w", "power_in_w", "sclk", "mclk", "fclk"):
        v = [s[k] for s in samples if k in s]
        if v:
            agg[k] = {"mean": round(sum(v) / len(v), 1), "max": round(max(v), 1), "n": len(v)}
    smi = [s["rocm_smi"] for s in samples if "rocm_smi" in s]
    print("RESULT " + json.dumps({"lib": sys.argv[2], "ms": round(ms, 4), "kernel_mhz": round(mhz), "all_ms": [round(r[0], 3) for r in res],
                                  "all_mhz": [round(r[1]) for r in res], "samples": agg, "rocm_smi_last": smi[-1] if smi else None}))
    sys.exit(0)

for name in sys.argv[1:]:
    env = dict(os.environ)
    if name != "default":
        env["PNR_LIB_PATH"] = os.path.join(ROOT, "build", "ab", "libpnr_%s.so" % name)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name], capture_output=True, text=True, env=env, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    print(line[0][7:] if line else "%s FAILED: %s" % (name, out.stderr[-500:]), flush=True)
