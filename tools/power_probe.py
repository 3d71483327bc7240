#!/usr/bin/env python3
"""Samples the GPU's power / clocks from sysfs (hwmon) in a background thread while a workload runs: is the chip power-limited?
    python tools/power_probe.py [--batch 16] [--steps 20]"""
import argparse, glob, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "generic-diffusion-feature_amd")]

def our_bus_id():
    """PCI address of HIP device 0 (the node's other cards run other tenants' jobs)"""
    try:
        import torch
        p = torch.cuda.get_device_properties(0)
        return "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def find_nodes(bus=None):
    out = {}
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        if bus and bus.lower() not in os.path.realpath(os.path.dirname(os.path.dirname(h))).lower():
            continue
        for f in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
            p = os.path.join(h, f)
            if os.path.exists(p):
                out.setdefault(h, {})[f] = p
    return out

def rd(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None

class Sampler(threading.Thread):
    def __init__(self, nodes, dt=0.01):
        super().__init__(daemon=True); self.nodes = nodes; self.dt = dt; self.rows = []; self.stop = False
    def run(self):
        while not self.stop:
            t = time.perf_counter()
            self.rows.append((t, {k: rd(p) for k, p in self.nodes.items()}))
            time.sleep(self.dt)

if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=16); ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--version", default="xl")
    a = ap.parse_args()
    import torch
    nodes = find_nodes(our_bus_id())
    print("hwmon nodes:", len(nodes), our_bus_id(), flush=True)
    os.system("rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40")
    import torch
    from components.native import NativeUNet, ARCH_CONFIGS
    import bench as BB
    dev = torch.device("cuda:0"); cfg = ARCH_CONFIGS[a.version]; lat = 128 if a.version == "xl" else 64
    net = NativeUNet(cfg, device=dev).init_synthetic(seed=0)
    g = torch.Generator(device=dev).manual_seed(1); B = a.batch
    x = torch.randn(B, 4, lat, lat, generator=g, device=dev).half()
    ctx = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g, device=dev).half().expand(B, -1, -1).contiguous()
    t = torch.full((B,), 100.0, device=dev); txt = tid = None
    if cfg["addition_embed_text_time"]:
        pooled = cfg["add_in_dim"] - 6 * cfg["addition_time_embed_dim"]
        txt = torch.randn(1, pooled, generator=g, device=dev).half().expand(B, -1).contiguous()
        tid = torch.tensor([[1024, 1024, 0, 0, 1024, 1024]], dtype=torch.float32, device=dev).repeat(B, 1)
    ids = BB.PRACTICAL[a.version]
    for _ in range(4): net.forward_raw(x, t, ctx, txt, tid, hook_ids=ids, shared_ctx=True)
    torch.cuda.synchronize()
    flat = {os.path.basename(os.path.dirname(os.path.dirname(os.path.dirname(h)))) + ":" + k: p for h, d in nodes.items() for k, p in d.items()
            if k in ("power1_input", "freq1_input")}
    smp = Sampler(flat, dt=0.02); smp.start()
    time.sleep(0.3)
    t0 = time.perf_counter()
    for _ in range(a.steps): net.forward_raw(x, t, ctx, txt, tid, hook_ids=ids, shared_ctx=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    time.sleep(0.3); smp.stop = True; smp.join()
    busy = [r for (ts, r) in smp.rows if t0 + 0.2 <= ts <= t1]
    idle = [r for (ts, r) in smp.rows if ts < t0]
    def avg(rows, k):
        v = [r[k] for r in rows if r.get(k) is not None]
        return (sum(v) / len(v), min(v), max(v), len(v)) if v else None
    print(json.dumps({"img_per_s": round(B * a.steps / (t1 - t0), 2), "ms_per_step": round(1e3 * (t1 - t0) / a.steps, 2),
                      "busy": {k: avg(busy, k) for k in flat}, "idle": {k: avg(idle, k) for k in flat}}, indent=0))
    # time series of the most active card
    best = max((k for k in flat if k.endswith("power1_input")), key=lambda k: (avg(busy, k) or (0,))[0])
    fk = best.replace("power1_input", "freq1_input")
    print("series", best, [(round(ts - t0, 2), r[best] // 1000000, (r[fk] or 0) // 1000000) for (ts, r) in smp.rows[::10]])
    os.system("rocm-smi --showpower --showclocks 2>&1 | head -30")
