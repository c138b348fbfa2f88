#!/usr/bin/env python3
"""Board power and shader clock WHILE the headline loop runs (hwmon / sysfs of the amdgpu device, sampled from a thread every ~5 ms): is the part at its power cap,
and what does a change of the kernels do to the clock?  Companion of the same-box A/B runs: ARCFLOW_HIP_LIB selects the build.

    python tools/power_probe.py [--steps 20] [--model flux]
prints one JSON line: images/s, power (mean / p5 / p95, W), the cap, sclk (mean / p5 / p95, MHz)."""
import argparse
import glob
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _find():
    """hwmon files of the card HIP device 0 is (matched by PCI address: the box may show other tenants' cards too)."""
    want = None
    try:
        pr = torch.cuda.get_device_properties(0)
        want = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
    except Exception:      # noqa: BLE001
        pass
    cands = []
    for card in sorted(glob.glob('/sys/class/drm/card*/device')):
        slot = None
        try:
            for line in open(os.path.join(card, 'uevent')):
                if line.startswith('PCI_SLOT_NAME='):
                    slot = line.strip().split('=', 1)[1].lower()
        except Exception:  # noqa: BLE001
            pass
        for hw in sorted(glob.glob(os.path.join(card, 'hwmon/hwmon*'))):
            p = [os.path.join(hw, f) for f in ('power1_average', 'power1_input') if os.path.exists(os.path.join(hw, f))]
            if p:
                cap, fq = os.path.join(hw, 'power1_cap'), os.path.join(hw, 'freq1_input')
                cands.append((slot, p[0], cap if os.path.exists(cap) else None, fq if os.path.exists(fq) else None))
    for c in cands:
        if want is not None and c[0] == want.lower():
            return c[1], c[2], c[3], c[0]
    if len(cands) == 1:
        return cands[0][1], cands[0][2], cands[0][3], cands[0][0]
    return None, None, None, f'no card matches {want} among {[c[0] for c in cands]}'


def _read(path, scale):
    try:
        with open(path) as f:
            return float(f.read().strip()) * scale
    except Exception:      # noqa: BLE001
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--model', default='flux')
    args = ap.parse_args()
    import bench
    eng, inp = bench.build_flux_engine(args.model)
    from arcflow_amd import ops
    sig = [1.0, 0.7619048461914063, 0.0]                      # the 2-NFE sigmas of bench.py (shift 3.2)
    x, t, ctx, pooled, guidance, hp, wp = inp
    tvec = [torch.full((1,), s_, device='cuda') for s_ in sig[:2]]
    lat = x.float()

    def image():
        z = lat
        for i in range(2):
            out = eng(z.bfloat16(), tvec[i], ctx, pooled, guidance, hp, wp)
            z = ops.arcflow_step(z, out.means, out.logweights, out.loggammas, sig[i], sig[i], sig[i + 1])
        return z
    for _ in range(3):
        image()
    torch.cuda.synchronize()
    ppath, cpath, fpath, slot = _find()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((_read(ppath, 1e-6) if ppath else None, _read(fpath, 1e-6) if fpath else None))
            time.sleep(0.005)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        image()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()

    def stats(v):
        v = sorted(a for a in v if a is not None)
        if not v:
            return None
        return {'mean': round(sum(v) / len(v), 1), 'p5': round(v[len(v) // 20], 1), 'p95': round(v[(len(v) * 19) // 20], 1), 'n': len(v)}
    print(json.dumps({'lib': os.environ.get('ARCFLOW_HIP_LIB', 'product'), 'images_per_s': round(args.steps / dt, 3), 'power_w': stats([s[0] for s in samples]),
                      'power_cap_w': _read(cpath, 1e-6) if cpath else None, 'sclk_mhz': stats([s[1] for s in samples]), 'source': ppath, 'pci': slot}))


if __name__ == '__main__':
    main()
