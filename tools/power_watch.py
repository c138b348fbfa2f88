#!/usr/bin/env python3
"""Board power / shader clock of HIP device 0 while a COMMAND runs (hwmon sampled every 10 ms from this process; the command is a child):
    python tools/power_watch.py -- python bench.py --train --steps 3 --warmup 1 --no-cpu-baseline
Prints the command's last stdout line, then one JSON line: power / sclk statistics over the samples above 60 % of the maximum seen (the loaded phase)."""
import json
import subprocess
import sys
import threading
import time

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import power_probe as pp  # noqa: E402


def main():
    cmd = sys.argv[sys.argv.index('--') + 1:]
    ppath, cpath, fpath, slot = pp._find()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.time(), pp._read(ppath, 1e-6) if ppath else None, pp._read(fpath, 1e-6) if fpath else None))
            time.sleep(0.01)
    th = threading.Thread(target=sampler)
    th.start()
    r = subprocess.run(cmd, capture_output=True, text=True)
    stop.set()
    th.join()
    last = [l for l in r.stdout.splitlines() if l.strip()]
    print(last[-1][:300] if last else r.stderr[-300:])
    pw = [s[1] for s in samples if s[1] is not None]
    if not pw:
        print(json.dumps({'error': 'no power samples', 'pci': slot}))
        return
    thr = 0.6 * max(pw)
    hot = [s for s in samples if s[1] is not None and s[1] >= thr]

    def stats(v):
        v = sorted(v)
        return {'mean': round(sum(v) / len(v), 1), 'p5': round(v[len(v) // 20], 1), 'p50': round(v[len(v) // 2], 1), 'p95': round(v[(len(v) * 19) // 20], 1), 'n': len(v)}
    print(json.dumps({'cmd': ' '.join(cmd)[-120:], 'power_w_loaded': stats([s[1] for s in hot]), 'sclk_mhz_loaded': stats([s[2] for s in hot if s[2] is not None]),
                      'power_cap_w': pp._read(cpath, 1e-6) if cpath else None, 'loaded_s': round(len(hot) * 0.01, 1), 'pci': slot}))


if __name__ == '__main__':
    main()
