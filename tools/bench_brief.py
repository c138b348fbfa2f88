#!/usr/bin/env python3
"""Print a one-line digest of a bench.py JSON line read from stdin."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'):
        continue
    d = json.loads(line)
    r, a = d.get('roofline', {}), d.get('roofline_attention', {})
    print(f"{sys.argv[1] if len(sys.argv) > 1 else ''} img/s={d['value']:.3f} ms={d['ms_per_step']:.1f} "
          f"gemm={r.get('achieved', 0):.0f}TF({r.get('share_of_step_time', 0):.2f}) "
          f"attn={a.get('achieved', 0):.0f}TF({a.get('share_of_step_time', 0):.2f})")
