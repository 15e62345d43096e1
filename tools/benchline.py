#!/usr/bin/env python3
"""Condense a bench.py JSON line: value, per-phase ms, roofline GB/s, avg attention us."""
import json
import sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline", {})
    print(f"{d['value']:.2f} img/s  step {d['ms_per_step']:.0f} ms  phases {d['last_step_ms']}  "
          f"attn {r.get('achieved', 0):.0f} GB/s {r.get('avg_launch_us', 0):.1f} us share {r.get('decode_loop_share', 0):.2f}")
