#!/usr/bin/env python3
"""One line per bench.py invocation: ms/step and the per-kernel event times (diagnostics).  usage: bench_kernels.py <bench args...>"""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-f32x3", "--no-graph"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
if not line:
    print("bench failed:", r.stderr[-800:]); sys.exit(1)
d = json.loads(line[-1])
ks = d.get("kernels", {})
print(" ".join(sys.argv[1:]) or "(default)", f"| {d['ms_per_step']:.4f} ms |", "  ".join(f"{k} {v['avg_us']:.1f}" for k, v in ks.items()), f"| sum {sum(v['avg_us'] for v in ks.values()):.0f} us")
