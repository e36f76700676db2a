#!/usr/bin/env python3
"""Pretty-print a bench.py JSON line (stdin or file)."""
import json, sys
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
o = json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
print(f"ms/step {o['ms_per_step']:.4f}  windows/s {o['windows_per_s']:.0f}  frames/s {o['value']:.0f}  "
      f"frac_of_fp32_peak {o['step_frac_of_fp32_mfma_peak']:.3f}  n_gpus {o['n_gpus']}  loss {o['loss']:.5f}")
tot = 0.0
for k, v in o.get("kernels", {}).items():
    tot += v["avg_us"]
    print(f"  {k:26s} {v['avg_us']:8.1f} us  {v.get('tflops', 0):6.1f} TF")
print(f"  {'(sum of kernels)':26s} {tot:8.1f} us")
if "roofline" in o:
    print("roofline:", o["roofline"])
if "cpu_baseline" in o:
    print("cpu_baseline:", o["cpu_baseline"])
