#!/usr/bin/env python3
"""Registers / spills / occupancy per kernel from a `hipcc -Rpass-analysis=kernel-resource-usage` log.  Usage: res_summary.py log [name substring ...]"""
import re, sys
t = open(sys.argv[1]).read()
for b in t.split('Function Name: ')[1:]:
    name = b.split('\n')[0].split(' ')[0]
    if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]): continue
    g = lambda k: (re.search(k + r': (\d+)', b) or [None, '?'])[1]
    scr, occ = g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')
    print(f"{name[:100]:100s} V {g('VGPRs')} A {g('AGPRs')} spillV {g('VGPRs Spill')} spillS {g('SGPRs Spill')} scratch {scr} occ {occ}")
