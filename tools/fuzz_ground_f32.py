#!/usr/bin/env python3
"""Run the fp32 "soft" lines of the randomized parity sweep to ground.

profiles/r04_fuzz_parity.txt holds 21 lines where the exact-fp32 (10) or f32x3 (11) fused step missed the suite's own tolerance on the
analysis-basis gradients (2e-4 of the tensor maximum) or on the parameters after one Adam step (2e-5 absolute); the sweep and
tests/test_gpu_parity.py filtered them out BY NAME ("conv_analysis").  This tool replaces the name filter with evidence, per configuration:

  f32-vs-f64   the oracle run in float32 arithmetic against the same oracle in float64 (same fp32 inputs): what fp32 arithmetic ITSELF does
               to that tensor -- the reference (PyTorch fp32) sits on this side;
  self-noise   the float64 oracle against itself under NPERT independent 1e-6 relative perturbations of inputs and parameters (max over draws);
  device       tests.gpu_checks.run_fused in that arithmetic mode against the float64 oracle (the flagged number) and against the float32 oracle.

Reading: device <= 3 x max(f32-vs-f64, self-noise)  =>  the miss is the conditioning of atan2's gradient (-im, re) / (re^2 + im^2) at near-silent bins
(nn_proc.py:309-310) -- an error of 1e-7 |x|_max in re / im is a RELATIVE error of 1e-7 |x|_max / mag in d re, d im, summed over frames into the basis gradient --
and the per-configuration tolerance max(2e-4, 3 x spread) that tests.gpu_checks.grounded() applies is the honest one.  device >> spread => a kernel
problem.  The same rule (no name filter) is what tools/fuzz_parity.py and tests/test_gpu_parity.py now use for EVERY tensor.

    python tools/fuzz_ground_f32.py            # CPU part only: spreads -> profiles/r05_fuzz_f32_spread.json
    gpurun -- python tools/fuzz_ground_f32.py gpu     # + the device columns, prints the table (profiles/r05_fuzz_f32_grounding.txt)
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# the 21 soft lines of profiles/r04_fuzz_parity.txt: (mode, kwargs of run_fused)
_L = "lean"
CASES = [
    ("f32x3", dict(B=6, seed=576, K=4, scale=1, scheme=_L, shrink=8)),
    ("f32x3", dict(B=2, seed=561, K=8, scale=8, scheme=_L, shrink=4)),
    ("f32", dict(B=13, seed=310, K=2, scale=1, scheme=_L, shrink=8)),
    ("f32", dict(B=13, seed=361, K=1, scale=1, scheme=_L, shrink=8)),
    ("f32x3", dict(B=5, seed=509, K=3, scale=1, scheme=_L, shrink=8)),
    ("f32x3", dict(B=2, seed=178, K=5, scale=2, scheme="legacy", shrink=4)),
    ("f32x3", dict(B=3, seed=445, K=3, scale=1, scheme=_L, shrink=2)),
    ("f32x3", dict(B=3, seed=104, K=4, scale=8, scheme=_L, shrink=4)),
    ("f32x3", dict(B=9, seed=130, K=5, scale=1, scheme=_L, shrink=4)),
    ("f32", dict(B=4, seed=768, K=12, scale=1, scheme=_L, shrink=4)),
    ("f32", dict(B=1, seed=266, K=4, scale=8, scheme=_L, shrink=4)),
    ("f32x3", dict(B=2, seed=202, K=2, scale=8, scheme=_L, shrink=4)),
    ("f32x3", dict(B=13, seed=93, K=3, scale=1, scheme=_L, shrink=4)),
    ("f32", dict(B=6, seed=195, K=2, scale=1, scheme=_L, shrink=2)),
    ("f32x3", dict(B=3, seed=677, K=3, scale=8, scheme=_L, shrink=4)),
    ("f32", dict(B=2, seed=229, K=16, scale=1, scheme=_L, shrink=4)),
    ("f32x3", dict(B=13, seed=576, K=1, scale=1, scheme=_L, shrink=4)),
    ("f32", dict(B=13, seed=93, K=3, scale=1, scheme=_L, shrink=4)),
    ("f32", dict(B=1, seed=680, K=3, scale=8, scheme=_L, shrink=4)),
    ("f32", dict(B=1, seed=633, K=8, scale=8, scheme=_L, shrink=4)),
    ("f32", dict(B=9, seed=546, K=4, scale=1, scheme=_L, shrink=4)),
]
# round 6, sweep with a fresh seed (profiles/r06_fuzz_parity_seed4242.txt): a SINGLE window at lean scale 2 whose analysis-basis gradient has a spread of ~1e-2 -- the
# device sits at 0.3 x that, but above the 10 x tolerance cap of tests/gpu_spread.py, so the sweep calls it hard
CASES += [
    ("f32x3", dict(B=1, seed=499, K=2, scale=2, scheme=_L, shrink=4)),
]
# ... and the one window two further fresh-seed sweeps (926 configurations: profiles/r06_fuzz_parity_seed9001.txt, _big_seed31337.txt) flagged hard, twice (K = 12 and K = 8): exact fp32,
# the imaginary analysis basis' gradient 1.9e-1 of its maximum off in ONE row (spread 3.2e-1), which moves the published clip norm by 1 % -- the norm had no spread entry until then
CASES += [
    ("f32", dict(B=5, seed=719, K=12, scale=1, scheme=_L, shrink=4)),
]
NPERT = 8
CACHE = os.path.join(ROOT, "profiles", "r05_fuzz_f32_spread.json")
AN = ("grad.dft_analysis.conv_analysis_real.weight", "grad.dft_analysis.conv_analysis_imag.weight", "train0.params", "step.l1norm")


def tag(mode, kw):
    return f"{mode} B={kw['B']} K={kw['K']} scale={kw['scale']} {kw['scheme']} shrink={kw['shrink']} seed={kw['seed']}"


def main():
    gpu = len(sys.argv) > 1 and sys.argv[1] == "gpu"
    from tests import gpu_spread as S
    spread = {}
    for mode, kw in CASES:
        tg = tag("f32", kw)                # the spread is a property of the configuration, not of the device mode (f32x3 is checked against the fp32 oracle too)
        assert tg == S.case_tag(kw)
        spread[tg] = S.spread_of(kw, npert=NPERT, write=True)      # through the committed cache (profiles/r05_fuzz_f32_spread.json)
        s = spread[tg]
        print(f"[spread] {tg}: " + ", ".join(f"{k.replace('grad.dft_analysis.conv_analysis_', 'g.an_').replace('.weight', '')} f32-vs-f64 {s['f32'].get(k, 0):.1e} / self-noise {s['noise'].get(k, 0):.1e}" for k in AN), flush=True)
    if not gpu:
        return
    from tests import gpu_checks as G
    print("\nconfig | tensor: device vs f64 oracle [vs f32 oracle] / spread = max(f32-vs-f64, self-noise) (ratio) | verdict")
    nsus = 0
    for mode, kw in CASES:
        s = spread[tag("f32", kw)]
        if mode == "f32x3":
            with G.split_mode():
                res, res32 = G.run_fused(steps=1, **kw), G.run_fused(steps=1, oracle_dtype="f32", **kw)
        else:
            res, res32 = G.run_fused(steps=1, **kw), G.run_fused(steps=1, oracle_dtype="f32", **kw)
        r32 = {r["name"]: r for r in res32}
        lines, verdict = [], "conditioning"
        for r in res:
            if r["ok"]:
                continue
            key = "train0.params" if r["name"].startswith("train0.params") else r["name"]
            sp = max(s["f32"].get(key, 0.0), s["noise"].get(key, 0.0))
            ratio = r["rel"] / sp if sp > 0 else float("inf")
            lines.append(f"{r['name'].replace('dft_analysis.conv_analysis_', 'an_').replace('.weight', '')} {r['rel']:.1e} [{r32.get(r['name'], r)['rel']:.1e}] / {sp:.1e} ({ratio:.1f}x)")
            if ratio > 3.0:
                verdict = "SUSPECT"
        nsus += verdict != "conditioning"
        print(f"{tag(mode, kw)} | " + ("; ".join(lines) if lines else "nothing flagged on this box") + f" | {verdict}", flush=True)
    print(f"\n{len(CASES)} configurations, {nsus} not explained by the conditioning of the quantity itself")


if __name__ == "__main__":
    main()
