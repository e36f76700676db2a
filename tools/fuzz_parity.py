"""Randomized parity sweep (GPU): random geometry / batch / knob count / precision through tests.gpu_checks.run_fused
for a fixed wall-clock budget.  Not part of the pytest suite (its coverage is fixed cases); run ad hoc:
    gpurun -- 'timeout 600 python tools/fuzz_parity.py'
"soft" lines (round 5): checks that missed their fixed tolerance but sit within 3 x the measured spread of that quantity for that configuration
(tests/gpu_spread.py; until round 4 such lines were filtered by tensor name -- "conv_analysis" -- without evidence).  In bf16 mode with B = 1 a single operand landing on the other side of a bf16 rounding
boundary is 0.4 % of one of only OT = 9 summands, so an occasional 2-4e-3 max-relative outlier there is rounding, not a bug.
Round-1 result: 260 configurations in 150 s, 0 hard failures in fp32, 1 such bf16 B=1 outlier.
With the bf16 levels drawn at random (1 = STFT GEMMs, 2 = also the autoencoder layers; fused tolerances 3e-3 / 2e-2 = the
noise floors of tools/bf16_noise_floor.py): 254 configurations, fp32 all green, 2 level-2 outliers of 4-5e-2 at B <= 3 --
a fused bf16 step is chaotic at that level (one flipped rounding per few thousand values, nine layers of amplification).
Round 2 (f32x3 = three-plane bfloat16 split against the fp32 oracle at fp32 tolerances, and f16_all added to the draw; 400 s):
628 configurations, fp32 and f32x3 all green (the same "soft" analysis-gradient conditioning lines in both), 4 level-2 outliers
(bf16_all 4-5e-2 at B <= 9, f16_all 1.0-1.3e-2 at B <= 3) of the kind described above.
Round 3 (400 s, profiles/r03_fuzz_parity.txt): 628 configurations, fp32 and f32x3 all green, 4 level-2 outliers of the same kind (bf16_all 4-5e-2 at B <= 9, f16_all 1.3e-2 at B = 2); after the wide-path pass, 500 s: 806 configurations, fp32 / f32x3 green, 6 such outliers (bf16_all 4-7e-2 at B <= 9, f16_all 1.0-1.3e-2 at B <= 3).
Round 4 (500 s, profiles/r04_fuzz_parity.txt): 824 configurations incl. odd batches on the wide path again (the library reports its effective arithmetic, the oracle follows); fp32 / f32x3 /
bf16 / bf16_all green; ONE hard line -- f16_all, L = 65536, B = 2, K = 16, seed 482: 1.3e-2 on the layer-1 weight gradient of the phase net -- the configuration tools/fuzz_ground.py shows to sit at
0.7 x the oracle's own spread (1.8e-2); the sweep held scale 8 to the scale-1 tolerance (8e-3) where the suite uses 2 x that: same multiplier here now.
Re-run on the final round-4 sources (2ebbf655b2257caa; 500 s): 824 configurations, 0 hard failures in any of the six arithmetic modes.
Round 5 (500 s, profiles/r05_fuzz_parity.txt): no tensor is exempt by name any more -- a miss of a fixed fp32 tolerance is graded on the spot against the measured spread of that
quantity for that configuration (tests/gpu_spread.py).  777-794 configurations (odd batches on the wide path now run 16-bit layers), 20 "soft" lines, every one within 1.6 x its spread
(bound: 3 x); ONE hard line -- f16_all, L = 65536, a single window (B = 1, K = 2, seed 300): phase-net encoder gradients 2.1-2.6e-2 against the sweep's 1.6e-2 -- which tools/fuzz_ground.py
shows at 0.4-0.6 x the rounding oracle's own spread for that window (4.1-6.6e-2; its analysis-basis gradients move by 120 % under a 1e-6 perturbation): the 14th case of
tests/test_gpu_parity.py::test_fuzz_outliers_grounded.
Round 6, a FRESH seed (330 s, seed 4242: profiles/r06_fuzz_parity_seed4242.txt; big, 450 s, seed 77: profiles/r06_fuzz_parity_big_seed77.txt): 426 + 66 configurations, 25 soft lines, four hard
lines, all four on the analysis-basis gradients: three f16_all (B = 13; single windows at legacy scale 2 and at L = 65536) -> tools/fuzz_ground.py cases 14-16, and one f32x3 single
window at lean scale 2 whose device error is 0.3 x its spread but 14 x the fixed tolerance, i.e. over the cap of tests/gpu_spread.py -> accepted there now only as a LOCALIZED miss
(<= 16 of the 1024 rows of the tensor over the tolerance), tools/fuzz_ground_f32.py case 21.
The default draw again on the final sources (500 s, seed 1234, profiles/r06_fuzz_parity.txt): 772 configurations, 20 soft lines, 0 hard -- the f16_all single window of round 5 (seed 300) is
graded on the spot now (0.4-0.6 x its spread).
    python tools/fuzz_parity.py [seconds] [big|small|geo] [seed]"""
import sys, time, random; sys.path.insert(0, '.')
from tests import gpu_checks as G
random.seed(int(sys.argv[3]) if len(sys.argv) > 3 else 1234)
GEO = len(sys.argv) > 2 and sys.argv[2] == "geo"
BIG = len(sys.argv) > 2 and sys.argv[2] == "big"      # round 5: batches of 33..160 windows at the 8192-sample window -- where 128-row tiles of the frame-major row order hold one or two
                                                       # frames and the structural-zero skipping of st_gemm_tn.h / st_gemm16.h is active (the default draw stays below 19 windows: one tile holds every frame)
GROUND16 = True
import importlib.util as _ilu, os as _os
_spec = _ilu.spec_from_file_location("fuzz_ground", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "fuzz_ground.py"))
FG = _ilu.module_from_spec(_spec); _spec.loader.exec_module(FG)
t0 = time.time(); nbad = 0; n = 0
while time.time() - t0 < float(sys.argv[1] if len(sys.argv) > 1 else 150):
    scale = random.choice([1, 1, 1, 2, 8]); scheme = "lean" if scale != 2 else random.choice(["lean", "legacy"])
    shrink = random.choice([1, 2, 4, 4, 8]) if scale == 1 else 4
    B = random.choice([1, 2, 3, 4, 5, 6, 9, 13]) if scale == 1 else random.choice([1, 2, 3])
    K = random.choice([1, 2, 3, 4, 4, 5, 8, 12, 16]); seed = random.randrange(1000)
    if BIG:
        scale, scheme, shrink, B = 1, "lean", random.choice([2, 4, 4, 8]), random.choice([33, 48, 64, 96, 100, 128, 130, 160])
    if GEO:
        # round 6: the geometry corners the default draw never visits -- every shrink factor at every window scale (the default ties shrink 4 to scale != 1), scale 4, and
        # K = 0 (a model without knobs); all of them on the wide autoencoder path except (scale 1, shrink >= 2)
        scale = random.choice([1, 2, 4, 4, 8, 8]); scheme = "lean" if scale != 2 else random.choice(["lean", "legacy"])
        shrink = random.choice([1, 2, 4, 8]); B = random.choice([1, 2, 3, 4]); K = random.choice([0, 0, 1, 2, 3, 4, 7, 16])
    bf = random.choice([0, 0, 0, 3, 3, 1, 1, 2, 2, 4])     # 0 = fp32, 1 = bf16 GEMMs, 2 = bf16 GEMMs + autoencoder layers, 3 = f32x3 (fp32 oracle, fp32 tolerances), 4 = f16_all
    # odd batches on the wide path (scale 8): the library runs the autoencoder layers in fp32 there and reports it (st_effective_prec); the
    # checks' oracle follows (gpu_checks.follow_effective_arithmetic), so the sweep draws them again
    kw = dict(B=B, seed=seed, K=K, steps=1, scale=scale, scheme=scheme, shrink=shrink)
    try:
        kw["B"] = B
        if bf == 3:
            with G.split_mode(): res = G.run_fused(**kw)
        elif bf == 4:
            with G.mixed_mode(2, half="f16", tol_scale=G.mixed_mode.FUSED_TOL_F16[2] * (2.0 if scale == 8 else 1.0)): res = G.run_fused(**kw)      # as tests/test_gpu_parity.py: 174-frame rows, more roundings per sum
        elif bf:
            with G.bf16_mode(bf, tol_scale=G.bf16_mode.FUSED_TOL[bf]): res = G.run_fused(**kw)
        else:
            res = G.run_fused(**kw)
        miss = [r for r in res if not r["ok"]]
        # round 5: no tensor is exempt by name.  A miss of the fixed tolerance in the fp32-grade modes is graded on the spot against the spread of that quantity for this
        # configuration (tests/gpu_spread.py: float32-vs-float64 oracle, float64 oracle under 1e-6 perturbations -- CPU work, ~10-60 s, cached in profiles/); in the 16-bit modes the
        # fused tolerances already ARE the measured noise floors of TYPICAL configurations; a miss there is graded against the rounding oracle's own spread below (round 6)
        bad, soft = miss, []
        if miss and bf in (0, 3):
            from tests import gpu_spread as S
            bad = S.grounded(res, kw)
            soft = [r for r in miss if r.get("grounded")]
        elif miss and GROUND16:
            # round 6: the 16-bit lines are graded on the spot too -- the rounding oracle against itself under eight 1e-6 perturbations for THIS configuration
            # (tools/fuzz_ground.py self_noise: CPU, 10-90 s) and the per-op run on oracle-fed inputs: "hard" is what neither explains
            mode = {1: "bf16", 2: "bf16_all", 4: "f16_all"}[bf]
            nz = FG.self_noise(mode, kw, level=1 if bf == 1 else 2)
            for r in miss:
                z = nz.get(r["name"]); r["spread"] = z
                if z and r["rel"] <= 3.0 * z: r["ratio"] = r["rel"] / z; r["grounded"] = True
            half = "bf16" if mode.startswith("bf16") else "f16"
            with G.mixed_mode(1 if bf == 1 else 2, half=half, tol_scale=(None if scale != 8 or bf == 1 else (40.0 if half == "bf16" else 20.0))):
                per = G.run_all(B=kw["B"], seed=kw["seed"], K=kw["K"], scale=kw["scale"], scheme=kw["scheme"], shrink=kw["shrink"])
            per_bad = [dict(r, name="per-op " + r["name"]) for r in per if not r["ok"] and r["rel"] > nz.get(r["name"].replace("ae_bwd.g.", "grad."), 0.0)]
            bad = [r for r in miss if not r.get("grounded")] + per_bad
            soft = [r for r in miss if r.get("grounded")]
    except Exception as e:
        bad = [dict(name="EXC " + str(e)[:160], rel=0)]; soft = []
    n += 1; nbad += bool(bad)
    if bad or soft:
        print(("BAD " if bad else "soft"), kw, ("f32", "bf16", "bf16_all", "f32x3", "f16_all")[bf],
              [(r['name'], f"{r['rel']:.1e}") + ((f"{r['ratio']:.2f} x spread",) if "ratio" in r else ()) + ((f"in {r['rows_over']} of {r['rows']} rows",) if r.get("localized") else ()) for r in (bad + soft)[:4]], flush=True)
print(f"{n} random configurations, {nbad} with hard failures, {time.time()-t0:.0f} s")
