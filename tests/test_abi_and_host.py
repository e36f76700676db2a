"""CPU: the C-ABI library loads and exports every symbol include/signaltrain_hip.h declares (no compute calls),
host-side geometry / layout logic, the Python mirror of the reference surface, and the CPU baseline port."""
import ctypes as C
import os
import re
import numpy as np
import pytest
import torch

from oracle import st_oracle as O
from signaltrain_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "signaltrain_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(st_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert lib.st_version() >= 100


def test_geometry_matches_oracle_and_golden(golden_dir):
    tab = np.load(os.path.join(golden_dir, "g1_geometry.npz"))["table"]
    for s, sh, legacy, L, y, T, OT, N, H in tab:
        d = _lib.geometry(int(s), int(sh), 4, 1, "legacy" if legacy else "lean")
        assert (d.L, d.y, d.T, d.OT, d.N, d.H) == (L, y, T, OT, N, H)


def test_param_layout_and_workspace():
    d = _lib.geometry(1, 4, 4, 256)
    offs, total = _lib.param_offsets(d)
    assert offs[:5] == [0, 1 << 20, 2 << 20, 3 << 20, 4 << 20] and total == 4211096 and all(o % 4 == 0 for o in offs)
    from signaltrain_amd.engine import param_names, param_shapes
    names, shapes = param_names(), param_shapes(d)
    assert names == O.param_order() and sum(int(np.prod(s)) for s in shapes) == 4211090
    lib = _lib.load()
    assert lib.st_kp(513) == 1056
    assert 50e6 < lib.st_workspace_bytes(C.byref(d)) < 2e9
    bad = _lib.st_dims(); bad.B = 1
    assert lib.st_workspace_bytes(C.byref(bad)) == 0 and b"dimension" in lib.st_last_error()


def test_error_reporting_without_gpu():
    lib = _lib.load()
    d = _lib.geometry(1, 4, 4, 2)
    rc = lib.st_analysis_fwd(C.byref(d), None, None, None, 0.5, None, None, None, None, None)
    assert rc == -1 and b"null" in lib.st_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc, "st_analysis_fwd")


def test_model_mirror_state_dict_and_init(golden_dir):
    from signaltrain_amd import nn_proc
    nn_proc._QUIET = True
    m = nn_proc.st_model(scale_factor=1, shrink_factor=4, num_knobs=4)
    sd = m.state_dict()
    assert list(sd.keys()) == O.param_order()
    assert (m.in_chunk_size, m.out_chunk_size, m.num_knobs, m.scale_factor, m.shrink_factor) == (8192, 2048, 4, 1, 4)
    g = np.load(os.path.join(golden_dir, "g2_init_bases.npz"))
    from tests.golden_util import SAMPLE_ROWS, STFT_KEYS
    for k in STFT_KEYS:                       # init bases == the reference's, to <= 1 ulp on the sampled rows
        np.testing.assert_allclose(sd[k].numpy()[SAMPLE_ROWS, 0], g["rows_" + k], rtol=0, atol=4e-9)
    assert m.mpaec.dft_analysis.conv_analysis_real.weight.shape == (1024, 1, 1024)     # io_methods.py:484-493 attribute path
    b = sd["mpaec.aenc.fnn_enc.bias"]; w = sd["mpaec.aenc.fnn_addknobs.weight"]
    assert float(b.abs().max()) == 0.0 and w.shape == (16, 20) and 0.1 < float(w.std()) < 0.4   # xavier_normal: sqrt(2/36)
    m8 = nn_proc.st_model(scale_factor=8, shrink_factor=4, num_knobs=4)
    assert (m8.in_chunk_size, m8.out_chunk_size, m8.mpaec.aenc._T, m8.mpaec.aenc._OT) == (65536, 16256, 174, 46)
    with pytest.raises(RuntimeError):         # the product path fails loudly without a GPU
        m(torch.zeros(1, 8192), torch.zeros(1, 4))


def test_checkpoint_roundtrip(tmp_path):
    from signaltrain_amd import nn_proc, misc, audio
    nn_proc._QUIET = True
    m = nn_proc.st_model(num_knobs=4)
    eff = audio.Compressor_4c()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    f = str(tmp_path / "modelcheckpoint.tar")
    misc.save_checkpoint(f, m, 4, False, opt, eff, 44100)
    sd, rv = misc.load_checkpoint(f, device="cpu")
    assert set(rv) == {"epoch", "optimizer", "effect_name", "knob_names", "knob_ranges", "scale_factor", "shrink_factor",
                       "in_chunk_size", "out_chunk_size", "sr"}           # misc.py:28-34 keys
    assert rv["epoch"] == 5 and rv["in_chunk_size"] == 8192 and rv["out_chunk_size"] == 2048
    m2 = nn_proc.st_model(num_knobs=4); m2.load_state_dict(sd)
    for k in sd:
        assert torch.equal(m2.state_dict()[k], m.state_dict()[k])


def test_learningrate_and_loss_mirrors(golden_dir):
    from signaltrain_amd import learningrate, loss_functions
    g = np.load(os.path.join(golden_dir, "g6_1cycle.npz"))
    lr, mom = learningrate.get_1cycle_schedule(lr_max=1e-4, n_data_points=200000, epochs=100, batch_size=200)
    np.testing.assert_allclose(lr[g["idx"]], g["lr"], rtol=1e-15); np.testing.assert_allclose(mom[g["idx"]], g["mom"], rtol=1e-15)
    rng = np.random.default_rng(0)
    yh, y, mh = rng.standard_normal((3, 64)), rng.standard_normal((3, 64)), rng.standard_normal((3, 9, 513))
    w = O.freq_weights(513, np.float64)
    ref = O.calc_loss(yh, y, mh, w)
    got = loss_functions.calc_loss(torch.from_numpy(yh), torch.from_numpy(y), torch.from_numpy(mh), scale_by_freq=torch.from_numpy(w))
    assert abs(float(got) - ref) < 1e-12


def test_audio_and_dataset_contract(golden_dir):
    from oracle import host_audio
    from signaltrain_amd import audio, datasets
    g = np.load(os.path.join(golden_dir, "g9_compressor.npz"))
    y = audio.compressor_4controls(g["x"], *g["knobs"][:4], sr=g["knobs"][4])
    assert np.abs(y - g["y"]).max() < 1e-6
    np.testing.assert_array_equal(host_audio.sliding_window(np.arange(10), 5, overlap=2), [[0, 1, 2, 3, 4], [3, 4, 5, 6, 7], [6, 7, 8, 9, 0]])
    # the checker-side item assembly (datasets.py:312-334) over the product's host effect: shapes, dtypes, knob range
    np.random.seed(1)
    x, yy, k = host_audio.batch(2, 8192, audio.Compressor_4c(), 2048)
    assert x.shape == (2, 8192) and yy.shape == (2, 2048) and k.shape == (2, 4) and x.dtype == np.float32 and k.dtype == np.float32
    assert np.all(np.abs(k) <= 0.5) and np.abs(x).max() < 1.5
    # the torch Dataset contract of the reference (datasets.py:305-334): items behind __getitem__ / a DataLoader, recycle=True keeps them fixed.
    # Here (no GPU) the items come from the CPU-device form of the batched generators + the effect's host go()
    ds = datasets.SynthAudioDataSet(8192, audio.Compressor_4c(), y_size=2048, item_chunk=4)
    assert len(ds) == 8000
    xi, yi, ki = ds[0]
    assert xi.shape == (8192,) and yi.shape == (2048,) and ki.shape == (4,) and xi.dtype == np.float32 and ki.dtype == np.float32
    assert np.all(np.abs(ki) <= 0.5) and 0.05 < np.abs(xi).max() < 1.5 and np.isfinite(yi).all()
    # the target IS the effect of the input at those knobs (up to the polarity flip of do_augment, applied to both)
    yy2 = audio.Compressor_4c().go(xi, ki)[0][-2048:]
    assert np.abs(yy2 - yi).max() < 1e-5
    xb, yb, kb = next(iter(torch.utils.data.DataLoader(ds, batch_size=3, num_workers=0)))
    assert tuple(xb.shape) == (3, 8192) and tuple(yb.shape) == (3, 2048) and tuple(kb.shape) == (3, 4)
    rec = datasets.SynthAudioDataSet(8192, audio.Compressor_4c(), datapoints=5, y_size=2048, recycle=True, item_chunk=4)
    assert rec.x.shape == (5, 8192) and rec.y.shape == (5, 2048) and np.array_equal(rec[3][0], rec[3][0]) and not np.array_equal(rec[3][0], rec[4][0])


def test_cpu_port_matches_oracle():
    """bench.py's cpu_baseline 'port' computes the same step as the oracle."""
    from oracle.torch_cpu_step import CpuPort
    from tests.golden_util import perturb_stft
    geo = O.geometry(1, 4)
    rng = np.random.default_rng(3)
    P = O.init_params(geo, 4, rng); perturb_stft(P, seed=5)
    X, Y, KN = O.synth_comp4c_batch(2, geo["L"], geo["y"], rng)
    port = CpuPort(P, lr=1e-3)
    Pq = {k: P[k].copy() for k in O.param_order()}
    M = {k: np.zeros_like(v) for k, v in Pq.items()}; V = {k: np.zeros_like(v) for k, v in Pq.items()}
    for it in range(2):
        lp = port.step(torch.from_numpy(X), torch.from_numpy(KN), torch.from_numpy(Y), 1e-3)
        lo, _, _ = O.train_step(X, KN, Y, Pq, M, V, it + 1, 1e-3, geo)
        assert abs(lp - lo) <= 3e-5 * abs(lo)
        for k in Pq:
            assert np.abs(port.P[k].detach().numpy() - Pq[k]).max() < 2e-5, k   # Adam normalises noise-level gradient elements: lr-sized differences are expected there


def test_no_cross_block_mfma_result_hazards():
    """ISA regression check (tools/check_mfma_hazards.py): no MFMA result is first read behind a basic-block
    boundary with fewer wait states than the compiler pads inside a block.  Round 1 found exactly that in
    ae_bwd_kernel (a run-time stage-timer branch after an MFMA chain): timing-dependent wrong results."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_hazards.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]


def test_hazard_scanner_flags_the_known_bad_pattern():
    """Self-test of tools/check_mfma_hazards.py on hand-written ISA: an accumulator read behind a conditional branch 3
    instructions after the MFMA (the round-1 bug) must be flagged; the same read behind an s_nop 9 in one block, or a
    dependent MFMA, must not."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("chk", os.path.join(ROOT, "tools", "check_mfma_hazards.py"))
    chk = importlib.util.module_from_spec(spec); spec.loader.exec_module(chk)
    bad = """
	v_mfma_f32_16x16x4_f32 a[0:3], v9, v159, a[0:3]
	s_and_saveexec_b64 s[2:3], s[68:69]
	s_cbranch_execz .LBB0_2
	s_memtime s[34:35]
.LBB0_2:
	s_or_b64 exec, exec, s[2:3]
	v_accvgpr_read_b32 v9, a3
""".splitlines()
    good = """
	v_mfma_f32_16x16x4_f32 a[0:3], v8, v131, a[0:3]
	v_mfma_f32_16x16x4_f32 a[0:3], v9, v159, a[0:3]
	s_nop 9
	v_accvgpr_read_b32 v9, a3
	s_cbranch_execz .LBB0_2
.LBB0_2:
	v_add_f32_e32 v1, v9, v9
""".splitlines()
    assert len(chk.scan_kernel("bad", bad)) == 1
    assert chk.scan_kernel("good", good) == []


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The drop-in boundary is a C ABI: include/signaltrain_hip.h compiles as strict C99 (no C++, no torch types), and a C program
    linked against libsignaltrain_hip.so can call the host-only entries (geometry, layout sizes, error reporting) without a GPU."""
    import shutil, subprocess, sys
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use_abi.c"
    src.write_text('''
#include <stdio.h>
#include "signaltrain_hip.h"
int main(void)
{
    st_dims d;
    long long offs[40];
    if (st_geometry(1.0f, 4.0f, 0, 4, 256, &d) != ST_OK) { printf("geometry failed: %s\\n", st_last_error()); return 1; }
    d.prec = ST_PREC_F32X3; d.loss_scale = 0.0f; d.clip_all = 0;
    printf("%d %d %d %d %d %lld %zu\\n", d.L, d.T, d.OT, d.F, d.y, (long long)st_param_offsets(&d, (int64_t*)offs), st_workspace_bytes(&d));
    d.B = -1;
    if (st_workspace_bytes(&d) != 0) return 2;                      /* bad dims are rejected on the host ... */
    if (st_last_error()[0] == 0) return 3;                          /* ... with a message */
    return 0;
}
''')
    exe = tmp_path / "use_abi"
    lib_dir = os.path.join(root, "signaltrain_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src),
                        "-o", str(exe), "-L", lib_dir, "-l:libsignaltrain_hip.so", "-Wl,-rpath," + lib_dir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-500:])
    L, T, OT, F, y, total, wsb = (int(v) for v in r.stdout.split())
    assert (L, T, OT, F, y) == (8192, 25, 9, 513, 2048) and total == 4211096 and wsb > 100e6


def test_tuning_defaults_are_frozen():
    """The shipped values of the library's diagnostic switches (ST_TUNING_LIST in st_api.hip), in st_get_tuning's order.  A changed default has to be
    changed HERE as well -- it cannot ship unnoticed -- and tests/conftest.py asserts after every test that the live state equals these values."""
    lib = _lib.load()
    n = lib.st_get_tuning(None, 0)
    cur, dflt = (C.c_int * n)(), (C.c_int * n)()
    assert lib.st_get_tuning(cur, n) == n and lib.st_tuning_defaults(dflt, n) == n
    frozen = [0, -1, 1, 1, 0, 0, 0, 3, 1, 64, 0, 0, 0, 1, 1, 32, 1, 0, 1, 1, 0, 32, 16, 16, 200, 4, 3, 3, 1, 0, 0, 1, 1, 3, 15]
    assert list(dflt) == frozen and list(cur) == frozen
    # a switch is visible in the state and reset restores it; timing-only ablations (invalid results) are not in the product build
    assert lib.st_set_tuning(9500) == 0 and list((lib.st_get_tuning(cur, n), cur)[1]) != frozen
    assert lib.st_reset_tuning() == 0 and list((lib.st_get_tuning(cur, n), cur)[1]) == frozen
    assert lib.st_set_tuning(9681) != 0 and lib.st_set_tuning(96801) != 0 and b"ST_DIAG" in lib.st_last_error()


def test_kept_activation_bytes_follow_geometry_and_arithmetic():
    """Round 6: what a training-step forward keeps for the backward (st_ae_kept_activation_bytes) -- 2 nets x B * ceil(F / 16) row groups x 17 tiles of 1 KB where the
    autoencoder layers run in fp32 on the fused kernels, nothing elsewhere (16-bit layers, the wide path, the recompute switch) -- and the autoencoder workspace has room for it."""
    lib = _lib.load()
    d = _lib.st_dims()
    for B in (1, 3, 256, 1024):
        assert lib.st_geometry(1.0, 4.0, 0, 4, B, C.byref(d)) == 0
        for prec, keeps in ((0, True), (1, True), (3, True), (5, True), (2, False), (4, False)):       # f32, bf16 GEMMs, f16 GEMMs, f32x3 keep fp32 layers; bf16_all / f16_all do not
            d.prec = prec
            want = 2 * B * ((d.F + 15) // 16) * 17 * 1024 if keeps else 0
            assert lib.st_ae_kept_activation_bytes(C.byref(d)) == want, (B, prec)
            if keeps: assert lib.st_ae_bwd_ws_floats(C.byref(d)) * 4 > want and lib.st_workspace_bytes(C.byref(d)) > want
        d.prec = 0
        try:
            assert lib.st_set_tuning(8200) == 0 and lib.st_ae_kept_activation_bytes(C.byref(d)) == 0
        finally:
            assert lib.st_set_tuning(8201) == 0
    assert lib.st_geometry(8.0, 4.0, 0, 4, 64, C.byref(d)) == 0      # the 65536-sample window: wide path, feature-major activations of its own
    d.prec = 0
    assert lib.st_ae_kept_activation_bytes(C.byref(d)) == 0


def test_effective_precision_is_reported():
    """st_effective_prec = the arithmetic a call really runs.  Until round 4 the wide autoencoder path (T > 32 or OT > 16: the 65536-sample window, lean
    scale 2, shrink 1) needed an EVEN batch for 16-bit Linear layers and ran fp32 layers for odd ones -- reported, but a cliff.  Round 5: its weight-gradient
    GEMMs take 16-deep k-tiles when B * 528 is 16 mod 32, so the effective arithmetic equals the request for every geometry and every batch."""
    lib = _lib.load()
    d = _lib.st_dims()
    for scale, shrink in ((8.0, 4.0), (2.0, 4.0), (1.0, 1.0), (1.0, 4.0)):
        for B in (1, 3, 9, 63, 64):
            assert lib.st_geometry(scale, shrink, 0, 4, B, C.byref(d)) == 0
            for prec in (0, 1, 2, 3, 4, 5):
                d.prec = prec
                assert lib.st_effective_prec(C.byref(d)) == prec, (scale, shrink, B, prec)


@pytest.mark.parametrize("B,shrink", [(256, 4), (128, 4), (192, 4), (64, 4), (3, 4), (1, 4), (256, 8), (100, 2), (512, 4), (384, 4), (1024, 4)])
def test_nt128_worklist_covers_exactly_the_live_taps(B, shrink):
    """Round 5: the 128 x 128-tile synthesis GEMMs no longer multiply the structural zeros of the cropped transposed convolution
    (cls_fe_dft.py:112-113: of output frame t' only the taps n with N <= H t' + n < N + y survive the crop).  Host logic only (st_nt128_worklist):
    frames GEMM -- every live (frame, tap) lies in a listed tile, each listed tile gets its whole reduction exactly once per active slab and zeros
    in the others; data gradient -- every tile row's slices partition a k range that holds all its frames' live taps, every spectral column is
    produced exactly once per slab (tiles + the Nyquist entries), unused slabs are zero-filled.  B = 256: one workgroup per CU; B = 384 / 512: several
    rounds of the same list (the k-slice count is the cheapest by the builder's cost model)."""
    lib = _lib.load()
    ncus = 256
    d = _lib.geometry(1, shrink, 4, B)
    N, H, OT, y, F = d.N, d.H, d.OT, d.y, d.F
    KP = lib.st_kp(F)
    live_t = [t for t in range(OT) if H * t + N > N and H * t < N + y]        # frames with at least one tap inside the crop [N, N + y)
    t_lo, Tv = live_t[0], len(live_t)
    assert live_t == list(range(t_lo, t_lo + Tv))
    R = B * Tv

    def tap_range(t):
        return max(0, N - H * t), min(N, N + y - H * t)

    def entries(which):
        out = (C.c_uint * 1024)(); head = (C.c_int * 6)()
        n = lib.st_nt128_worklist(C.byref(d), which, ncus, out, 1024, head)
        assert 0 <= n <= 768
        ku = head[4] if n else 1
        ents = [dict(mt=out[i] & 255, nt=(out[i] >> 8) & 63, z=(out[i] >> 14) & 3, zf=(out[i] >> 16) & 3, kind=(out[i] >> 18) & 1,
                     k0=((out[i] >> 19) & 63) * ku, kl=(out[i] >> 25) * ku) for i in range(n)]
        for e in ents:
            e["k1"] = min(e["k0"] + e["kl"], head[5])
        return n, ents, list(head)

    # ---------------------------------------------------------------- frames GEMM: C[row][tap] over k in [0, KP)
    n, ents, head = entries(0)
    nsl = lib.st_synth_frame_slabs(C.byref(d))
    if n:
        assert head[0] == nsl and head[1] == 0 and head[3] == B and head[5] == KP // 32 and all(e["kind"] == 0 for e in ents)
        tiles = {}
        for e in ents:
            tiles.setdefault((e["mt"], e["nt"]), []).append(e)
        nacts = set()
        for (mt, nt), es in tiles.items():
            es.sort(key=lambda e: e["z"])
            nact = len(es); nacts.add(nact)
            assert [e["z"] for e in es] == list(range(nact)) and nact <= nsl
            assert es[0]["k0"] == 0 and es[-1]["k1"] == KP // 32
            assert all(es[i]["k1"] == es[i + 1]["k0"] for i in range(nact - 1))
            assert es[0]["zf"] == (nact if nact < nsl else 0) and all(e["zf"] == 0 for e in es[1:])       # slabs [nact, nsl) zero-filled once, by slice 0
        assert len(nacts) == 1
        for r in range(R):                                     # frame-major compact rows
            t = t_lo + r // B
            lo, hi = tap_range(t)
            for ntile in range(N // 128):
                if lo < 128 * ntile + 128 and hi > 128 * ntile:
                    assert (r // 128, ntile) in tiles, (r, t, ntile)
        if B % 128 == 0:                                       # one frame per tile row: nothing but live tiles is computed
            for (mt, nt) in tiles:
                lo, hi = tap_range(t_lo + (128 * mt) // B)
                assert lo < 128 * nt + 128 and hi > 128 * nt
        if (B, shrink) == (256, 4):
            assert len(tiles) == 84 and n == 252               # 84 of 112 tiles, three k-slices: one workgroup per CU
        if (B, shrink) == (512, 4):
            assert len(tiles) == 168 and n == 504              # two rounds of three k-slices
        if (B, shrink) == (384, 4):
            assert len(tiles) == 126 and n == 252              # two k-slices: one full round beats three slices in two rounds
    else:
        assert B >= 1024 or nsl > 3                            # B = 1024: 336 tiles, one slab -> a second round at 31 %: stays on gemm_kernel<4, ...>
    # ---------------------------------------------------------------- data gradient: C[row][spectral column] over the taps
    n, ents, head = entries(1)
    nsl = lib.st_synth_slabs(C.byref(d))
    if n:
        assert head[0] == nsl
        col_h, col_stride = head[1], head[2]
        assert (col_h, col_stride) == (F - 1, KP // 2)         # N = 1024: the Nyquist columns are kind-1 entries
        MT = (R + 127) // 128
        for mt in range(MT):
            rows = range(128 * mt, min(128 * mt + 128, R))
            lo = min(tap_range(t_lo + r // B)[0] for r in (rows[0], rows[-1])); hi = max(tap_range(t_lo + r // B)[1] for r in (rows[0], rows[-1]))
            mine = [e for e in ents if e["mt"] == mt]
            nyq = [e for e in mine if e["kind"] == 1]
            assert len(nyq) == 1 and 32 * nyq[0]["k0"] <= lo and 32 * nyq[0]["k1"] >= hi
            cols = sorted(set(e["nt"] for e in mine if e["kind"] == 0))
            assert cols == list(range(2 * (F - 1) // 128))
            for nt in cols:
                es = sorted((e for e in mine if e["kind"] == 0 and e["nt"] == nt), key=lambda e: e["z"])
                sl = len(es)
                assert [e["z"] for e in es] == list(range(sl)) and sl <= nsl and es[0]["zf"] == (sl if sl < nsl else 0) and all(e["zf"] == 0 for e in es[1:])
                assert 32 * es[0]["k0"] <= lo and 32 * es[-1]["k1"] >= hi
                assert all(es[i]["k1"] == es[i + 1]["k0"] for i in range(sl - 1))
                if B % 128 == 0:                               # no k-tile outside the live taps
                    assert 32 * es[0]["k0"] >= lo - 31 and 32 * es[-1]["k1"] <= hi + 31
        if (B, shrink) == (256, 4):
            assert n == 240 + 14 and max(e["kl"] for e in ents if e["kind"] == 0) == 12      # 3 / 2 / 1 slices of <= 384 taps, Nyquist columns on the 14 spare CUs
        if (B, shrink) == (512, 4):
            assert n == 480 + 28 and max(e["kl"] for e in ents if e["kind"] == 0) == 12
    else:
        assert B > 1024


def test_frame_major_division_bound_and_worklist_is_a_pure_function():
    """ADVICE round 5.  (1) The frame-major row order splits a compact row r into (r // B, r % B) with a multiply-high by floor(2^32 / B) + 1
    (st_gemm.h RowMap::magic); that is exact only while r * B < 2^32.  st_fm_div_exact is the host-side gate every frame-major switch asks: where it
    says yes the multiply-high is exact for the rows near every multiple of B up to R (brute force on the boundary rows, where it fails first), and
    just past the bound a wrong quotient exists (B = 32768 at 7 live frames: r >= 2^17 + ...).  (2) st_nt128_worklist with an explicit CU count
    is a pure function of its arguments: two different counts give the lists their own cost model asks for, never the probed device's."""
    lib = _lib.load()

    def mulhi_div(r, B):
        magic = (1 << 32) // B + 1
        return (r * magic) >> 32

    def first_wrong(R, B):
        for q in range(1, R // B + 1):            # the quotient is first wrong just below a multiple of B (or at it)
            for r in (q * B - 1, q * B):
                if r < R and mulhi_div(r, B) != r // B:
                    return r
        return None

    for B, Tv in [(256, 7), (1024, 23), (4096, 7), (24000, 7), (32768, 7), (40000, 7), (65536, 23), (9000, 46)]:
        R = B * Tv
        ok = lib.st_fm_div_exact(R, B)
        assert ok == (1 if R * B < (1 << 32) else 0)
        if ok:
            assert first_wrong(R, B) is None
    assert lib.st_fm_div_exact(7 * 32768, 32768) == 0 and first_wrong(7 * 32768, 32768) is not None       # the advisor's case really is wrong without the gate
    # (2) purity in ncus: B = 3 has 1 tile row -- with 8 "CUs" the one-round rule fails, the crop rule (>= 10 % of the taps) still holds: the list does not depend on a device
    d = _lib.geometry(1, 4, 4, 256)
    out = (C.c_uint * 1024)(); head = (C.c_int * 6)()
    n256 = lib.st_nt128_worklist(C.byref(d), 0, 256, out, 1024, head); a = list(out[:n256])
    n256b = lib.st_nt128_worklist(C.byref(d), 0, 256, out, 1024, head); b = list(out[:n256b])
    n304 = lib.st_nt128_worklist(C.byref(d), 0, 304, out, 1024, head)
    assert n256 == n256b == 252 and a == b and n304 > 0


def test_run_train_accepts_the_reference_effect_keys():
    """run_train.py --effect: the reference's keys (run_train.py:55-80).  comp_4c / comp_large / files are built (argument parsing only here: the run itself
    needs a GPU); the reference's other keys and unknown ones exit with a message that names what is available, as the reference does for unknown effects."""
    import subprocess, sys
    run = lambda *a: subprocess.run([sys.executable, os.path.join(ROOT, "run_train.py"), *a], capture_output=True, text=True, timeout=300)
    r = run("--effect", "denoise")
    assert r.returncode != 0 and "does not build" in r.stderr and "comp_large" in r.stderr
    r = run("--effect", "bogus")
    assert r.returncode != 0 and "is not yet added" in r.stderr
    r = run("--effect", "comp_large", "--target", "nope")
    assert r.returncode != 0 and "invalid target type" in r.stderr            # comp_large passed the effect check (argparse used to reject it)


def test_dims_edge_cases_are_refused_with_a_message():
    """Edge cases at the C ABI, before any launch (host-side validation only: runs without a GPU).  Empty batch, a geometry that is not the reference's
    (F != N/2 + 1, y != (OT - 1) H - N: nn_proc.py:357-385), more knobs than the fused kernels carry, an unknown arithmetic level, a negative / infinite loss
    scale, and the MAXIMUM sizes: batches whose row count or operand sizes leave the 24-bit row / 32-bit element indexing of the kernels (include/signaltrain_hip.h)
    are refused by name instead of wrapping around.  The largest admitted batch still reports a layout and a workspace size."""
    import ctypes as C
    lib = _lib.load()
    good = _lib.geometry(1, 4, 4, 256)
    assert lib.st_param_offsets(C.byref(good), None) > 0 and lib.st_workspace_bytes(C.byref(good)) > 0

    def refused(mut, needle):
        d = good.with_batch(good.B); mut(d)
        assert lib.st_param_offsets(C.byref(d), None) == -1, needle
        msg = lib.st_last_error().decode()
        assert needle in msg, (needle, msg)
        # a compute entry refuses the same dims before it looks at any pointer
        assert lib.st_analysis_fwd(C.byref(d), None, None, None, 0.5, None, None, None, None, None) < 0
    refused(lambda d: setattr(d, "B", 0), "non-positive")
    refused(lambda d: setattr(d, "B", -3), "non-positive")
    refused(lambda d: setattr(d, "K", -1), "non-positive")
    refused(lambda d: setattr(d, "K", 17), "at most 16 knobs")
    refused(lambda d: setattr(d, "F", d.F - 1), "F must be N/2+1")
    refused(lambda d: setattr(d, "y", d.y + 4), "y must equal")
    refused(lambda d: setattr(d, "OT", d.T + 1), "y must equal")           # OT > T also breaks y = (OT - 1) H - N first
    refused(lambda d: setattr(d, "prec", 99), "ST_PREC")
    refused(lambda d: setattr(d, "loss_scale", -1.0), "loss_scale")
    refused(lambda d: setattr(d, "loss_scale", float("inf")), "loss_scale")
    refused(lambda d: setattr(d, "L", d.L + 2), "required")                # ragged window length: L % 4
    # maximum sizes: rows B * T < 2^24, B * T * KP and B * (L + 2 N) < 2^30 elements
    kp = int(lib.st_kp(good.F))
    b_max = min(((1 << 24) - 1) // good.T, ((1 << 30) - 1) // (good.T * kp), ((1 << 30) - 1) // (good.L + 2 * good.N))
    ok = good.with_batch(b_max)
    assert lib.st_param_offsets(C.byref(ok), None) > 0, lib.st_last_error()
    assert lib.st_workspace_bytes(C.byref(ok)) > lib.st_workspace_bytes(C.byref(good))
    refused(lambda d: setattr(d, "B", b_max + 1), "batch too large")
    refused(lambda d: setattr(d, "B", 1 << 30), "batch too large")
    # the 65536-sample window has its own, smaller limit
    wide = _lib.geometry(8, 4, 4, 64)
    kpw = int(lib.st_kp(wide.F))
    bw = min(((1 << 24) - 1) // wide.T, ((1 << 30) - 1) // (wide.T * kpw), ((1 << 30) - 1) // (wide.L + 2 * wide.N))
    assert bw < b_max and lib.st_param_offsets(C.byref(wide.with_batch(bw)), None) > 0
    assert lib.st_param_offsets(C.byref(wide.with_batch(bw + 1)), None) == -1 and b"batch too large" in lib.st_last_error()
    # a null dims pointer
    assert lib.st_param_offsets(None, None) == -1 and b"null dims" in lib.st_last_error()


def test_workspace_of_the_fp32_level_bounds_every_arithmetic_level():
    """st_workspace_bytes depends on the arithmetic level of the dims (fp32 autoencoder layers keep their activations for the backward, the 16-bit levels carry
    operand copies and the h4 exchange of the split backward).  Hosts size ONE workspace and may switch levels on a live engine (StepEngine.set_arithmetic,
    st_model.set_compute_dtype): the size reported for ST_PREC_F32 must bound every other level's, at every geometry / batch / knob count / clip scope --
    and StepEngine sizes by the maximum anyway."""
    import ctypes as C
    lib = _lib.load()
    for scale, scheme in ((1, "lean"), (2, "lean"), (2, "legacy"), (8, "lean")):
        for shrink in (1, 4, 8):
            for B in (1, 3, 64, 130, 256, 257, 1024):
                for K in (0, 4, 16):
                    d = _lib.geometry(scale, shrink, K, B, scale_scheme=scheme)
                    base = lib.st_workspace_bytes(C.byref(d.with_arith(prec=_lib.PREC["f32"], loss_scale=0.0, clip_all=0)))
                    assert base > 0
                    for name, p in _lib.PREC.items():
                        for ca in (0, 1):
                            n = lib.st_workspace_bytes(C.byref(d.with_arith(prec=p, loss_scale=(4096.0 if name.startswith("f16") else 0.0), clip_all=ca)))
                            assert 0 < n <= base, (scale, scheme, shrink, B, K, name, ca, n, base)


def test_one_workspace_serves_every_smaller_batch():
    """st_workspace_bytes is NOT monotonic in the batch: the split-K slab counts of the weight-gradient / synthesis GEMMs are picked per batch, so e.g. 585 windows of 8192
    samples need 85.6 MB MORE than 586 do, and 93 windows of 65536 samples 63 MB more than 94.  A host that sizes its workspace for its largest batch and then runs a smaller
    one (a last partial batch, predict_long's remainder, a validation batch) would be written past the end.  engine.workspace_bytes_upto is what StepEngine allocates: the
    maximum over every batch 1 .. max_batch, every arithmetic level and clip scope -- the C entry st_workspace_bytes_max (INTEGRATION.md "Sizing the workspace")."""
    import ctypes as C
    from signaltrain_amd.engine import workspace_bytes_upto
    lib = _lib.load()
    seen_non_monotonic = False
    for scale, shrink, bmax in ((1, 4, 586), (1, 4, 600), (1, 4, 256), (1, 1, 179), (8, 4, 94), (8, 4, 64), (2, 4, 342)):
        d = _lib.geometry(scale, shrink, 4, bmax)
        cap = workspace_bytes_upto(lib, d, bmax)
        own = {name: lib.st_workspace_bytes(C.byref(d.with_arith(prec=p))) for name, p in _lib.PREC.items()}
        assert cap >= max(own.values())
        for b in range(1, bmax + 1):
            for name, p in _lib.PREC.items():
                n = lib.st_workspace_bytes(C.byref(d.with_batch(b).with_arith(prec=p, clip_all=int(name.startswith("f16")))))
                assert 0 < n <= cap, (scale, shrink, bmax, b, name, n, cap)
                seen_non_monotonic |= n > own[name]
    assert seen_non_monotonic          # the property this test exists for (if the library ever becomes monotonic, drop this line and the loop stays a valid guarantee)
    # st_workspace_bytes_max IS that maximum (the C entry the engine calls), also from dims that carry a 16-bit level, and 0 for bad dims
    d = _lib.geometry(1, 4, 4, 600)
    brute = max(lib.st_workspace_bytes(C.byref(d.with_batch(b).with_arith(prec=p, clip_all=ca))) for b in range(1, 601) for p in _lib.PREC.values() for ca in (0, 1))
    assert lib.st_workspace_bytes_max(C.byref(d)) == brute == lib.st_workspace_bytes_max(C.byref(d.with_arith(prec=_lib.PREC["bf16_all"], clip_all=1)))
    assert lib.st_workspace_bytes_max(C.byref(d.with_batch(0))) == 0
