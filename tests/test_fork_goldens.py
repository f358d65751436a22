"""Pinning parity to the REAL fork (emjay73/diff_gaussian_rasterization_with_depth, reference .gitmodules:7-9).

The fork is an un-vendored, CUDA-only submodule: nothing under /root/reference can produce a rasterizer output, so the
oracle is "parity unpinned" (oracle/vr_oracle.c header).  tools/fork_pin/dump_fork_goldens.py is a standalone script
that anyone with an NVIDIA box and the reference environment runs once; it writes tests/golden/fork/fork_case_*.npz
(outputs + gradients of the real fork for the committed INPUT cases tests/golden/raster_case_*.npz).  With those files
present these tests
  * find which of the 16 combinations of the fork switches (VrFlags bits 0-3) reproduces the fork's
    forward within 1e-4 abs (the north star's tolerance) on every case,
  * check the oracle's (CPU) and the HIP kernels' (GPU) forward and gradients against the fork under that combination.
Without the files the fork tests SKIP; the kit's own machinery (matching, reporting) is still tested against stand-in
files written by the oracle under a known flag combination.
"""
import glob
import os

import numpy as np
import pytest

from helpers import GOLDEN, OUT_NAMES, assert_grad_close, case_gouts, case_inputs, oracle_cam_from_case

FORK_DIR = os.path.join(GOLDEN, "fork")
FWD_ATOL = 1e-4            # BASELINE.json north_star: "outputs within 1e-4 abs"
# ... EXCEPT threshold flips.  A fragment whose alpha sits within an ulp of 1/255 is classified one way by one correct exp()
# and the other way by another (CUDA's expf, this build's polynomial, v_exp_f32 all differ in the last place): its pixel
# then moves by up to alpha * |attribute| ~ 1/255 of a channel -- far above 1e-4 and no defect of either side.  The rule is
# the one tests/test_gpu_parity.py applies to VR_FLAG_FAST_EXP against the checker: a BOUNDED number of such pixels
# (<= max(2, 2e-5 H W) per case), each off by at most 1.01 / 255 x the channel's scale; they are printed.
FLIP_FRACTION = 2e-5
FLIP_BOUND = 1.01 / 255.0


def forward_deviation(out, ref, atol=FWD_ATOL, report=None):
    """Worst |out - ref| over the five images of one case OUTSIDE the pixels excused as threshold flips; inf when the
    flips exceed their count or size bound.  `report` (a list) receives one line per excused pixel."""
    H, W = out["color"].shape[-2:]
    bad = np.zeros((H, W), bool)
    worst = 0.0
    per = {}
    for n in OUT_NAMES:
        a = out[n].astype(np.float64).reshape(-1, H, W)
        b = ref[n].astype(np.float64).reshape(-1, H, W)
        d = np.abs(a - b)
        scale = max(1.0, float(np.abs(b).max()))
        if float(d.max(initial=0.0)) > FLIP_BOUND * scale:
            return float("inf")
        off = (d > atol).any(axis=0)
        bad |= off
        per[n] = d
        worst = max(worst, float(d[:, ~off].max(initial=0.0)))
    if int(bad.sum()) > max(2, int(FLIP_FRACTION * H * W)):
        return float("inf")
    if report is not None:
        for y, x in zip(*np.nonzero(bad)):
            report.append(f"   threshold-flip pixel ({x}, {y}): " +
                          ", ".join(f"{n} {float(per[n][:, y, x].max()):.2e}" for n in OUT_NAMES if per[n][:, y, x].max() > atol))
    return worst


GRAD_NAMES = ["means3D", "means2D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp"]


def _cases(fork_dir):
    out = []
    for f in sorted(glob.glob(os.path.join(fork_dir, "fork_case_*.npz"))):
        name = os.path.basename(f)[len("fork_"):-len(".npz")]
        src = os.path.join(GOLDEN, f"raster_{name}.npz")
        if os.path.exists(src):
            out.append((name, {k: v for k, v in np.load(src).items()}, {k: v for k, v in np.load(f).items()}))
    return out


def _oracle_run(c, flags, three):
    from oracle import oracle as orc
    import ctypes
    cam = oracle_cam_from_case(c)
    cam.flags = ctypes.c_uint(flags & 0xF).value
    out, st = orc.forward(cam, **case_inputs(c))
    which = ("color", "cov_quat", "cov_scale") if three else OUT_NAMES
    full = dict(zip(OUT_NAMES, case_gouts(c)))          # (a cropped fixture's upstream gradients, padded to the frame)
    g = [full[n] if n in which else None for n in OUT_NAMES]
    return out, orc.backward(cam, st, *g)


def matching_flag_sets(cases, atol=FWD_ATOL):
    """Flag combinations (0..15) under which the oracle's forward equals the fork's on EVERY case; plus the per-combination
    worst deviation for the report."""
    worst = {}
    for flags in range(16):
        w = 0.0
        for name, c, f in cases:
            out, _ = _oracle_run(c, flags, True)
            if not np.array_equal(out["radii"], f["radii"]):
                w = float("inf")
                break
            notes = []
            w = max(w, forward_deviation(out, {n: f["out_" + n].reshape(out[n].shape) for n in OUT_NAMES}, atol, notes))
            if notes and w <= atol:
                print(f"flags {flags}, {name}: {len(notes)} pixel(s) excused as threshold flips")
                print("\n".join(notes))
        worst[flags] = w
    return [k for k, v in worst.items() if v <= atol], worst


def _describe(flags):
    names = ["SCALE_MODIFIED", "DEPTH_NORMALIZED", "EXTRA_NO_ALPHA_GRAD", "FILL_EMPTY"]
    return " | ".join(n for i, n in enumerate(names) if flags >> i & 1) or "0 (the documented defaults)"


def check_oracle_against_fork(cases):
    ok, worst = matching_flag_sets(cases)
    print("forward deviation of the oracle from the fork per flag combination:")
    for k, v in sorted(worst.items(), key=lambda kv: kv[1]):
        print(f"   flags {k:2d} = {_describe(k):60s} max |diff| {v:.3g}")
    assert ok, "no combination of the fork switches reproduces the fork's forward within 1e-4: " \
               "the fork differs in a way SURVEY.md A.8 does not model"
    # EXTRA_NO_ALPHA_GRAD (bit 2) does not change the forward: the gradients decide it
    verdicts = []
    for flags in ok:
        good = True
        for name, c, f in cases:
            for tag, three in (("grad3", True), ("grad", False)):
                _, og = _oracle_run(c, flags, three)
                for k in GRAD_NAMES:
                    key = f"{tag}_{k}"
                    if key not in f or og.get(k) is None:
                        continue
                    try:
                        assert_grad_close(f"{name} {key} flags {flags}", og[k], f[key].reshape(og[k].shape), rtol=2e-3,
                                          floor=1e-5, outliers=0.0, near=0.0, cap=3.0)
                    except AssertionError:
                        # `grad` (all five upstream gradients) may legitimately differ if the fork's backward ignores
                        # dL/ddepth or dL/dalpha (A.8 item 7: nothing in VEGS feeds them); only grad3 decides
                        if three:
                            good = False
        verdicts.append((flags, good))
    winners = [k for k, g in verdicts if g]
    print("flag combinations matching forward AND the three-gradient backward:", [(k, _describe(k)) for k in winners])
    assert winners, f"forward matches under {ok} but no combination matches the fork's gradients"
    return winners


def _standin_fork_files(tmp_path, flags):
    """fork_case_*.npz as the dump script would write them, produced by the ORACLE under a known flag combination."""
    d = tmp_path / "fork"
    d.mkdir()
    for name in ("case_sh3", "case_cull_deg1"):
        c = {k: v for k, v in np.load(os.path.join(GOLDEN, f"raster_{name}.npz")).items()}
        out, g5 = _oracle_run(c, flags, False)
        _, g3 = _oracle_run(c, flags, True)
        res = {"out_" + n: out[n] for n in OUT_NAMES}
        res["radii"] = out["radii"]
        for tag, g in (("grad", g5), ("grad3", g3)):
            for k in GRAD_NAMES:
                if g.get(k) is not None:
                    res[f"{tag}_{k}"] = g[k]
        np.savez_compressed(d / f"fork_{name}.npz", **res)
    return str(d)


@pytest.mark.parametrize("flags", [0, 2 | 8, 1 | 4])
def test_kit_identifies_a_known_flag_combination(tmp_path, flags):
    """The matching machinery on stand-in files: whatever combination produced them is among the reported winners, and
    combinations that change the forward are rejected."""
    cases = _cases(_standin_fork_files(tmp_path, flags))
    assert len(cases) == 2
    winners = check_oracle_against_fork(cases)
    assert flags in winners
    for w in winners:
        # bit 0 (SCALE_MODIFIED) is invisible when the case's scale_modifier is 1; bits 1, 3 must agree, and bit 2
        # (gradient routing) too
        assert (w ^ flags) & ~1 == 0 or all(float(c["scale_modifier"]) == 1.0 for _, c, _ in cases) and (w ^ flags) & 0xE == 0


def test_threshold_flip_pixels_are_excused_but_bounded(tmp_path):
    """The forward comparison tolerates what two correct exp() implementations legitimately disagree on -- a fragment at
    the 1/255 threshold classified the other way moves ONE pixel by up to ~1/255 of a channel -- and nothing else: more
    such pixels than 2e-5 of the frame, a larger step, or a deviation spread over the image all fail the match."""
    d = _standin_fork_files(tmp_path, 0)
    f = os.path.join(d, "fork_case_sh3.npz")
    base = {k: v for k, v in np.load(f).items()}

    def winners_with(edit):
        z = {k: v.copy() for k, v in base.items()}
        edit(z)
        np.savez_compressed(f, **z)
        return matching_flag_sets(_cases(d))[0]

    def one_flip(z):
        z["out_color"][1, 7, 9] += 0.9 / 255.0
        z["out_depth"].reshape(z["out_depth"].shape[-2:])[7, 9] += 0.9 / 255.0 * max(1.0, float(np.abs(z["out_depth"]).max()))
    assert 0 in winners_with(one_flip)

    def too_large(z):
        z["out_color"][1, 7, 9] += 3.0 / 255.0
    assert winners_with(too_large) == []

    def too_many(z):
        z["out_color"][0, 3, ::2] += 0.5 / 255.0
    assert winners_with(too_many) == []

    def everywhere(z):
        z["out_color"] += 2e-4
    assert winners_with(everywhere) == []


def test_dump_script_is_standalone():
    """tools/fork_pin/dump_fork_goldens.py must run where this repository is not importable."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "tools", "fork_pin", "dump_fork_goldens.py")).read()
    mods = set()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Import):
            mods.update(a.name.split(".")[0] for a in node.names)
        elif isinstance(node, ast.ImportFrom):
            mods.add((node.module or "").split(".")[0])
    assert mods <= {"argparse", "glob", "os", "numpy", "torch", "diff_gaussian_rasterization"}, mods


def test_oracle_matches_the_fork():
    cases = _cases(FORK_DIR)
    if not cases:
        pytest.skip("tests/golden/fork/fork_case_*.npz absent: run tools/fork_pin/dump_fork_goldens.py on an NVIDIA box "
                    "(INTEGRATION.md section 6); until then parity with the fork is unpinned")
    check_oracle_against_fork(cases)


@pytest.mark.gpu
def test_hip_matches_the_fork():
    cases = _cases(FORK_DIR)
    if not cases:
        pytest.skip("tests/golden/fork/fork_case_*.npz absent (see test_oracle_matches_the_fork)")
    import torch
    from test_gpu_parity import _run_hip, _settings
    dev = torch.device("cuda:0")
    flags = check_oracle_against_fork(cases)[0]
    for name, c, f in cases:
        full = dict(zip(OUT_NAMES, case_gouts(c)))
        g = [full[n] if n in ("color", "cov_quat", "cov_scale") else None for n in OUT_NAMES]
        out, grads, _ = _run_hip(_settings(c, None, None, None, dev), case_inputs(c), dev, g, flags=flags)
        assert np.array_equal(out["radii"], f["radii"])
        notes = []
        dev_ = forward_deviation(out, {n: f["out_" + n].reshape(out[n].shape) for n in OUT_NAMES}, FWD_ATOL, notes)
        print("\n".join(notes))
        assert dev_ <= FWD_ATOL, (name, dev_)
        for k in GRAD_NAMES:
            if grads.get(k) is not None and f"grad3_{k}" in f:
                assert_grad_close(f"hip vs fork {name} {k}", grads[k], f[f"grad3_{k}"].reshape(grads[k].shape), rtol=2e-3, floor=1e-5)
