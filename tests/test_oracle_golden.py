"""Pins the CPU oracle: it must reproduce every golden PNG the reference commits."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_png_rgb

with open(os.path.join(GOLDEN, "manifest.json")) as _fh:
    _MANIFEST = json.load(_fh)
_GOLDENS = _MANIFEST["goldens"]


def _params(orc, e, w, h):
    return orc.OracleParams(
        width=w, height=h, chroma_a=e["chroma_a"], chroma_b=e["chroma_b"],
        y_bits=e["bits"][0], cb_bits=e["bits"][1], cr_bits=e["bits"][2],
        factor=e["factor"], op=tuple(e["op"]),
        rounding=orc.ROUND_TRUNC_SW if e["rounding"] == "TRUNC_SW" else orc.ROUND_FLOOR_HW)


def test_fixture_files_intact(manifest):
    for v in list(manifest["inputs"].values()) + manifest["goldens"]:
        data = open(os.path.join(GOLDEN, v["file"]), "rb").read()
        assert hashlib.sha256(data).hexdigest() == v["sha256"], v["file"]


@pytest.mark.parametrize("form", ["stream", "closed"])
@pytest.mark.parametrize("e", _GOLDENS, ids=[e["name"] for e in _GOLDENS])
def test_oracle_reproduces_golden(oracle, input_images, e, form):
    rgb_in = input_images[e["input"]]
    want = load_png_rgb(os.path.join(GOLDEN, e["file"]))
    assert hashlib.sha256(want.tobytes()).hexdigest() == e["px_sha256"]
    if e["rounding"] == "IDENTITY":          # readImage -> writeImage round trip, no arithmetic
        assert np.array_equal(want, rgb_in)
        return
    h, w = rgb_in.shape[:2]
    p = _params(oracle, e, w, h)
    got = oracle.argb_to_rgb(oracle.process(p, oracle.rgb_to_argb(rgb_in), form=form))
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{int((got != want).any(-1).sum())} mismatching pixels"


def test_app_golden_is_chroma_before_spatial(oracle, input_images, manifest):
    """The sf2 app golden matches every C-before-S order and no S-before-C order (595/4096 differ)."""
    e = next(g for g in manifest["goldens"] if g["name"] == "app_422_888_sf2_128")
    rgb_in = input_images[e["input"]]
    want = load_png_rgb(os.path.join(GOLDEN, e["file"]))
    argb = oracle.rgb_to_argb(rgb_in)
    S, Q, Cc = oracle.OP_SPATIAL, oracle.OP_QUANT, oracle.OP_CHROMA
    for op in [(Cc, S, Q), (Cc, Q, S), (Q, Cc, S)]:
        p = _params(oracle, dict(e, op=list(op)), 128, 128)
        assert np.array_equal(oracle.argb_to_rgb(oracle.process(p, argb)), want)
    for op in [(S, Cc, Q), (S, Q, Cc), (Q, S, Cc)]:
        p = _params(oracle, dict(e, op=list(op)), 128, 128)
        got = oracle.argb_to_rgb(oracle.process(p, argb))
        assert int((got != want).any(-1).sum()) == 595
