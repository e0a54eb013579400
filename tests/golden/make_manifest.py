#!/usr/bin/env python3
"""Collects the reference's committed input/output PNGs as golden fixtures.

Run ONCE in the build container (where /root/reference exists):
    python tests/golden/make_manifest.py
It copies image DATA only (3 inputs + 29 outputs, MIT licence, /root/reference/LICENSE) into
tests/golden/{inputs,outputs}/ under shell-safe names and writes manifest.json with, per golden:
the original path, file sha256, sha256 of the decoded HxWx3 RGB bytes, and the pipeline that
reproduces it.  No reference source text is copied.
Nothing at test/bench time reads /root/reference -- only the committed copies.
"""
import hashlib
import json
import os
import shutil

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

INPUTS = {
    "in16": "test_images/in16x16.png",
    "in128": "test_images/in128x128.png",
    "in512": "test_images/in512x512.png",
}

CSQ = [3, 1, 2]  # chroma, spatial, quant  (C-before-S class; ProcessingStep ordinals)


def g(path, name, inp, rounding, a, b, bits=(8, 8, 8), f=1, op=CSQ, note=""):
    return dict(ref_path=path, name=name, input=inp, rounding=rounding, chroma_a=a, chroma_b=b,
                bits=list(bits), factor=f, op=list(op), note=note)


T, F = "TRUNC_SW", "FLOOR_HW"
CH = "APP_OUTPUT/chroma_subsampler_parameterized_tests/"
QZ = "APP_OUTPUT/quantizer_parameterized_tests/"
GOLDENS = [
    # ChromaSubsamplerImageSpec.scala:229 (spec-local SW colour model => TRUNC_SW)
    g(CH + "output_chroma_4-4-4_444_16x16.png", "chroma_444_16", "in16", T, 4, 4),
    g(CH + "output_chroma_4-2-2_422_16x16.png", "chroma_422_16", "in16", T, 2, 2),
    g(CH + "output_chroma_4-2-0_420_16x16.png", "chroma_420_16", "in16", T, 2, 0),
    g(CH + "output_chroma_4-1-1_411_16x16.png", "chroma_411_16", "in16", T, 1, 1),
    # ColorQuantizerImageSpec.scala:209
    g(QZ + "output_quantized_Y8Cb8Cr8_128x128.png", "quant_888_128", "in128", T, 4, 4, (8, 8, 8)),
    g(QZ + "output_quantized_Y6Cb5Cr5_128x128.png", "quant_655_128", "in128", T, 4, 4, (6, 5, 5)),
    g(QZ + "output_quantized_Y3Cb3Cr2_128x128.png", "quant_332_128", "in128", T, 4, 4, (3, 3, 2)),
    g(QZ + "output_quantized_Y8Cb4Cr4_128x128.png", "quant_844_128", "in128", T, 4, 4, (8, 4, 4)),
    g(QZ + "output_quantized_Y4Cb4Cr4_128x128.png", "quant_444_128", "in128", T, 4, 4, (4, 4, 4)),
    g(QZ + "output_quantized_Y1Cb1Cr1_128x128.png", "quant_111_128", "in128", T, 4, 4, (1, 1, 1)),
    # SpatialDownsamplerSpec.scala:227 (ImageProcessor RTL => FLOOR_HW), and older runs of it
    g("APP_OUTPUT/spatial_downsampler_integration_420_sf2.png", "ip_420_sf2_16", "in16", F, 2, 0, f=2),
    g("output_images/out16x16_processed.png", "ip_420_sf2_16_old1", "in16", F, 2, 0, f=2),
    g("output_images/out16x16.png", "ip_420_sf2_16_old2", "in16", F, 2, 0, f=2),
    g("output_images/out8x8.png", "ip_420_sf2_16_old3", "in16", F, 2, 0, f=2),
    # identity round trip of readImage -> writeImage
    g("output_images/out16x16_model_copy.png", "model_copy_16", "in16", "IDENTITY", 4, 4),
    # ImageCompressionApp outputs (ImageCompressorTop RTL => FLOOR_HW)
    g("APP_OUTPUT/in128x128_processed_chroma4-2-2_Y8Cb8Cr8_sf2_order-Pr-Pr-Pr.png",
      "app_422_888_sf2_128", "in128", F, 2, 2, f=2, note="C-before-S (all three such orders match)"),
    g("APP_OUTPUT/in128x128_processed_chromaChromaSubsamplingMode(2=CHROMA_420)_quantQuantizationMode(2=Q_8BIT)_sf1.png",
      "app_420_q8_sf1_128", "in128", F, 2, 0, (3, 3, 2), note="older-API app run; closest relative of BASELINE cfg 2"),
    # older ChromaSubsamplerImageSpec runs
    g("output_images_chroma/output_chroma_444_16x16.png", "old_chroma_444_16", "in16", T, 4, 4),
    g("output_images_chroma/output_chroma_422_16x16.png", "old_chroma_422_16", "in16", T, 2, 2),
    g("output_images_chroma/output_chroma_420_16x16.png", "old_chroma_420_16", "in16", T, 2, 0),
    g("output_images_chroma/output_chroma_444_128x128.png", "old_chroma_444_128", "in128", T, 4, 4),
    g("output_images_chroma/output_chroma_422_128x128.png", "old_chroma_422_128", "in128", T, 2, 2),
    g("output_images_chroma/output_chroma_420_128x128.png", "old_chroma_420_128", "in128", T, 2, 0),
    g("output_images_chroma/output_chroma_444_512x512.png", "old_chroma_444_512", "in512", T, 4, 4),
    g("output_images_chroma/output_chroma_422_512x512.png", "old_chroma_422_512", "in512", T, 2, 2),
    g("output_images_chroma/output_chroma_420_512x512.png", "old_chroma_420_512", "in512", T, 2, 0),
    # older ColorQuantizerImageSpec runs
    g("output_images_quantizer/output_quantized_Q24bit_128x128.png", "old_quant_q24_128", "in128", T, 4, 4, (8, 8, 8)),
    g("output_images_quantizer/output_quantized_Q16bit_128x128.png", "old_quant_q16_128", "in128", T, 4, 4, (6, 5, 5)),
    g("output_images_quantizer/output_quantized_Q8bit_128x128.png", "old_quant_q8_128", "in128", T, 4, 4, (3, 3, 2)),
]


def sha256_file(p):
    return hashlib.sha256(open(p, "rb").read()).hexdigest()


def px_hash(p):
    rgb = np.asarray(Image.open(p).convert("RGB"), dtype=np.uint8)
    return hashlib.sha256(rgb.tobytes()).hexdigest(), rgb.shape


def main():
    os.makedirs(os.path.join(HERE, "inputs"), exist_ok=True)
    os.makedirs(os.path.join(HERE, "outputs"), exist_ok=True)
    manifest = {"inputs": {}, "goldens": []}
    for key, rel in INPUTS.items():
        dst = os.path.join(HERE, "inputs", key + ".png")
        shutil.copyfile(os.path.join(REF, rel), dst)
        h, shape = px_hash(dst)
        manifest["inputs"][key] = dict(ref_path=rel, file="inputs/" + key + ".png",
                                       sha256=sha256_file(dst), px_sha256=h,
                                       height=shape[0], width=shape[1])
    for e in GOLDENS:
        dst = os.path.join(HERE, "outputs", e["name"] + ".png")
        shutil.copyfile(os.path.join(REF, e["ref_path"]), dst)
        h, shape = px_hash(dst)
        e = dict(e)
        e.update(file="outputs/" + e["name"] + ".png", sha256=sha256_file(dst), px_sha256=h,
                 height=shape[0], width=shape[1])
        manifest["goldens"].append(e)
    with open(os.path.join(HERE, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1)
    print(f"wrote {len(manifest['goldens'])} goldens, {len(manifest['inputs'])} inputs")


if __name__ == "__main__":
    main()
