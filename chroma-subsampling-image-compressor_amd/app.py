"""Drop-in for `sbt "Test / runMain jpeg.ImageCompressionApp ..."`:

    python -m csic_amd.app --input test_images/in128x128.png --a 2 --b 0 --sf 2 --op1 chroma --op2 spatial --op3 color

ImageCompressionApp <- object ImageCompressionApp, src/test/scala/jpeg/ImageCompressorTopApp.scala:18-216
Same flags (space-separated `--key value` pairs, :149-151), same defaults (:164-173; note sf = 8 and the
order spatial, color, chroma), same banner, same output naming including the literal `order-Pr-Pr-Pr`
tag (:188; ChiselEnum.toString quirk).  The simulated-RTL run is replaced by one fused
HIP launch.
"""
from __future__ import annotations

import os
import sys
from typing import Dict, List

import numpy as np

from .compressor import ImageCompressorTop
from .model import Image, ImageProcessorModel
from .params import ProcessingStep


class ImageCompressionApp:
    @staticmethod
    def processImage(inputImagePath: str, outputImagePath: str,
                     chromaParamA: int, chromaParamB: int,
                     yTargetBits: int, cbTargetBits: int, crTargetBits: int,
                     spatialFactorToUse: int,
                     op1: ProcessingStep, op2: ProcessingStep, op3: ProcessingStep, *, device: int = 0,
                     emulateCollectorBudget: bool = False) -> None:
        """ImageCompressorTopApp.scala:23-145.

        emulateCollectorBudget: the reference's collector loop gives up after (W/f)*(H/f)*40 + 10000 simulated cycles (:110)
        and the pixels it did not get keep the image's magenta fill (:133-141).  ImageCompressorTop accepts one pixel every
        two cycles (three non-pipe Queue(1)s), so that budget is too small for sf = 8 on anything larger than ~85x85 --
        including the app's own defaults (in128x128.png, sf = 8: 160 of 256 pixels).  Off (default), every pixel is
        written; on, the cycle-level model (stream.ImageCompressorTop) is asked how many pixels the collector would have
        got and the rest stays magenta, which is what a run of the reference app is predicted to produce.  The pixel
        VALUES come from the GPU either way; the model contributes a count."""
        inputImage = ImageProcessorModel.readImage(inputImagePath)
        W, H = inputImage.width, inputImage.height
        f = spatialFactorToUse
        spatial_in = ProcessingStep.SpatialSampling in (op1, op2, op3)
        top = ImageCompressorTop(W, H, chromaParamA, chromaParamB, yTargetBits, cbTargetBits, crTargetBits,
                                 f, op1, op2, op3, device=device)
        finalW = W // f if spatial_in else W                      # :44-45
        finalH = H // f if spatial_in else H
        if spatial_in and (W % f != 0 or H % f != 0):
            print(f"[WARN] Image dimensions ({W}x{H}) are not perfectly divisible by spatialFactor ({f}). "
                  "SpatialDownsampler might truncate.")                # :47-49
        out = top.process(inputImage.argb)                           # (ceil(H/f), ceil(W/f)) stream, row-major
        top.close()
        # The harness collects the first finalW*finalH pixels of the OUTPUT STREAM and lays them out
        # finalW per row (:108-124, :133-142); identical to `out` when the dimensions divide.
        stream = out.reshape(-1)[: finalW * finalH]
        if emulateCollectorBudget:
            from .stream import ImageCompressorTop as CycleModel
            with CycleModel(W, H, chromaParamA, chromaParamB, yTargetBits, cbTargetBits, crTargetBits, f, op1, op2, op3) as dut:
                got, _ = dut.run(inputImage.argb, max_out=finalW * finalH, max_cycles=dut.collection_budget())
            if got.size < finalW * finalH:
                print(f"[WARN] Output collection timed out. Collected {got.size} out of {finalW * finalH} pixels.")    # :126-128
            stream = stream[: got.size]
        if stream.size < finalW * finalH:
            pad = np.full(finalW * finalH - stream.size, 0xFFFF00FF, dtype=np.uint32)   # AwtColor.MAGENTA fill, :133
            stream = np.concatenate([stream, pad])
        ImageProcessorModel.writeImage(Image(stream.reshape(finalH, finalW)), outputImagePath)


    @staticmethod
    def processImages(inputImagePaths, outputImagePaths, chromaParamA: int, chromaParamB: int,
                      yTargetBits: int, cbTargetBits: int, crTargetBits: int, spatialFactorToUse: int,
                      op1: ProcessingStep, op2: ProcessingStep, op3: ProcessingStep, *, device: int = 0,
                      depth: int = 3, decodeThreads: int = None, encodeThreads: int = None, compression: int = 6):
        """Many same-sized images through one plan.  Output files are what processImage would write for each input.

        Default: csic_process_png_files -- pools of decoder and encoder threads inside the library (no GIL) around
        pinned frame slots; a decoder inflates a PNG straight into a slot and launches the kernel on the slot's stream,
        an encoder waits for the slot and writes the output file.  decodeThreads / encodeThreads = None: the library's
        choice; returns the call's csic_files_stats as a dict.
        decodeThreads=0: the serial reference flow of round 2 -- this thread decodes into a FramePipeline slot, submits,
        collects and encodes, one image after the other (only H2D / kernel / D2H of neighbouring images overlap); kept
        as the baseline the pooled path is byte-compared against (tests/test_gpu_parity.py)."""
        import ctypes as C
        from . import _native as N
        inputImagePaths, outputImagePaths = list(inputImagePaths), list(outputImagePaths)
        n = len(inputImagePaths)
        if n == 0:
            raise N.IllegalArgumentException(N.EINVAL_SIZE, "requirement failed: no input image")
        if n != len(outputImagePaths):
            raise N.IllegalArgumentException(N.EINVAL_SIZE, "requirement failed: need as many output as input paths")
        W, H = ImageProcessorModel.imageSize(inputImagePaths[0])
        f = spatialFactorToUse
        finalW, finalH = W // f, H // f
        if finalW == 0 or finalH == 0:
            # the reference's collector would build a 0-pixel image (ImageCompressorTopApp.scala:44-45); there is nothing to write,
            # and a final size of 0 means "the plan's output size" to the C ABI -- refuse instead of writing something else
            raise N.IllegalArgumentException(N.EINVAL_SIZE, f"requirement failed: a {W}x{H} image at factor {f} leaves no whole output pixel")
        top = ImageCompressorTop(W, H, chromaParamA, chromaParamB, yTargetBits, cbTargetBits, crTargetBits,
                                 f, op1, op2, op3, device=device)
        try:
            for path in outputImagePaths:
                os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)     # outputFile.getParentFile().mkdirs(), ImageProcessorModel.scala:20
            if decodeThreads == 0:
                ImageCompressionApp._processImagesSerial(top, inputImagePaths, outputImagePaths, finalW, finalH, depth, compression)
                return None
            ins = (C.c_char_p * n)(*[os.fsencode(p) for p in inputImagePaths])
            outs = (C.c_char_p * n)(*[os.fsencode(p) for p in outputImagePaths])
            st = N.CsicFilesStats()
            N.check(N.lib().csic_process_png_files(top.plan()._h, ins, outs, n, decodeThreads or 0, encodeThreads or 0, compression,
                                                   finalW, finalH, C.byref(st)))
            return {k: getattr(st, k) for k, _ in N.CsicFilesStats._fields_}
        finally:
            top.close()

    @staticmethod
    def _processImagesSerial(top, inputImagePaths, outputImagePaths, finalW, finalH, depth, compression):
        from .pipeline import FramePipeline
        todo = list(outputImagePaths)

        def write(stream_frame, path):
            stream = stream_frame.reshape(-1)[: finalW * finalH]
            if stream.size < finalW * finalH:
                stream = np.concatenate([stream, np.full(finalW * finalH - stream.size, 0xFFFF00FF, dtype=np.uint32)])
            ImageProcessorModel.writeImage(Image(stream.reshape(finalH, finalW).copy()), path, compression=compression)

        with FramePipeline(top.plan(), depth) as pipe:
            done = 0
            for path in inputImagePaths:
                if pipe.pending == depth:
                    write(pipe.collect()[1], todo[done]); done += 1
                # decode straight into the pinned staging buffer (a size mismatch raises IllegalArgumentException)
                ImageProcessorModel.readImageInto(path, pipe.acquire_input())
                pipe.submit()
            while pipe.pending:
                write(pipe.collect()[1], todo[done]); done += 1


def _order_tag(step: ProcessingStep) -> str:
    # `${op.toString.split('.').last.take(2)}` on a ChiselEnum value "ProcessingStep(1=SpatialSampling)"
    # yields "Pr" for every step (ImageCompressorTopApp.scala:188).
    return f"ProcessingStep({int(step)}={step.name})".split(".")[-1][:2]


def main(argv: List[str] = None) -> int:
    args = list(sys.argv[1:] if argv is None else argv)
    argsMap: Dict[str, str] = {}
    for i in range(0, len(args) - 1, 2):                              # args.sliding(2, 2), :149-151
        if args[i].startswith("--"):
            argsMap[args[i]] = args[i + 1]
    inputPath = argsMap.get("--input", "test_images/in128x128.png")
    a = int(argsMap.get("--a", "4"))
    b = int(argsMap.get("--b", "4"))
    yq = int(argsMap.get("--yq", "8"))
    cbq = int(argsMap.get("--cbq", "8"))
    crq = int(argsMap.get("--crq", "8"))
    sf = int(argsMap.get("--sf", "8"))
    op1 = ProcessingStep.parse(argsMap.get("--op1", "spatial"))
    op2 = ProcessingStep.parse(argsMap.get("--op2", "color"))
    op3 = ProcessingStep.parse(argsMap.get("--op3", "chroma"))
    imageName = os.path.basename(inputPath).split(".")[0]

    print("----------------------------------------------------")
    print("Image Compressor Application Parameters:")
    print("----------------------------------------------------")
    print(f"Input Image: {inputPath}")
    print(f"Selected Chroma Subsampling (J:a:b): 4:{a}:{b}")
    print(f"Selected Quantization Bits (Y/Cb/Cr): {yq}/{cbq}/{crq}")
    print(f"Selected Spatial Downsampling Factor: {sf}")
    print(f"Selected Pipeline Order: {op1.name} -> {op2.name} -> {op3.name}")
    print("----------------------------------------------------")

    outDir = argsMap.get("--outdir", "APP_OUTPUT")
    order = f"order-{_order_tag(op1)}-{_order_tag(op2)}-{_order_tag(op3)}"
    suffix = f"chroma4-{a}-{b}_Y{yq}Cb{cbq}Cr{crq}_sf{sf}_{order}"
    outputPath = f"{outDir}/{imageName}_processed_{suffix}.png"
    os.makedirs(outDir, exist_ok=True)
    if not os.path.exists(inputPath):
        print(f"[ERROR] Input image not found: {inputPath}")          # :197-199 (not an exception)
        return 0
    # not a key of the reference's CLI: `--collector-budget emulate` reproduces its collector's cycle budget (see processImage)
    emulate = argsMap.get("--collector-budget", "off") == "emulate"
    ImageCompressionApp.processImage(inputPath, outputPath, a, b, yq, cbq, crq, sf, op1, op2, op3, emulateCollectorBudget=emulate)
    print(f"Image processing complete. Output saved to: {outputPath}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
