/*
 * SoftwareModelBench.java -- the "Scala/JVM CPU path" baseline as ONE source file, so that any host with a JDK >= 11 can time
 * it without scalac or a build step (JEP 330: `java SoftwareModelBench.java ...`).  It is the int[] algorithm of
 * jvm/scala/jpeg/SoftwareModel.scala -- the reference's stage state machines run in op order on a stream of (y, cb, cr) ints --
 * written from SURVEY.md App. A:
 *   forward    App. A.1  /root/reference/src/main/scala/jpeg/ReferenceModel.scala:8-19 (floor, the RTL's result);
 *                        RGB2YCbCr.scala:95-121 (trunc, YCbCrUtils)
 *   chroma     App. A.3  ChromaSubsampler.scala:37-65    sample-and-hold, counters wrap at the FULL width / height
 *   spatial    App. A.4  SpatialDownsampler.scala:17-55  keep (row % f == 0 && col % f == 0)
 *   quantiser  App. A.2  ColorQuantizer.scala:29-44      v & (0xFF << (8 - bits))
 *   inverse    App. A.5  YCbCr2RGB.scala:17-26
 * bench.py runs it when `java` is on the PATH and reports the result as cpu_baseline.jvm; the `checksum` it prints (the
 * order-sensitive sum csic_checksum_device computes on the GPU) lets bench.py check the JVM's output frame against the oracle's.
 * Not compiled or run in this repository's image (no JDK); tests/test_jvm_sources.py checks coefficients and CLI at source level.
 *
 *   java SoftwareModelBench.java [width height a b yBits cbBits crBits factor op1,op2,op3 seconds [trunc]]
 *   defaults: 8192 8192 2 0 8 8 8 2 3,1,2 10      (op ordinals: 1 = SpatialSampling, 2 = ColorQuantization, 3 = ChromaSubsampling)
 */
public final class SoftwareModelBench {
    /** rows Y, Cb, Cr of the forward matrix, times 256 (SURVEY.md App. A.1) */
    static final int[] FORWARD = {77, 150, 29, -43, -85, 128, 128, -107, -21};

    static int sat8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

    static int fmix32(int x) {
        x ^= x >>> 16; x *= 0x85ebca6b; x ^= x >>> 13; x *= 0xc2b2ae35; x ^= x >>> 16;
        return x;
    }

    final int width, height, factor, h, v, maskY, maskCb, maskCr, outWidth, outHeight;
    final int[] order;
    final boolean floorRounding;

    SoftwareModelBench(int width, int height, int a, int b, int yBits, int cbBits, int crBits, int factor, int[] order, boolean floorRounding) {
        // the reference's require()s (ImageProcessor.scala:22-28, ColorQuantizer.scala:12-15, ImageCompressorTop.scala:27-31)
        if (width <= 0 || height <= 0) throw new IllegalArgumentException("requirement failed: width and height must be positive");
        if (factor != 1 && factor != 2 && factor != 4 && factor != 8) throw new IllegalArgumentException("requirement failed: factor must be 1, 2, 4, or 8");
        if (a != 4 && a != 2 && a != 1) throw new IllegalArgumentException("requirement failed: chroma a must be 4, 2, or 1");
        if (b != a && b != 0) throw new IllegalArgumentException("requirement failed: chroma b must be a or 0");
        for (int bits : new int[] {yBits, cbBits, crBits})
            if (bits < 1 || bits > 8) throw new IllegalArgumentException("requirement failed: target bits must be between 1 and 8");
        if (order.length != 3 || (1 << order[0] | 1 << order[1] | 1 << order[2]) != 0b1110)
            throw new IllegalArgumentException("requirement failed: op1, op2, op3 must be a permutation of 1, 2, 3");
        this.width = width; this.height = height; this.factor = factor; this.order = order; this.floorRounding = floorRounding;
        this.h = 4 / a;
        this.v = b == 0 ? 2 : 1;
        this.maskY = (0xFF << (8 - yBits)) & 0xFF;
        this.maskCb = (0xFF << (8 - cbBits)) & 0xFF;
        this.maskCr = (0xFF << (8 - crBits)) & 0xFF;
        this.outWidth = (width + factor - 1) / factor;
        this.outHeight = (height + factor - 1) / factor;
    }

    /** ARGB frame (width * height ints) -> reconstructed ARGB frame (outWidth * outHeight ints), alpha = 255. */
    int[] process(int[] argb) {
        if (argb.length != width * height) throw new IllegalArgumentException("requirement failed: expected " + width * height + " pixels");
        final int[] out = new int[outWidth * outHeight];
        // stage state, as in the RTL: chroma latch + its two counters, the decimator's two counters
        int lastCb = 0, lastCr = 0, cPix = 0, cLine = 0, sCol = 0, sRow = 0, n = 0;
        final int o0 = order[0], o1 = order[1], o2 = order[2];
        for (int i = 0; i < argb.length; ++i) {
            final int px = argb[i];
            final int r = (px >> 16) & 255, g = (px >> 8) & 255, b = px & 255;
            final int ty = FORWARD[0] * r + FORWARD[1] * g + FORWARD[2] * b + 128;
            final int tb = FORWARD[3] * r + FORWARD[4] * g + FORWARD[5] * b + 128;
            final int tr = FORWARD[6] * r + FORWARD[7] * g + FORWARD[8] * b + 128;
            int y, cb, cr;
            if (floorRounding) { y = sat8(ty >> 8); cb = sat8((tb >> 8) + 128); cr = sat8((tr >> 8) + 128); }
            else { y = sat8(ty / 256); cb = sat8(tb / 256 + 128); cr = sat8(tr / 256 + 128); }
            boolean alive = true;
            for (int k = 0; k < 3 && alive; ++k) {
                final int op = k == 0 ? o0 : (k == 1 ? o1 : o2);
                if (op == 1) {                                            // SpatialSampling
                    alive = (sCol % factor == 0) && (sRow % factor == 0);
                    if (++sCol == width) { sCol = 0; if (++sRow == height) sRow = 0; }
                } else if (op == 2) {                                     // ColorQuantization
                    y &= maskY; cb &= maskCb; cr &= maskCr;
                } else {                                                  // ChromaSubsampling
                    if (cPix % h == 0 && cLine % v == 0) { lastCb = cb; lastCr = cr; } else { cb = lastCb; cr = lastCr; }
                    if (++cPix == width) { cPix = 0; if (++cLine == height) cLine = 0; }
                }
            }
            if (alive) {
                final int d = cb - 128, e = cr - 128, luma = 298 * y + 128;
                out[n++] = 0xFF000000 | (sat8((luma + 409 * e) >> 8) << 16) | (sat8((luma - 100 * d - 208 * e) >> 8) << 8) | sat8((luma + 516 * d) >> 8);
            }
        }
        return out;
    }

    /** what csic_checksum_device computes: sum over i of fmix32(px[i] + 0x9E3779B9 * i), 64-bit */
    static long checksum(int[] px) {
        long acc = 0;
        for (int i = 0; i < px.length; ++i) acc += fmix32(px[i] + 0x9E3779B9 * i) & 0xFFFFFFFFL;
        return acc;
    }

    public static void main(String[] args) {
        final int w = args.length > 0 ? Integer.parseInt(args[0]) : 8192;
        final int hgt = args.length > 1 ? Integer.parseInt(args[1]) : 8192;
        final int a = args.length > 2 ? Integer.parseInt(args[2]) : 2;
        final int b = args.length > 3 ? Integer.parseInt(args[3]) : 0;
        final int yq = args.length > 4 ? Integer.parseInt(args[4]) : 8;
        final int cbq = args.length > 5 ? Integer.parseInt(args[5]) : 8;
        final int crq = args.length > 6 ? Integer.parseInt(args[6]) : 8;
        final int sf = args.length > 7 ? Integer.parseInt(args[7]) : 2;
        final String[] ops = (args.length > 8 ? args[8] : "3,1,2").split(",");
        final double budget = args.length > 9 ? Double.parseDouble(args[9]) : 10.0;
        final boolean floor = !(args.length > 10 && args[10].equals("trunc"));
        final int[] order = {Integer.parseInt(ops[0]), Integer.parseInt(ops[1]), Integer.parseInt(ops[2])};
        // the counter-based generator of SURVEY.md 8(d): frame 0, seed 20250629
        final int[] frame = new int[w * hgt];
        for (int i = 0; i < frame.length; ++i) frame[i] = 0xFF000000 | (fmix32(i + 20250629 * 0x9E3779B9) & 0xFFFFFF);
        final SoftwareModelBench model = new SoftwareModelBench(w, hgt, a, b, yq, cbq, crq, sf, order, floor);
        int[] out = null;
        for (int i = 0; i < 3; ++i) out = model.process(frame);          // warm-up: let C2 compile the loop
        final long sum = checksum(out);
        int frames = 0;
        final long t0 = System.nanoTime();
        do { out = model.process(frame); ++frames; } while ((System.nanoTime() - t0) * 1e-9 < budget);
        final double secs = (System.nanoTime() - t0) * 1e-9;
        System.out.println(String.format(java.util.Locale.ROOT,
            "{\"value\": %.2f, \"unit\": \"Mpixels/s\", \"cores\": 1, \"sample\": \"%d full %dx%d frames through SoftwareModelBench.process "
            + "(one thread, after 3 warm-up frames) in %.1f s on Java %s\", \"out_pixels\": %d, \"checksum\": \"0x%016x\"}",
            frames * (double) w * hgt / secs / 1e6, frames, w, hgt, secs, System.getProperty("java.version"), out.length, sum));
    }
}
