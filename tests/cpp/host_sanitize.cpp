// host_sanitize.cpp -- ASan/UBSan driver for the host-only half of the library (no HIP): parameter
// validation, geometry, stripes, and the PNG codec under a mutation fuzzer that re-signs chunk CRCs so
// that corrupt payloads reach the inflate / unfilter / unpack code.  Built and run by
// tests/test_sanitizers.py with  g++ -fsanitize=address,undefined  (CPU only; GPU ASan is unavailable).
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "csic.h"
#include "csic_internal.h"

static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }

static std::vector<unsigned char> slurp(const char *p)
{
    std::vector<unsigned char> b;
    FILE *f = std::fopen(p, "rb");
    if (!f) return b;
    std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    b.resize(n); if (std::fread(b.data(), 1, n, f) != (size_t)n) b.clear();
    std::fclose(f);
    return b;
}

static void resign(std::vector<unsigned char> &png)      // recompute every chunk CRC
{
    size_t pos = 8;
    while (pos + 12 <= png.size()) {
        uint32_t len = (png[pos] << 24) | (png[pos + 1] << 16) | (png[pos + 2] << 8) | png[pos + 3];
        if (pos + 12 + (size_t)len > png.size()) break;
        uint32_t crc = crc32(crc32(0, Z_NULL, 0), &png[pos + 4], 4 + len);
        png[pos + 8 + len] = crc >> 24; png[pos + 9 + len] = crc >> 16; png[pos + 10 + len] = crc >> 8; png[pos + 11 + len] = crc;
        pos += 12 + len;
    }
}

// csic_inflate.cpp against zlib itself: streams of every block type, level, strategy, window and block size must decode
// to the same bytes; mutated streams must be accepted or rejected exactly as uncompress() does, byte-identical where
// accepted; crc32_update / adler32_update must agree with zlib's on every length, offset and chaining.
static int check_inflate()
{
    long streams = 0, mutated = 0, accepted = 0;
    std::vector<unsigned char> src, comp, got, ref;
    for (int it = 0; it < 700; ++it) {
        const unsigned kind = rnd() % 8;
        size_t n = rnd() % 8 == 0 ? rnd() % 40 : rnd() % 4 == 0 ? 100000 + rnd() % 200000 : rnd() % 20000;
        src.resize(n);
        for (size_t i = 0; i < n; ++i) {
            switch (kind) {
            case 0: src[i] = (unsigned char)rnd(); break;                                             // incompressible
            case 1: src[i] = (unsigned char)("the quick brown fox "[rnd() % 20]); break;
            case 2: src[i] = (unsigned char)(i / 977); break;                                         // long runs
            case 3: src[i] = i >= 3 && rnd() % 16 ? src[i - 3] : (unsigned char)rnd(); break;         // distance 3, PNG-like
            case 4: src[i] = (unsigned char)((rnd() % 7) - 3); break;                                 // filter residuals
            case 5: src[i] = 0; break;
            case 6: src[i] = i >= 4 && rnd() % 64 ? (unsigned char)(src[i - 4] + (rnd() % 3) - 1) : (unsigned char)rnd(); break;
            default: { const size_t d = 1 + (it % 40000); src[i] = i >= d && rnd() % 200 ? src[i - d] : (unsigned char)rnd(); break; }
            }
        }
        static const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        const int level = (int)(rnd() % 10), wbits = 9 + (int)(rnd() % 7), mem = 1 + (int)(rnd() % 9);
        if (deflateInit2(&zs, level, Z_DEFLATED, wbits, mem, strategies[rnd() % 5]) != Z_OK) { std::printf("deflateInit2 failed\n"); return 1; }
        comp.resize(deflateBound(&zs, (uLong)n) + 64);
        zs.next_in = src.data(); zs.avail_in = (uInt)n; zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
        if (rnd() % 3 == 0 && n > 10) {                                                               // a flush point in the middle: empty stored block
            zs.avail_in = (uInt)(n / 2);
            deflate(&zs, rnd() % 2 ? Z_SYNC_FLUSH : Z_FULL_FLUSH);
            zs.avail_in = (uInt)(n - n / 2);
        }
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { std::printf("deflate failed\n"); return 1; }
        comp.resize(zs.total_out);
        deflateEnd(&zs);
        got.assign(n + 1, 0xA5);
        int st = csic::zlib_decode_exact(comp.data(), comp.size(), got.data(), n);
        if (st != 0 || std::memcmp(got.data(), src.data(), n) != 0 || got[n] != 0xA5) {
            std::printf("inflate mismatch: status %d, n %zu level %d wbits %d mem %d kind %u\n", st, n, level, wbits, mem, kind);
            return 1;
        }
        if (csic::zlib_decode_exact(comp.data(), comp.size(), got.data(), n + 1) != 4) { std::printf("a short stream was accepted\n"); return 1; }
        if (n && csic::zlib_decode_exact(comp.data(), comp.size(), got.data(), n - 1) != 3) { std::printf("a long stream was accepted\n"); return 1; }
        for (size_t cut = comp.size() > 12 ? comp.size() - 12 : 0; cut < comp.size(); ++cut)          // truncations near the end
            if (csic::zlib_decode_exact(comp.data(), cut, got.data(), n) == 0) { std::printf("a truncated stream was accepted\n"); return 1; }
        ++streams;
        for (int m = 0; m < 12; ++m) {
            std::vector<unsigned char> mu = comp;
            const int nmut = 1 + rnd() % 3;
            for (int k = 0; k < nmut; ++k) {
                const size_t pos = rnd() % mu.size();
                if (rnd() % 2) mu[pos] ^= 1u << (rnd() % 8); else mu[pos] = (unsigned char)rnd();
            }
            if (rnd() % 8 == 0) mu.resize(1 + rnd() % mu.size());
            ref.assign(n, 0);
            uLongf rl = (uLongf)n;
            unsigned char dummy = 0;
            const int zr = uncompress(n ? ref.data() : &dummy, &rl, mu.data(), (uLong)mu.size());
            const bool z_ok = zr == Z_OK && rl == n;
            got.assign(n + 1, 0xA5);
            st = csic::zlib_decode_exact(mu.data(), mu.size(), got.data(), n);
            if (got[n] != 0xA5) { std::printf("wrote past the output\n"); return 1; }
            if ((st == 0) != z_ok) { std::printf("mutated stream: zlib %d (len %lu of %zu), ours %d\n", zr, (unsigned long)rl, n, st); return 1; }
            if (z_ok && std::memcmp(got.data(), ref.data(), n) != 0) { std::printf("mutated stream decodes differently\n"); return 1; }
            accepted += z_ok;
            ++mutated;
        }
    }
    std::vector<unsigned char> buf(70000);
    for (auto &b : buf) b = (unsigned char)rnd();
    for (int it = 0; it < 3000; ++it) {
        const size_t off = rnd() % 64, n = it < 300 ? (size_t)it : rnd() % (buf.size() - off), cut = n ? rnd() % (n + 1) : 0;
        const unsigned char *p = buf.data() + off;
        const uint32_t c = csic::crc32_update(csic::crc32_update(0, p, cut), p + cut, n - cut);
        const uint32_t a = csic::adler32_update(csic::adler32_update(1, p, cut), p + cut, n - cut);
        if (c != (uint32_t)crc32(crc32(0, Z_NULL, 0), p, (uInt)n) || a != (uint32_t)adler32(adler32(0, Z_NULL, 0), p, (uInt)n)) {
            std::printf("checksum mismatch at off %zu n %zu\n", off, n);
            return 1;
        }
    }
    std::memset(buf.data(), 0xFF, buf.size());                                                        // the sums' worst case
    if (csic::adler32_update(0xFFF0FFF0u % 65521u | ((0xFFF0u % 65521u) << 16), buf.data(), buf.size()) !=
        (uint32_t)adler32(0xFFF0FFF0u % 65521u | ((0xFFF0u % 65521u) << 16), buf.data(), (uInt)buf.size())) { std::printf("adler overflow\n"); return 1; }
    std::printf("inflate ok: %ld streams, %ld mutated (%ld still valid)\n", streams, mutated, accepted);
    return 0;
}

// magic_div must reproduce the hardware divide for every dividend below 2^31
static int check_magic_div()
{
    const uint32_t edges[] = {1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 125, 250, 255, 256, 257, 500, 960, 1000, 1023, 1024, 1025,
                              1920, 3840, 4095, 4096, 4097, 7680, 8191, 8192, 8193, 65535, 65536, 65537, 1u << 20, (1u << 20) + 1, 16777215,
                              16777216, 16777217, (1u << 30) - 1, 1u << 30, (1u << 30) + 1, 2147483646u, 2147483647u};
    long n_checked = 0;
    auto one = [&](uint32_t d) -> bool {
        uint32_t m = 0, k = 0;
        csic::magic_div(d, &m, &k);
        if (k < 31 || k > 62) return false;
        auto ok = [&](uint32_t n) { ++n_checked; return (uint32_t)(((uint64_t)n * m) >> k) == n / d; };
        for (uint32_t n : edges) if (!ok(n)) return false;
        for (uint64_t q = 0; q < 6; ++q)                                  // multiples of d and their neighbours
            for (int64_t off = -2; off <= 2; ++off) {
                const int64_t n = (int64_t)(q * 357913941ull % (2147483648ull / d + 1)) * d + off;
                if (n >= 0 && n < 2147483648ll && !ok((uint32_t)n)) return false;
            }
        const int64_t top = (int64_t)(2147483647u / d) * d;              // the largest multiple below 2^31
        for (int64_t off = -2; off <= 2; ++off)
            if (top + off >= 0 && top + off < 2147483648ll && !ok((uint32_t)(top + off))) return false;
        for (int i = 0; i < 64; ++i) if (!ok(rnd() & 0x7FFFFFFFu)) return false;
        return true;
    };
    for (uint32_t d : edges) if (!one(d)) { std::printf("magic_div wrong for d = %u\n", d); return 1; }
    for (int i = 0; i < 200000; ++i) {
        uint32_t d = rnd() & 0x7FFFFFFFu;
        if (i & 1) d >>= (rnd() % 31);                                   // small divisors matter most (frame widths)
        if (d == 0) d = 1;
        if (!one(d)) { std::printf("magic_div wrong for d = %u\n", d); return 1; }
    }
    std::printf("magic_div: %ld quotients checked\n", n_checked);
    return 0;
}

// the cycle-level stream model: every kind, random shapes and orders, random producer gaps and back-pressure; the collected stream
// must not depend on the handshakes, and eval / step must agree with run
static int check_stream_model()
{
    long runs = 0;
    for (int it = 0; it < 300; ++it) {
        csic_params p; csic_params_default(&p, 1 + rnd() % 33, 1 + rnd() % 17);
        const int ab[5][2] = {{4, 4}, {2, 2}, {2, 0}, {1, 1}, {1, 0}};
        const int perms[6][3] = {{1, 2, 3}, {1, 3, 2}, {2, 1, 3}, {2, 3, 1}, {3, 1, 2}, {3, 2, 1}};
        const int k = rnd() % 5;
        p.chroma_a = ab[k][0]; p.chroma_b = ab[k][1];
        p.factor = 1 << (rnd() % 4);
        p.y_bits = 1 + rnd() % 8; p.cb_bits = 1 + rnd() % 8; p.cr_bits = 1 + rnd() % 8;
        std::memcpy(p.op, perms[rnd() % 6], sizeof p.op);
        p.out_format = rnd() % 2;
        const int kind = rnd() % 6;
        if (kind >= CSIC_STREAM_CHROMA) p.in_format = CSIC_FMT_YCBCR888X;
        csic_stream *s = nullptr;
        if (csic_stream_create(&p, kind, &s) != 0) { std::printf("stream create failed: %s\n", csic_last_error()); return 1; }
        const size_t n = (size_t)p.width * p.height;
        std::vector<uint32_t> in(n), a(n + 1), b(n + 1);
        for (auto &v : in) v = rnd();
        size_t na = 0, nb = 0; int64_t ca = 0, cb = 0;
        if (csic_stream_run(s, in.data(), n, a.data(), n, -1, nullptr, 0, nullptr, 0, &na, &ca) != 0) return 1;
        uint8_t pv[7], pr[5];
        for (auto &v : pv) v = rnd() & 1;
        for (auto &v : pr) v = rnd() & 1;
        pv[rnd() % 7] = 1; pr[rnd() % 5] = 1;
        csic_stream_reset(s);
        if (csic_stream_run(s, in.data(), n, b.data(), n, -1, pv, 7, pr, 5, &nb, &cb) != 0) return 1;
        if (na != nb || std::memcmp(a.data(), b.data(), na * 4) != 0 || cb < ca) { std::printf("stream depends on the handshakes\n"); return 1; }
        // the same through eval + step
        csic_stream_reset(s);
        size_t fed = 0, got = 0;
        for (int64_t c = 0; c < cb + 8 && got < na; ++c) {
            csic_stream_in pin{fed < n && pv[c % 7] ? 1 : 0, fed < n ? in[fed] : 0u, pr[c % 5] ? 1 : 0, 0, (int32_t)(rnd() & 1)};
            csic_stream_out o1, o2;
            if (csic_stream_eval(s, &pin, &o1) != 0 || csic_stream_step(s, &pin, &o2) != 0) return 1;
            if (o1.in_ready != o2.in_ready || o1.out_valid != o2.out_valid || (o1.out_valid && o1.out_bits != o2.out_bits)) { std::printf("eval != step\n"); return 1; }
            if (o1.out_valid && pin.out_ready) { if (o1.out_bits != a[got]) { std::printf("step stream differs at %zu\n", got); return 1; } ++got; }
            if (pin.in_valid && o1.in_ready) ++fed;
        }
        if (got != na || csic_stream_cycles(s) <= 0 || csic_stream_depth(s) < 1) { std::printf("step stream incomplete\n"); return 1; }
        uint8_t zero = 0;
        if (csic_stream_run(s, in.data(), n, b.data(), n, -1, nullptr, 0, &zero, 1, &nb, &cb) == 0) { std::printf("all-zero ready pattern accepted\n"); return 1; }
        csic_stream_destroy(s);
        ++runs;
    }
    csic_params bad; csic_params_default(&bad, 8, 8);
    bad.rounding = CSIC_ROUND_TRUNC_SW;
    csic_stream *s = nullptr;
    if (csic_stream_create(&bad, CSIC_STREAM_TOP, &s) != CSIC_EINVAL_ROUNDING || s) { std::printf("TRUNC_SW accepted by the RTL model\n"); return 1; }
    if (csic_stream_create(&bad, 9, &s) == 0 || csic_stream_create(nullptr, 0, &s) == 0 || csic_stream_step(nullptr, nullptr, nullptr) == 0) return 1;
    std::printf("stream model: %ld random configurations\n", runs);
    return 0;
}

int main(int argc, char **argv)
{
    if (check_magic_div()) return 1;
    if (check_stream_model()) return 1;
    // ---- validation / geometry / stripes over a grid of good and bad parameters
    long checked = 0;
    for (int W : {-1, 0, 1, 5, 16, 8192, 65536}) for (int H : {0, 1, 3, 4096, 40000})
    for (int f : {0, 1, 2, 3, 4, 8, 16}) for (int a : {0, 1, 2, 3, 4}) for (int b : {0, 1, 2, 4})
    for (int o = 0; o < 4; ++o) {
        csic_params p; csic_params_default(&p, W, H);
        p.factor = f; p.chroma_a = a; p.chroma_b = b;
        const int ops[4][3] = {{3, 1, 2}, {1, 2, 3}, {1, 1, 2}, {0, 7, -3}};
        for (int k = 0; k < 3; ++k) p.op[k] = ops[o][k];
        p.sampling = o & 1;
        int32_t wo, ho, r0, nr, o0, on; int64_t bytes;
        int st = csic_validate(&p);
        int st2 = csic_out_dims(&p, &wo, &ho), st3 = csic_algorithmic_bytes(&p, &bytes);
        if ((st == 0) != (st2 == 0) || (st == 0) != (st3 == 0)) { std::printf("inconsistent validation\n"); return 1; }
        for (int n : {1, 2, 7, 8}) for (int r = 0; r < n; ++r) {
            (void)csic_stripe_rows(&p, n, r, &r0, &nr, &o0, &on);
            int32_t splits[9], pr0, pn, halo, tail;
            splits[0] = 0;
            for (int k = 1; k < n; ++k) splits[k] = (H > 0) ? (int32_t)(rnd() % (unsigned)(H + 1)) : 0;
            splits[n] = H;
            for (int i = 1; i < n; ++i) for (int j = i + 1; j < n; ++j) if (splits[j] < splits[i]) { int32_t t = splits[i]; splits[i] = splits[j]; splits[j] = t; }
            (void)csic_stripe_halo(&p, n, r, splits, &pr0, &pn, &halo, &tail, &o0, &on);
        }
        (void)csic_strerror(st); (void)csic_last_error();
        ++checked;
    }
    if (check_inflate() != 0) return 1;
    // ---- PNG codec: round trip, then mutation fuzzing of every fixture given on the command line
    std::vector<uint32_t> img(37 * 23);
    for (auto &v : img) v = rnd();
    const std::string tmp = std::string(argv[1]) + "/rt.png";
    if (csic_png_write_argb(tmp.c_str(), img.data(), 37, 23, 6) != 0) { std::printf("write failed\n"); return 1; }
    std::vector<uint32_t> back(37 * 23);
    if (csic_png_read_argb(tmp.c_str(), back.data(), back.size()) != 0) { std::printf("read failed\n"); return 1; }
    for (size_t i = 0; i < img.size(); ++i) if ((img[i] | 0xFF000000u) != back[i]) { std::printf("roundtrip mismatch\n"); return 1; }
    {   // an image of two deflate pieces (1.26 MB filtered), written on three threads
        std::vector<uint32_t> big(700 * 600), bigback(700 * 600);
        for (size_t i = 0; i < big.size(); ++i) big[i] = (i / 700) % 3 ? rnd() : (uint32_t)(i * 2654435761u >> 8);
        setenv("CSIC_PNG_THREADS", "3", 1);
        if (csic_png_write_argb(tmp.c_str(), big.data(), 700, 600, 4) != 0) { std::printf("threaded write failed\n"); return 1; }
        unsetenv("CSIC_PNG_THREADS");
        if (csic_png_read_argb(tmp.c_str(), bigback.data(), bigback.size()) != 0) { std::printf("read of the threaded file failed\n"); return 1; }
        for (size_t i = 0; i < big.size(); ++i) if ((big[i] | 0xFF000000u) != bigback[i]) { std::printf("threaded roundtrip mismatch\n"); return 1; }
    }
    long fuzzed = 0, accepted = 0;
    const std::string mut = std::string(argv[1]) + "/mut.png";
    for (int a = 2; a < argc; ++a) {
        const std::vector<unsigned char> orig = slurp(argv[a]);
        if (orig.empty()) { std::printf("cannot read %s\n", argv[a]); return 1; }
        int32_t w = 0, h = 0;
        if (csic_png_info(argv[a], &w, &h) != 0) { std::printf("info failed on %s\n", argv[a]); return 1; }
        std::vector<uint32_t> dst((size_t)w * h);
        const int iters = orig.size() > 100000 ? 60 : 400;
        for (int it = 0; it < iters; ++it) {
            std::vector<unsigned char> m = orig;
            const int nmut = 1 + rnd() % 4;
            for (int k = 0; k < nmut; ++k) {
                const size_t pos = 8 + rnd() % (m.size() - 8);
                switch (rnd() % 4) {
                case 0: m[pos] ^= 1u << (rnd() % 8); break;
                case 1: m[pos] = (unsigned char)rnd(); break;
                case 2: if (m.size() > 64) m.resize(m.size() - 1 - rnd() % 32); break;   // truncate
                default: m[16 + rnd() % 13] = (unsigned char)(rnd() % 20); break;       // IHDR fields
                }
            }
            if (rnd() % 4) resign(m);
            FILE *f = std::fopen(mut.c_str(), "wb");
            std::fwrite(m.data(), 1, m.size(), f); std::fclose(f);
            int32_t w2, h2;
            if (csic_png_info(mut.c_str(), &w2, &h2) == 0) {
                // the destination is sized from the (possibly mutated) header, like a real caller would do
                if ((int64_t)w2 * h2 <= (1 << 24)) {
                    std::vector<uint32_t> d2((size_t)w2 * h2);
                    if (csic_png_read_argb(mut.c_str(), d2.data(), d2.size()) == 0) ++accepted;
                }
                (void)csic_png_read_argb(mut.c_str(), dst.data(), dst.size());   // and with a mismatching size
            }
            ++fuzzed;
        }
    }
    std::printf("sanitize ok: %ld parameter sets, %ld fuzzed PNGs (%ld still decodable)\n", checked, fuzzed, accepted);
    return 0;
}
