/*
 * tests/cpp/jni_harness.c -- drives every Java_jpeg_NativeBackend_00024_* export of jvm/jni/csic_jni.c through a FAKE JNIEnv
 * (test infrastructure; there is no JDK in the image).  The fake implements the function-table slots of
 * tests/cpp/jni_stub/jni.h over malloc'd int[] objects:
 *   - ThrowNew records the class name and the message (one pending exception, as in a JVM);
 *   - every Get/SetIntArrayRegion call MOVES the array's storage afterwards (new allocation, old one poisoned and freed), which
 *     is what a compacting collector may do between two JNI calls: glue that kept a raw element pointer would read poison;
 *   - Get/ReleasePrimitiveArrayCritical exist and are counted: the glue must not open a critical region around a GPU round
 *     trip (JNI spec: no blocking calls inside one), so the count has to stay 0.
 * What the reference does at this boundary: ImageProcessorParams' require()s throw IllegalArgumentException at construction
 * (/root/reference/src/main/scala/jpeg/ImageProcessor.scala:22-28, tested by src/test/scala/jpeg/SpatialDownsamplerSpec.scala:147-151)
 * and the integration flow in16x16.png -> ImageProcessor(420, sf 2) -> PNG (SpatialDownsamplerSpec.scala:172-227), whose committed
 * output is tests/golden/outputs/ip_420_sf2_16.png.
 *
 *   jni_harness cpu                          : everything that needs no device
 *   jni_harness gpu <in16.png> <ip_420_sf2_16.png> <in128.png> <app_422_888_sf2_128.png> : plus the two golden flows on device 0
 */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "csic.h"

/* the exports under test (jvm/jni/csic_jni.c) */
JNIEXPORT void JNICALL Java_jpeg_NativeBackend_00024_validate(JNIEnv *env, jobject self, jintArray jp);
JNIEXPORT jlong JNICALL Java_jpeg_NativeBackend_00024_planCreate(JNIEnv *env, jobject self, jintArray jp, jint device);
JNIEXPORT void JNICALL Java_jpeg_NativeBackend_00024_planDestroy(JNIEnv *env, jobject self, jlong handle);
JNIEXPORT jintArray JNICALL Java_jpeg_NativeBackend_00024_outDims(JNIEnv *env, jobject self, jintArray jp);
JNIEXPORT void JNICALL Java_jpeg_NativeBackend_00024_process(JNIEnv *env, jobject self, jlong handle, jintArray jin, jintArray jout);
JNIEXPORT jlongArray JNICALL Java_jpeg_NativeBackend_00024_planarLayout(JNIEnv *env, jobject self, jintArray jp);
JNIEXPORT void JNICALL Java_jpeg_NativeBackend_00024_processPlanar(JNIEnv *env, jobject self, jlong handle, jintArray jin, jbyteArray jout);

/* ---- the fake JVM ---------------------------------------------------------------------------------------------------------*/
enum { K_CLASS = 0x434c4153, K_INTARRAY = 0x494e5441, K_BYTEARRAY = 0x42595441, K_LONGARRAY = 0x4c4f4e41 };
struct _jobject {
    int kind;
    jsize len;
    jint *data;                 /* the elements (int[]: as such; byte[] / long[]: the same storage, elem bytes each) */
    char name[96];
    int elem;                   /* bytes per element (0 = 4: objects built before the field existed) */
};
static size_t elem_of(jobject a) { return a->elem ? (size_t)a->elem : sizeof(jint); }
static int is_array(jobject a) { return a && (a->kind == K_INTARRAY || a->kind == K_BYTEARRAY || a->kind == K_LONGARRAY); }

static struct {
    int pending;
    char cls[96], msg[700];
    long critical_get, critical_release, critical_open;
    long region_get, region_set, find_class, moves;
} vm;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static jobject new_object(int kind)
{
    jobject o = (jobject)calloc(1, sizeof *o);
    if (!o) { perror("calloc"); exit(2); }
    o->kind = kind;
    return o;
}

static jintArray new_int_array(jsize len)
{
    jobject o = new_object(K_INTARRAY);
    o->len = len;
    o->data = (jint *)malloc((size_t)(len > 0 ? len : 1) * sizeof(jint));
    if (!o->data) { perror("malloc"); exit(2); }
    memset(o->data, 0, (size_t)len * sizeof(jint));
    return o;
}

static jarray new_array(int kind, jsize len, int elem)
{
    jobject o = new_object(kind);
    o->len = len;
    o->elem = elem;
    o->data = (jint *)calloc((size_t)(len > 0 ? len : 1), (size_t)elem);
    if (!o->data) { perror("calloc"); exit(2); }
    return o;
}
static jbyteArray new_byte_array(jsize len) { return new_array(K_BYTEARRAY, len, 1); }

static void free_object(jobject o)
{
    if (!o) return;
    free(o->data);
    free(o);
}

/* a compacting GC between two JNI calls */
static void move_array(jobject a)
{
    const size_t bytes = (size_t)(a->len > 0 ? a->len : 1) * elem_of(a);
    jint *fresh = (jint *)malloc(bytes);
    if (!fresh) { perror("malloc"); exit(2); }
    memcpy(fresh, a->data, (size_t)a->len * elem_of(a));
    memset(a->data, 0xA5, (size_t)a->len * elem_of(a));
    free(a->data);
    a->data = fresh;
    ++vm.moves;
}

static void raise_vm(const char *cls, const char *msg)
{
    vm.pending = 1;
    snprintf(vm.cls, sizeof vm.cls, "%s", cls);
    snprintf(vm.msg, sizeof vm.msg, "%s", msg ? msg : "");
}

static jclass JNICALL f_FindClass(JNIEnv *env, const char *name)
{
    (void)env;
    ++vm.find_class;
    CHECK(vm.critical_open == 0, "FindClass inside a critical region");
    static const char *known[] = {"java/lang/IllegalArgumentException", "java/lang/RuntimeException", "java/lang/OutOfMemoryError",
                                  "java/lang/NullPointerException", "java/lang/IllegalStateException"};
    for (size_t i = 0; i < sizeof known / sizeof known[0]; ++i)
        if (strcmp(known[i], name) == 0) {
            jobject c = new_object(K_CLASS);          /* (leaked: a local reference of the fake VM) */
            snprintf(c->name, sizeof c->name, "%s", name);
            return c;
        }
    raise_vm("java/lang/NoClassDefFoundError", name);
    return NULL;
}

static jint JNICALL f_ThrowNew(JNIEnv *env, jclass clazz, const char *msg)
{
    (void)env;
    CHECK(clazz && clazz->kind == K_CLASS, "ThrowNew on something that is not a class");
    CHECK(!vm.pending, "ThrowNew with an exception already pending (%s)", vm.cls);
    CHECK(vm.critical_open == 0, "ThrowNew inside a critical region");
    if (clazz && clazz->kind == K_CLASS) raise_vm(clazz->name, msg);
    return 0;
}

static jboolean JNICALL f_ExceptionCheck(JNIEnv *env) { (void)env; return vm.pending ? JNI_TRUE : JNI_FALSE; }

static jsize JNICALL f_GetArrayLength(JNIEnv *env, jarray a)
{
    (void)env;
    CHECK(is_array(a), "GetArrayLength on a non-array");
    return a ? a->len : 0;
}

static jintArray JNICALL f_NewIntArray(JNIEnv *env, jsize len) { (void)env; return new_int_array(len); }

static void JNICALL f_GetIntArrayRegion(JNIEnv *env, jintArray a, jsize start, jsize len, jint *buf)
{
    (void)env;
    ++vm.region_get;
    CHECK(!vm.pending, "JNI call with an exception pending");
    if (!a || a->kind != K_INTARRAY || start < 0 || len < 0 || start + len > a->len) { raise_vm("java/lang/ArrayIndexOutOfBoundsException", "GetIntArrayRegion"); return; }
    memcpy(buf, a->data + start, (size_t)len * sizeof(jint));
    move_array(a);
}

static void JNICALL f_SetIntArrayRegion(JNIEnv *env, jintArray a, jsize start, jsize len, const jint *buf)
{
    (void)env;
    ++vm.region_set;
    CHECK(!vm.pending, "JNI call with an exception pending");
    if (!a || a->kind != K_INTARRAY || start < 0 || len < 0 || start + len > a->len) { raise_vm("java/lang/ArrayIndexOutOfBoundsException", "SetIntArrayRegion"); return; }
    memcpy(a->data + start, buf, (size_t)len * sizeof(jint));
    move_array(a);
}

static jlongArray JNICALL f_NewLongArray(JNIEnv *env, jsize len) { (void)env; return new_array(K_LONGARRAY, len, 8); }

static void JNICALL f_SetByteArrayRegion(JNIEnv *env, jbyteArray a, jsize start, jsize len, const jbyte *buf)
{
    (void)env;
    ++vm.region_set;
    CHECK(!vm.pending, "JNI call with an exception pending");
    if (!a || a->kind != K_BYTEARRAY || start < 0 || len < 0 || start + len > a->len) { raise_vm("java/lang/ArrayIndexOutOfBoundsException", "SetByteArrayRegion"); return; }
    memcpy((jbyte *)a->data + start, buf, (size_t)len);
    move_array(a);
}

static void JNICALL f_SetLongArrayRegion(JNIEnv *env, jlongArray a, jsize start, jsize len, const jlong *buf)
{
    (void)env;
    ++vm.region_set;
    CHECK(!vm.pending, "JNI call with an exception pending");
    if (!a || a->kind != K_LONGARRAY || start < 0 || len < 0 || start + len > a->len) { raise_vm("java/lang/ArrayIndexOutOfBoundsException", "SetLongArrayRegion"); return; }
    memcpy((jlong *)a->data + start, buf, (size_t)len * sizeof(jlong));
    move_array(a);
}

static jint *JNICALL f_GetIntArrayElements(JNIEnv *env, jintArray a, jboolean *is_copy)
{
    (void)env;
    if (is_copy) *is_copy = JNI_TRUE;
    jint *c = (jint *)malloc((size_t)(a->len > 0 ? a->len : 1) * sizeof(jint));
    memcpy(c, a->data, (size_t)a->len * sizeof(jint));
    return c;
}

static void JNICALL f_ReleaseIntArrayElements(JNIEnv *env, jintArray a, jint *elems, jint mode)
{
    (void)env;
    if (mode != JNI_ABORT) memcpy(a->data, elems, (size_t)a->len * sizeof(jint));
    if (mode != JNI_COMMIT) free(elems);
}

static void *JNICALL f_GetPrimitiveArrayCritical(JNIEnv *env, jarray a, jboolean *is_copy)
{
    (void)env;
    ++vm.critical_get; ++vm.critical_open;
    if (is_copy) *is_copy = JNI_FALSE;
    return a->data;
}

static void JNICALL f_ReleasePrimitiveArrayCritical(JNIEnv *env, jarray a, void *carray, jint mode)
{
    (void)env; (void)a; (void)carray; (void)mode;
    ++vm.critical_release; --vm.critical_open;
}

static struct JNINativeInterface_ table;
static JNIEnv env_value = &table;
static JNIEnv *env = &env_value;
static struct _jobject module_instance = {0, 0, NULL, "jpeg.NativeBackend$", 0};

static void vm_init(void)
{
    memset(&table, 0, sizeof table);      /* every slot the glue must not use is NULL: a call through one crashes the harness */
    table.FindClass = f_FindClass;
    table.ThrowNew = f_ThrowNew;
    table.ExceptionCheck = f_ExceptionCheck;
    table.GetArrayLength = f_GetArrayLength;
    table.NewIntArray = f_NewIntArray;
    table.NewLongArray = f_NewLongArray;
    table.SetByteArrayRegion = f_SetByteArrayRegion;
    table.SetLongArrayRegion = f_SetLongArrayRegion;
    table.GetIntArrayRegion = f_GetIntArrayRegion;
    table.SetIntArrayRegion = f_SetIntArrayRegion;
    table.GetIntArrayElements = f_GetIntArrayElements;
    table.ReleaseIntArrayElements = f_ReleaseIntArrayElements;
    table.GetPrimitiveArrayCritical = f_GetPrimitiveArrayCritical;
    table.ReleasePrimitiveArrayCritical = f_ReleasePrimitiveArrayCritical;
}

static void clear_pending(void) { vm.pending = 0; vm.cls[0] = 0; vm.msg[0] = 0; }

static int thrown(const char *cls, const char *msg_part)
{
    const int ok = vm.pending && strcmp(vm.cls, cls) == 0 && (!msg_part || strstr(vm.msg, msg_part));
    if (!ok) printf("  (pending=%d class='%s' message='%s'; wanted %s containing '%s')\n", vm.pending, vm.cls, vm.msg, cls, msg_part ? msg_part : "");
    clear_pending();
    return ok;
}

/* NativeBackend.pack (jvm/scala/jpeg/NativeBackend.scala): Array[Int](16) in csic_params field order */
static jintArray pack(int w, int h, int a, int b, int yq, int cbq, int crq, int sf, int op1, int op2, int op3, int rounding, int out_format,
                      int strict)
{
    jintArray p = new_int_array(16);
    const jint v[16] = {w, h, a, b, yq, cbq, crq, sf, op1, op2, op3, rounding, 0, CSIC_FMT_ARGB8888, out_format, strict};
    memcpy(p->data, v, sizeof v);
    return p;
}
/* case class ImageProcessorParams(width, height, factor, chromaParamA, chromaParamB) -- GpuImageCompressor.scala */
static jintArray image_processor_params(int w, int h, int factor, int a, int b)
{
    return pack(w, h, a, b, 8, 8, 8, factor, 3, 1, 2, CSIC_ROUND_FLOOR_HW, CSIC_FMT_ARGB8888, 1);
}

static void expect_iae(const char *what, jintArray p, const char *msg_part)
{
    Java_jpeg_NativeBackend_00024_validate(env, &module_instance, p);
    CHECK(thrown("java/lang/IllegalArgumentException", "requirement failed: "), "%s: no IllegalArgumentException(\"requirement failed: ...\")", what);
    Java_jpeg_NativeBackend_00024_validate(env, &module_instance, p);
    CHECK(thrown("java/lang/IllegalArgumentException", msg_part), "%s: message does not mention '%s'", what, msg_part);
    free_object(p);
}

static void host_checks(void)
{
    /* a valid parameter set: no exception */
    jintArray ok = image_processor_params(16, 16, 2, 2, 0);
    Java_jpeg_NativeBackend_00024_validate(env, &module_instance, ok);
    CHECK(!vm.pending, "validate threw %s: %s", vm.cls, vm.msg);
    clear_pending();

    /* SpatialDownsamplerSpec.scala:147-151 "fail for invalid factor": ImageProcessorParams(8, 8, 3, 4, 4) -> IllegalArgumentException */
    expect_iae("factor 3", image_processor_params(8, 8, 3, 4, 4), "actor");
    /* the other require()s of ImageProcessor.scala:22-28, ColorQuantizer.scala:12-15, ImageCompressorTop.scala:27-31 */
    expect_iae("width 0", image_processor_params(0, 8, 2, 4, 4), "idth");
    expect_iae("height -1", image_processor_params(8, -1, 2, 4, 4), "eight");
    expect_iae("not divisible", image_processor_params(9, 8, 2, 4, 4), "ivisible");
    expect_iae("chroma a 3", image_processor_params(8, 8, 2, 3, 3), "hroma");
    expect_iae("chroma b 1", image_processor_params(8, 8, 2, 2, 1), "hroma");
    expect_iae("y bits 0", pack(8, 8, 4, 4, 0, 8, 8, 1, 3, 1, 2, 0, 0, 0), "its");
    expect_iae("cr bits 9", pack(8, 8, 4, 4, 8, 8, 9, 1, 3, 1, 2, 0, 0, 0), "its");
    expect_iae("ops not a permutation", pack(8, 8, 4, 4, 8, 8, 8, 1, 3, 3, 2, 0, 0, 0), "ermutation");
    /* non-strict: ImageCompressorTop accepts dimensions the factor does not divide */
    jintArray loose = pack(5, 3, 4, 4, 8, 8, 8, 2, 3, 1, 2, 0, 0, 0);
    Java_jpeg_NativeBackend_00024_validate(env, &module_instance, loose);
    CHECK(!vm.pending, "non-strict 5x3 sf 2 threw %s: %s", vm.cls, vm.msg);
    clear_pending();
    /* outDims: ceil sizes, 5x3 sf 2 -> 3x2 (SpatialDownsamplerSpec.scala:120-122: 6 pixels) */
    jintArray d = Java_jpeg_NativeBackend_00024_outDims(env, &module_instance, loose);
    CHECK(!vm.pending && d && d->kind == K_INTARRAY && d->len == 2 && d->data[0] == 3 && d->data[1] == 2, "outDims(5x3, sf 2) is not 3x2");
    clear_pending();
    free_object(d);
    free_object(loose);
    jintArray bad = image_processor_params(8, 8, 3, 4, 4);
    d = Java_jpeg_NativeBackend_00024_outDims(env, &module_instance, bad);
    CHECK(d == NULL && thrown("java/lang/IllegalArgumentException", "requirement failed: "), "outDims(factor 3) did not throw");
    /* planCreate: the require()s fire before any device is looked at */
    jlong h = Java_jpeg_NativeBackend_00024_planCreate(env, &module_instance, bad, 0);
    CHECK(h == 0 && thrown("java/lang/IllegalArgumentException", "requirement failed: "), "planCreate(factor 3) did not throw IllegalArgumentException");
    free_object(bad);
    /* a device that does not exist (or no GPU at all): a RuntimeException carrying the library's message, never a crash or a CPU fallback */
    h = Java_jpeg_NativeBackend_00024_planCreate(env, &module_instance, ok, 4096);
    CHECK(h == 0 && thrown("java/lang/RuntimeException", "evice"), "planCreate(device 4096) did not throw RuntimeException");
    /* malformed params arrays */
    jintArray shortp = new_int_array(11);
    Java_jpeg_NativeBackend_00024_validate(env, &module_instance, shortp);
    CHECK(thrown("java/lang/IllegalArgumentException", "Array[Int](16)"), "an 11-element params array was accepted");
    free_object(shortp);
    Java_jpeg_NativeBackend_00024_validate(env, &module_instance, NULL);
    CHECK(thrown("java/lang/NullPointerException", NULL), "null params were accepted");
    /* closed / null handles */
    Java_jpeg_NativeBackend_00024_planDestroy(env, &module_instance, 0);
    CHECK(!vm.pending, "planDestroy(0) threw");
    jintArray one = new_int_array(1);
    Java_jpeg_NativeBackend_00024_process(env, &module_instance, 0, one, one);
    CHECK(thrown("java/lang/IllegalStateException", "closed"), "process on a null handle did not throw IllegalStateException");
    free_object(one);
    /* the planar layout is host arithmetic (csic_planar_layout_of): 64x16 4:2:0 at factor 1 -> 1024 Y bytes + 2 x 256 samples */
    jintArray pl = pack(64, 16, 2, 0, 8, 8, 8, 1, 3, 1, 2, CSIC_ROUND_FLOOR_HW, CSIC_FMT_PLANAR, 0);
    jlongArray lay = Java_jpeg_NativeBackend_00024_planarLayout(env, &module_instance, pl);
    CHECK(!vm.pending && lay && lay->kind == K_LONGARRAY && lay->len == 14, "planarLayout did not return an Array[Long](14)");
    if (lay && lay->len == 14) {
        const jlong *v = (const jlong *)lay->data;
        CHECK(v[0] == 64 && v[1] == 16 && v[2] == 32 && v[3] == 8 && v[4] == 64 && v[5] == 2 && v[6] == 2 && v[7] == 1 && v[8] == 256,
              "planarLayout(64x16 4:2:0): %ld %ld %ld %ld %ld %ld %ld %ld %ld", (long)v[0], (long)v[1], (long)v[2], (long)v[3], (long)v[4], (long)v[5], (long)v[6], (long)v[7], (long)v[8]);
        CHECK(v[9] == 0 && v[10] % 256 == 0 && v[11] % 256 == 0 && v[12] % 256 == 0 && v[13] == 1536, "planarLayout offsets / sizes: %ld %ld %ld %ld %ld",
              (long)v[9], (long)v[10], (long)v[11], (long)v[12], (long)v[13]);
    }
    clear_pending();
    free_object(lay);
    free_object(pl);
    jintArray badp = image_processor_params(8, 8, 3, 4, 4);
    lay = Java_jpeg_NativeBackend_00024_planarLayout(env, &module_instance, badp);
    CHECK(lay == NULL && thrown("java/lang/IllegalArgumentException", "requirement failed: "), "planarLayout(factor 3) did not throw");
    free_object(badp);
    jbyteArray nob = new_byte_array(1);
    jintArray one2 = new_int_array(1);
    Java_jpeg_NativeBackend_00024_processPlanar(env, &module_instance, 0, one2, nob);
    CHECK(thrown("java/lang/IllegalStateException", "closed"), "processPlanar on a null handle did not throw IllegalStateException");
    free_object(nob);
    free_object(one2);
    free_object(ok);
    printf("ok: host checks (requires -> IllegalArgumentException, device errors -> RuntimeException, outDims)\n");
}

static jintArray read_png(const char *path, int32_t *w, int32_t *h)
{
    if (csic_png_info(path, w, h) != CSIC_OK) { printf("FAIL cannot read %s: %s\n", path, csic_last_error()); exit(2); }
    jintArray a = new_int_array(*w * *h);
    if (csic_png_read_argb(path, (uint32_t *)a->data, (size_t)a->len) != CSIC_OK) { printf("FAIL cannot decode %s: %s\n", path, csic_last_error()); exit(2); }
    return a;
}

static void golden_flow(const char *what, jintArray params, const char *in_path, const char *golden_path)
{
    int32_t w, h, gw, gh;
    jintArray in = read_png(in_path, &w, &h);
    jintArray golden = read_png(golden_path, &gw, &gh);
    jintArray dims = Java_jpeg_NativeBackend_00024_outDims(env, &module_instance, params);
    CHECK(!vm.pending && dims && dims->data[0] == gw && dims->data[1] == gh, "%s: outDims %dx%d, golden is %dx%d", what, dims ? dims->data[0] : -1,
          dims ? dims->data[1] : -1, gw, gh);
    clear_pending();
    jlong plan = Java_jpeg_NativeBackend_00024_planCreate(env, &module_instance, params, 0);
    CHECK(plan != 0 && !vm.pending, "%s: planCreate failed: %s: %s", what, vm.cls, vm.msg);
    clear_pending();
    if (plan) {
        for (int rep = 0; rep < 3; ++rep) {          /* the second and third frame reuse the handle's pinned slots */
            jintArray out = new_int_array(gw * gh);
            memset(out->data, 0x5A, (size_t)out->len * sizeof(jint));
            Java_jpeg_NativeBackend_00024_process(env, &module_instance, plan, in, out);
            CHECK(!vm.pending, "%s: process threw %s: %s", what, vm.cls, vm.msg);
            clear_pending();
            long bad = 0;
            for (jsize i = 0; i < out->len; ++i) bad += out->data[i] != golden->data[i];
            CHECK(bad == 0, "%s (frame %d): %ld of %d pixels differ from %s", what, rep, bad, (int)out->len, golden_path);
            free_object(out);
        }
        /* wrong array sizes are a require(), reported before anything is copied */
        jintArray small = new_int_array(gw * gh - 1);
        Java_jpeg_NativeBackend_00024_process(env, &module_instance, plan, in, small);
        CHECK(thrown("java/lang/IllegalArgumentException", "requirement failed: expected"), "%s: a short output array was accepted", what);
        free_object(small);
        Java_jpeg_NativeBackend_00024_planDestroy(env, &module_instance, plan);
        CHECK(!vm.pending, "planDestroy threw");
    }
    printf("ok: %s: %dx%d -> %dx%d bit-exact against %s (3 frames)\n", what, w, h, gw, gh, golden_path);
    free_object(dims);
    free_object(in);
    free_object(golden);
    free_object(params);
}

/* The planar format through the shim: a handle created with out_format = CSIC_FMT_PLANAR, processPlanar into an Array[Byte], against
 * the packed YCbCr stream of a second handle with the same parameters -- the Y plane is the stream's Y bytes, the chroma planes are
 * its Cb / Cr at the layout's sample points (hold_h x hold_v, csic.h) -- three frames through one handle. */
static void planar_flow(const char *what, int w, int h, int a, int b, int sf, const char *in_path)
{
    int32_t iw, ih;
    jintArray in = read_png(in_path, &iw, &ih);
    CHECK(iw == w && ih == h, "%s: %s is %dx%d", what, in_path, iw, ih);
    jintArray pp = pack(w, h, a, b, 8, 8, 8, sf, 3, 1, 2, CSIC_ROUND_FLOOR_HW, CSIC_FMT_PLANAR, 0);
    jintArray py = pack(w, h, a, b, 8, 8, 8, sf, 3, 1, 2, CSIC_ROUND_FLOOR_HW, CSIC_FMT_YCBCR888X, 0);
    jlongArray lay = Java_jpeg_NativeBackend_00024_planarLayout(env, &module_instance, pp);
    CHECK(!vm.pending && lay && lay->len == 14, "%s: planarLayout failed", what);
    clear_pending();
    const jlong *L = (const jlong *)lay->data;
    const int yw = (int)L[0], yh = (int)L[1], cw = (int)L[2], hh = (int)L[5], hv = (int)L[6];
    const jlong cb_off = L[10], cr_off = L[11], frame_bytes = L[12];
    jlong hplanar = Java_jpeg_NativeBackend_00024_planCreate(env, &module_instance, pp, 0);
    jlong hycc = Java_jpeg_NativeBackend_00024_planCreate(env, &module_instance, py, 0);
    CHECK(hplanar != 0 && hycc != 0 && !vm.pending, "%s: planCreate failed: %s: %s", what, vm.cls, vm.msg);
    clear_pending();
    if (hplanar && hycc) {
        jintArray ycc = new_int_array(yw * yh);
        Java_jpeg_NativeBackend_00024_process(env, &module_instance, hycc, in, ycc);
        CHECK(!vm.pending, "%s: process (YCbCr) threw %s: %s", what, vm.cls, vm.msg);
        clear_pending();
        for (int rep = 0; rep < 3; ++rep) {
            jbyteArray out = new_byte_array((jsize)frame_bytes);
            memset(out->data, 0x5A, (size_t)frame_bytes);
            Java_jpeg_NativeBackend_00024_processPlanar(env, &module_instance, hplanar, in, out);
            CHECK(!vm.pending, "%s: processPlanar threw %s: %s", what, vm.cls, vm.msg);
            clear_pending();
            const uint8_t *bytes = (const uint8_t *)out->data;
            long bad = 0;
            for (int r = 0; r < yh; ++r)
                for (int c = 0; c < yw; ++c) {
                    const uint32_t px = (uint32_t)ycc->data[r * yw + c];
                    bad += bytes[r * yw + c] != (px & 0xFF);
                    if (r % hv == 0 && c % hh == 0) {
                        const jlong k = (jlong)(r / hv) * cw + c / hh;
                        bad += bytes[cb_off + k] != ((px >> 8) & 0xFF);
                        bad += bytes[cr_off + k] != ((px >> 16) & 0xFF);
                    }
                }
            CHECK(bad == 0, "%s (frame %d): %ld plane bytes differ from the packed YCbCr stream", what, rep, bad);
            free_object(out);
        }
        /* the same frame buffer through process(): frame_bytes / 4 ints */
        jintArray words = new_int_array((jsize)(frame_bytes / 4));
        Java_jpeg_NativeBackend_00024_process(env, &module_instance, hplanar, in, words);
        CHECK(!vm.pending && (((const uint8_t *)words->data)[0] == ((uint32_t)ycc->data[0] & 0xFF)), "%s: process() on a planar handle", what);
        clear_pending();
        free_object(words);
        /* wrong sizes and the wrong kind of handle are require()s */
        jbyteArray small = new_byte_array((jsize)frame_bytes - 1);
        Java_jpeg_NativeBackend_00024_processPlanar(env, &module_instance, hplanar, in, small);
        CHECK(thrown("java/lang/IllegalArgumentException", "requirement failed: expected"), "%s: a short byte array was accepted", what);
        free_object(small);
        jbyteArray full = new_byte_array((jsize)frame_bytes);
        Java_jpeg_NativeBackend_00024_processPlanar(env, &module_instance, hycc, in, full);
        CHECK(thrown("java/lang/IllegalArgumentException", "CSIC_FMT_PLANAR"), "%s: processPlanar on a packed plan was accepted", what);
        free_object(full);
        free_object(ycc);
    }
    Java_jpeg_NativeBackend_00024_planDestroy(env, &module_instance, hplanar);
    Java_jpeg_NativeBackend_00024_planDestroy(env, &module_instance, hycc);
    printf("ok: %s: %dx%d -> planes %dx%d + 2 x %ld samples (hold %dx%d) equal to the packed YCbCr stream (3 frames)\n", what, w, h, yw, yh, (long)L[8], hh, hv);
    free_object(lay);
    free_object(in);
    free_object(pp);
    free_object(py);
}

int main(int argc, char **argv)
{
    setvbuf(stdout, NULL, _IOLBF, 0);
    if (argc < 2 || (strcmp(argv[1], "cpu") != 0 && strcmp(argv[1], "gpu") != 0) || (strcmp(argv[1], "gpu") == 0 && argc < 6)) {
        fprintf(stderr, "usage: %s cpu | gpu <in16.png> <ip_420_sf2_16.png> <in128.png> <app_422_888_sf2_128.png>\n", argv[0]);
        return 2;
    }
    vm_init();
    host_checks();
    if (strcmp(argv[1], "gpu") == 0) {
        if (csic_device_count() <= 0) { printf("FAIL no HIP device: %s\n", csic_last_error()); return 1; }
        /* SpatialDownsamplerSpec.scala:172-227: ImageProcessorParams(16, 16, factor 2, 4:2:0) on in16x16.png */
        golden_flow("ImageProcessor integration flow", image_processor_params(16, 16, 2, 2, 0), argv[2], argv[3]);
        /* ImageCompressorTopApp.scala:189-190: --a 2 --b 2 --sf 2, order chroma, spatial, color on in128x128.png */
        golden_flow("ImageCompressionApp flow", pack(128, 128, 2, 2, 8, 8, 8, 2, 3, 1, 2, CSIC_ROUND_FLOOR_HW, CSIC_FMT_ARGB8888, 0), argv[4], argv[5]);
        /* round 4: the subsampled planar format -- 4:2:0 at factor 1 (samples on even rows / columns) and at factor 2 (one per pixel) */
        planar_flow("planar 4:2:0 sf 1", 16, 16, 2, 0, 1, argv[2]);
        planar_flow("planar 4:2:2 sf 2", 128, 128, 2, 2, 2, argv[4]);
    }
    CHECK(vm.critical_get == 0 && vm.critical_release == 0, "the glue opened %ld critical regions", vm.critical_get);
    printf("jni calls: FindClass %ld, GetIntArrayRegion %ld, SetIntArrayRegion %ld, array moves %ld, critical regions %ld\n", vm.find_class,
           vm.region_get, vm.region_set, vm.moves, vm.critical_get);
    if (failures) { printf("%d check(s) FAILED\n", failures); return 1; }
    printf("all checks passed\n");
    return 0;
}
