/*
 * csic_jni.c -- 1:1 JNI glue between the Scala host layer (jpeg.NativeBackend) and the C ABI of
 * include/csic.h.  Build where a JDK is:
 *
 *   cc -O2 -fPIC -shared -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" -I../../../include \
 *      jni/csic_jni.c -L.. -lcsic_hip -Wl,-rpath,'$ORIGIN' -o ../libcsic_jni.so
 *
 * This repository's image has no JDK; here the file is compiled against tests/cpp/jni_stub/jni.h (the JNI types and
 * the function-table slots used below) and every export is driven through a fake JNIEnv by tests/cpp/jni_harness.c
 * (tests/test_jni_harness.py; on the GPU: in16.png -> the reference's golden spatial_downsampler_integration_420_sf2.png
 * bit for bit).  That is a compile-and-logic check of this file, not a proof against a JVM's ABI.
 *
 * Symbol names.  jpeg.NativeBackend is a Scala `object`: scalac compiles its `@native def`s to INSTANCE methods
 * of the module class `jpeg.NativeBackend$` (the static forwarders it adds to class `NativeBackend` are plain
 * bytecode that calls MODULE$.m(), not native methods).  The JVM therefore resolves
 *     Java_jpeg_NativeBackend_00024_<method>(JNIEnv *, jobject self, ...)
 * -- `$` is escaped as `_00024` (JNI spec, "Resolving Native Method Names") and the second parameter is the
 * module instance, not a jclass.  tests/test_jni_symbols.py derives these names from NativeBackend.scala and
 * checks them against this file.
 *
 * Pixel arrays.  No GetPrimitiveArrayCritical: a critical region must not span a blocking call (JNI spec, "Get/Release
 * PrimitiveArrayCritical": the GC of the whole JVM may be stalled while one is open), and a frame is a GPU round trip.
 * Instead a handle owns, besides its csic_plan, a depth-1 csic_pipeline, whose frame slots are pinned host memory the
 * GPU reads and writes directly: GetIntArrayRegion copies the Java array straight into the pinned input slot, the fused
 * kernel runs on it, SetIntArrayRegion copies the pinned output slot into the Java array -- one copy per direction, no
 * pageable staging, no region held across the launch.  No jarray reference is kept past a call.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "csic.h"

typedef struct csic_jni_handle {
    csic_plan *plan;
    csic_pipeline *pipe;        /* created by the first process(); NULL again after a failed frame */
    size_t in_px, out_px;       /* out_px: 4-byte words per output frame (a planar plan: frame_bytes / 4) */
    int planar;                 /* the plan's out_format is CSIC_FMT_PLANAR */
} csic_jni_handle;

static int is_require(int status) { return status <= CSIC_EINVAL_NULL && status >= CSIC_EINVAL_SIZE; }

static void throw_msg(JNIEnv *env, const char *cls_name, const char *msg)
{
    jclass cls = (*env)->FindClass(env, cls_name);
    if (cls) (*env)->ThrowNew(env, cls, msg);      /* FindClass failing leaves its own NoClassDefFoundError pending */
}

static void throw_for(JNIEnv *env, int status)
{
    /* every CSIC_EINVAL_* is a require() of the reference -> IllegalArgumentException("requirement failed: ...") */
    const char *msg = csic_last_error();
    if (!msg || !*msg) msg = csic_strerror(status);
    if (is_require(status)) {
        char buf[640];
        snprintf(buf, sizeof buf, "requirement failed: %s", msg);
        throw_msg(env, "java/lang/IllegalArgumentException", buf);
    } else {
        throw_msg(env, status == CSIC_ENOMEM ? "java/lang/OutOfMemoryError" : "java/lang/RuntimeException", msg);
    }
}

/* int[16] in csic_params field order; 0 = an exception is pending */
static int fill(JNIEnv *env, jintArray jp, csic_params *p)
{
    jint v[16];
    if (!jp) { throw_msg(env, "java/lang/NullPointerException", "params is null"); return 0; }
    if ((*env)->GetArrayLength(env, jp) != 16) {
        throw_msg(env, "java/lang/IllegalArgumentException", "requirement failed: params must be an Array[Int](16) in csic_params field order");
        return 0;
    }
    (*env)->GetIntArrayRegion(env, jp, 0, 16, v);
    if ((*env)->ExceptionCheck(env)) return 0;
    memcpy(p, v, sizeof *p);
    return 1;
}

JNIEXPORT void JNICALL Java_jpeg_NativeBackend_00024_validate(JNIEnv *env, jobject self, jintArray jp)
{
    (void)self;
    csic_params p;
    if (!fill(env, jp, &p)) return;
    int st = csic_validate(&p);
    if (st != CSIC_OK) throw_for(env, st);
}

JNIEXPORT jlong JNICALL Java_jpeg_NativeBackend_00024_planCreate(JNIEnv *env, jobject self, jintArray jp, jint device)
{
    (void)self;
    csic_params p;
    if (!fill(env, jp, &p)) return 0;
    int32_t wo = 0, ho = 0;
    int st = csic_out_dims(&p, &wo, &ho);          /* validates: the require()s fire here, before the device is touched */
    if (st != CSIC_OK) { throw_for(env, st); return 0; }
    csic_jni_handle *h = (csic_jni_handle *)calloc(1, sizeof *h);
    if (!h) { throw_msg(env, "java/lang/OutOfMemoryError", "csic_jni: out of host memory"); return 0; }
    st = csic_plan_create(&p, device, &h->plan);
    if (st != CSIC_OK) { free(h); throw_for(env, st); return 0; }
    h->in_px = (size_t)p.width * (size_t)p.height;
    h->out_px = (size_t)wo * (size_t)ho;
    if (p.out_format == CSIC_FMT_PLANAR) {         /* the pipeline hands back the planar frame buffer (csic.h) */
        csic_planar_layout lay;
        st = csic_planar_layout_of(&p, &lay);
        if (st != CSIC_OK) { csic_plan_destroy(h->plan); free(h); throw_for(env, st); return 0; }
        h->out_px = (size_t)(lay.frame_bytes / 4);
        h->planar = 1;
    }
    return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL Java_jpeg_NativeBackend_00024_planDestroy(JNIEnv *env, jobject self, jlong handle)
{
    (void)env; (void)self;
    csic_jni_handle *h = (csic_jni_handle *)(intptr_t)handle;
    if (!h) return;
    if (h->pipe) csic_pipeline_destroy(h->pipe);
    csic_plan_destroy(h->plan);
    free(h);
}

JNIEXPORT jintArray JNICALL Java_jpeg_NativeBackend_00024_outDims(JNIEnv *env, jobject self, jintArray jp)
{
    (void)self;
    csic_params p;
    if (!fill(env, jp, &p)) return NULL;
    int32_t wh[2];
    int st = csic_out_dims(&p, &wh[0], &wh[1]);
    if (st != CSIC_OK) { throw_for(env, st); return NULL; }
    jintArray r = (*env)->NewIntArray(env, 2);
    if (!r) return NULL;                           /* OutOfMemoryError pending */
    (*env)->SetIntArrayRegion(env, r, 0, 2, (const jint *)wh);
    return r;
}

/* One frame through the handle's depth-1 pipeline: Java int[] -> pinned input slot -> fused kernel -> pinned output slot.
 * Returns the pinned output (valid until the next frame) or NULL with an exception pending. */
static const uint32_t *round_trip(JNIEnv *env, csic_jni_handle *h, jintArray jin, jsize nin)
{
    int st = CSIC_OK;
    if (!h->pipe) st = csic_pipeline_create(h->plan, 1, &h->pipe);
    uint32_t *pin = NULL;
    const uint32_t *pout = NULL;
    if (st == CSIC_OK) st = csic_pipeline_acquire_input(h->pipe, &pin);
    if (st == CSIC_OK) {
        (*env)->GetIntArrayRegion(env, jin, 0, nin, (jint *)pin);           /* Java heap -> pinned slot; not a critical region */
        if ((*env)->ExceptionCheck(env)) { csic_pipeline_destroy(h->pipe); h->pipe = NULL; return NULL; }
        st = csic_pipeline_submit(h->pipe, NULL);                           /* the fused kernel reads / writes the pinned slots */
    }
    if (st == CSIC_OK) st = csic_pipeline_collect(h->pipe, &pout, NULL);    /* blocks this thread only */
    if (st != CSIC_OK) {
        throw_for(env, st);                                                 /* reads csic_last_error() first ... */
        if (h->pipe) { csic_pipeline_destroy(h->pipe); h->pipe = NULL; }    /* ... a frame that failed leaves no half-used slot behind */
        return NULL;
    }
    return pout;
}

/* in: ARGB ints (Java int == CSIC_FMT_ARGB8888); out: ARGB or Y|Cb<<8|Cr<<16 per the plan's out_format (a planar plan: its
 * frame buffer as frame_bytes / 4 ints -- processPlanar hands the same bytes to an Array[Byte]) */
JNIEXPORT void JNICALL Java_jpeg_NativeBackend_00024_process(JNIEnv *env, jobject self, jlong handle, jintArray jin, jintArray jout)
{
    (void)self;
    csic_jni_handle *h = (csic_jni_handle *)(intptr_t)handle;
    if (!h || !h->plan) { throw_msg(env, "java/lang/IllegalStateException", "plan handle is closed"); return; }
    if (!jin || !jout) { throw_msg(env, "java/lang/NullPointerException", "pixel array is null"); return; }
    const jsize nin = (*env)->GetArrayLength(env, jin), nout = (*env)->GetArrayLength(env, jout);
    if ((size_t)nin != h->in_px || (size_t)nout != h->out_px) {
        char buf[256];
        snprintf(buf, sizeof buf, "requirement failed: expected %zu input and %zu output pixels, got %ld and %ld",
                 h->in_px, h->out_px, (long)nin, (long)nout);
        throw_msg(env, "java/lang/IllegalArgumentException", buf);
        return;
    }
    const uint32_t *pout = round_trip(env, h, jin, nin);
    if (pout) (*env)->SetIntArrayRegion(env, jout, 0, nout, (const jint *)pout);      /* pinned slot -> Java heap */
}

/* The subsampled planar format (csic.h: CSIC_FMT_PLANAR, csic_planar_layout) -- what the reference's README describes and its code
 * never builds (README.md:35-46, ChromaSubsampler.scala:57-65).  planarLayout: the 14 fields of csic_planar_layout as an
 * Array[Long], in declaration order; needs no device. */
JNIEXPORT jlongArray JNICALL Java_jpeg_NativeBackend_00024_planarLayout(JNIEnv *env, jobject self, jintArray jp)
{
    (void)self;
    csic_params p;
    if (!fill(env, jp, &p)) return NULL;
    csic_planar_layout lay;
    int st = csic_planar_layout_of(&p, &lay);
    if (st != CSIC_OK) { throw_for(env, st); return NULL; }
    const jlong v[14] = {lay.y_width, lay.y_height, lay.chroma_width, lay.chroma_height, lay.module_width, lay.hold_h, lay.hold_v,
                         lay.replay_last, lay.chroma_samples, lay.y_offset, lay.cb_offset, lay.cr_offset, lay.frame_bytes, lay.payload_bytes};
    jlongArray r = (*env)->NewLongArray(env, 14);
    if (!r) return NULL;                           /* OutOfMemoryError pending */
    (*env)->SetLongArrayRegion(env, r, 0, 14, v);
    return r;
}

/* in: ARGB ints; out: one planar frame buffer, frame_bytes bytes (planes at planarLayout's offsets).  The handle's plan must have
 * been created with out_format = CSIC_FMT_PLANAR.  1.5 bytes per pixel cross into the Java heap for 4:2:0 instead of 4. */
JNIEXPORT void JNICALL Java_jpeg_NativeBackend_00024_processPlanar(JNIEnv *env, jobject self, jlong handle, jintArray jin, jbyteArray jout)
{
    (void)self;
    csic_jni_handle *h = (csic_jni_handle *)(intptr_t)handle;
    if (!h || !h->plan) { throw_msg(env, "java/lang/IllegalStateException", "plan handle is closed"); return; }
    if (!jin || !jout) { throw_msg(env, "java/lang/NullPointerException", "pixel array is null"); return; }
    if (!h->planar) {
        throw_msg(env, "java/lang/IllegalArgumentException", "requirement failed: processPlanar needs a plan created with out_format = CSIC_FMT_PLANAR");
        return;
    }
    const jsize nin = (*env)->GetArrayLength(env, jin), nout = (*env)->GetArrayLength(env, jout);
    if ((size_t)nin != h->in_px || (size_t)nout != 4 * h->out_px) {
        char buf[256];
        snprintf(buf, sizeof buf, "requirement failed: expected %zu input pixels and %zu output bytes, got %ld and %ld",
                 h->in_px, 4 * h->out_px, (long)nin, (long)nout);
        throw_msg(env, "java/lang/IllegalArgumentException", buf);
        return;
    }
    const uint32_t *pout = round_trip(env, h, jin, nin);
    if (pout) (*env)->SetByteArrayRegion(env, jout, 0, nout, (const jbyte *)pout);    /* pinned slot -> Java heap */
}
