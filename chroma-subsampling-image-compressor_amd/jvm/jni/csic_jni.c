/*
 * csic_jni.c -- 1:1 JNI glue between the Scala host layer (jpeg.NativeBackend) and the C ABI of
 * include/csic.h.  NOT compiled in this repository's image (no JDK / jni.h here); build where a JDK is:
 *
 *   cc -O2 -fPIC -shared -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" -I../../../include \
 *      jni/csic_jni.c -L.. -lcsic_hip -Wl,-rpath,'$ORIGIN' -o ../libcsic_jni.so
 *
 * Symbol names.  jpeg.NativeBackend is a Scala `object`: scalac compiles its `@native def`s to INSTANCE methods
 * of the module class `jpeg.NativeBackend$` (the static forwarders it adds to class `NativeBackend` are plain
 * bytecode that calls MODULE$.m(), not native methods).  The JVM therefore resolves
 *     Java_jpeg_NativeBackend_00024_<method>(JNIEnv *, jobject self, ...)
 * -- `$` is escaped as `_00024` (JNI spec, "Resolving Native Method Names") and the second parameter is the
 * module instance, not a jclass.  tests/test_jni_symbols.py derives these names from NativeBackend.scala and
 * checks them against this file, so the two cannot drift apart unnoticed while no JDK is available to link them.
 *
 * Ownership: no jarray reference is kept past a call; pixel arrays are pinned with
 * GetPrimitiveArrayCritical only for the duration of csic_process_host.
 */
#include <jni.h>
#include <stdint.h>
#include <string.h>

#include "csic.h"

static void throw_for(JNIEnv *env, int status)
{
    /* every CSIC_EINVAL_* is a require() of the reference -> IllegalArgumentException */
    const char *cls = (status <= CSIC_EINVAL_NULL && status >= CSIC_EINVAL_SIZE)
                          ? "java/lang/IllegalArgumentException" : "java/lang/RuntimeException";
    const char *msg = csic_last_error();
    char buf[640];
    if (status <= CSIC_EINVAL_NULL && status >= CSIC_EINVAL_SIZE) {
        strcpy(buf, "requirement failed: ");
        strncat(buf, (msg && *msg) ? msg : csic_strerror(status), sizeof buf - 24);
        msg = buf;
    }
    (*env)->ThrowNew(env, (*env)->FindClass(env, cls), (msg && *msg) ? msg : csic_strerror(status));
}

static void fill(JNIEnv *env, jintArray jp, csic_params *p)
{
    /* int[16] in csic_params field order */
    jint v[16];
    (*env)->GetIntArrayRegion(env, jp, 0, 16, v);
    memcpy(p, v, sizeof *p);
}

JNIEXPORT void JNICALL Java_jpeg_NativeBackend_00024_validate(JNIEnv *env, jobject self, jintArray jp)
{
    (void)self;
    csic_params p; fill(env, jp, &p);
    int st = csic_validate(&p);
    if (st != CSIC_OK) throw_for(env, st);
}

JNIEXPORT jlong JNICALL Java_jpeg_NativeBackend_00024_planCreate(JNIEnv *env, jobject self, jintArray jp, jint device)
{
    (void)self;
    csic_params p; fill(env, jp, &p);
    csic_plan *plan = NULL;
    int st = csic_plan_create(&p, device, &plan);
    if (st != CSIC_OK) { throw_for(env, st); return 0; }
    return (jlong)(intptr_t)plan;
}

JNIEXPORT void JNICALL Java_jpeg_NativeBackend_00024_planDestroy(JNIEnv *env, jobject self, jlong h)
{
    (void)env; (void)self;
    csic_plan_destroy((csic_plan *)(intptr_t)h);
}

JNIEXPORT jintArray JNICALL Java_jpeg_NativeBackend_00024_outDims(JNIEnv *env, jobject self, jintArray jp)
{
    (void)self;
    csic_params p; fill(env, jp, &p);
    int32_t wh[2];
    int st = csic_out_dims(&p, &wh[0], &wh[1]);
    if (st != CSIC_OK) { throw_for(env, st); return NULL; }
    jintArray r = (*env)->NewIntArray(env, 2);
    (*env)->SetIntArrayRegion(env, r, 0, 2, (const jint *)wh);
    return r;
}

/* in: ARGB ints (Java int == CSIC_FMT_ARGB8888); out: ARGB or Y|Cb<<8|Cr<<16 per the plan's out_format */
JNIEXPORT void JNICALL Java_jpeg_NativeBackend_00024_process(JNIEnv *env, jobject self, jlong h, jintArray jin, jintArray jout)
{
    (void)self;
    const jsize nin = (*env)->GetArrayLength(env, jin), nout = (*env)->GetArrayLength(env, jout);
    void *pin = (*env)->GetPrimitiveArrayCritical(env, jin, NULL);
    void *pout = (*env)->GetPrimitiveArrayCritical(env, jout, NULL);
    int st = (pin && pout) ? csic_process_host((csic_plan *)(intptr_t)h, (const uint32_t *)pin, (size_t)nin,
                                               (uint32_t *)pout, (size_t)nout)
                           : CSIC_ENOMEM;
    if (pout) (*env)->ReleasePrimitiveArrayCritical(env, jout, pout, 0);
    if (pin) (*env)->ReleasePrimitiveArrayCritical(env, jin, pin, JNI_ABORT);
    if (st != CSIC_OK) throw_for(env, st);
}
