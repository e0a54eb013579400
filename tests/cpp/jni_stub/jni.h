/*
 * tests/cpp/jni_stub/jni.h -- TEST-ONLY stand-in for a JDK's <jni.h>, so that jvm/jni/csic_jni.c can meet a compiler and be
 * driven through a fake JNIEnv (tests/cpp/jni_harness.c) in an image that has no JDK.  Written from the JNI specification
 * (Java SE "JNI Types and Data Structures", "JNI Functions: Interface Function Table"); not copied from any JDK.
 *
 * It declares the primitive / reference types csic_jni.c uses and a function table that has, AT THE INDICES THE
 * SPECIFICATION GIVES THEM, the slots csic_jni.c calls; every other slot is an anonymous pointer.  The indices are written
 * down from the specification's table and asserted below with offsetof -- they have not been compared with a real JDK
 * header here (there is none), which is one reason why passing the harness is a compile-and-logic check and not an ABI proof.
 * Never install this file or put it on the include path of a build that has a JDK.
 */
#ifndef CSIC_TEST_JNI_STUB_H
#define CSIC_TEST_JNI_STUB_H

#include <stddef.h>
#include <stdint.h>

#define CSIC_JNI_STUB 1

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef jint jsize;

struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jbyteArray;
typedef jarray jlongArray;

#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_OK 0
#define JNI_COMMIT 1
#define JNI_ABORT 2

#define JNIEXPORT __attribute__((visibility("default")))
#define JNIIMPORT
#define JNICALL

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;     /* C binding: JNIEnv is a pointer to the function table */

struct JNINativeInterface_ {
    void *slot_0_5[6];                                                            /* 0-3 reserved, 4 GetVersion, 5 DefineClass */
    jclass (JNICALL *FindClass)(JNIEnv *env, const char *name);                   /* 6 */
    void *slot_7_13[7];
    jint (JNICALL *ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);         /* 14 */
    void *slot_15_170[156];
    jsize (JNICALL *GetArrayLength)(JNIEnv *env, jarray array);                   /* 171 */
    void *slot_172_178[7];
    jintArray (JNICALL *NewIntArray)(JNIEnv *env, jsize len);                     /* 179 */
    jlongArray (JNICALL *NewLongArray)(JNIEnv *env, jsize len);                   /* 180 */
    void *slot_181_186[6];
    jint *(JNICALL *GetIntArrayElements)(JNIEnv *env, jintArray array, jboolean *isCopy);               /* 187 */
    void *slot_188_194[7];
    void (JNICALL *ReleaseIntArrayElements)(JNIEnv *env, jintArray array, jint *elems, jint mode);      /* 195 */
    void *slot_196_202[7];
    void (JNICALL *GetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, jint *buf); /* 203 */
    void *slot_204_207[4];
    void (JNICALL *SetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize len, const jbyte *buf); /* 208 */
    void *slot_209_210[2];
    void (JNICALL *SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf); /* 211 */
    void (JNICALL *SetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, const jlong *buf); /* 212 */
    void *slot_213_221[9];
    void *(JNICALL *GetPrimitiveArrayCritical)(JNIEnv *env, jarray array, jboolean *isCopy);            /* 222 */
    void (JNICALL *ReleasePrimitiveArrayCritical)(JNIEnv *env, jarray array, void *carray, jint mode);  /* 223 */
    void *slot_224_227[4];
    jboolean (JNICALL *ExceptionCheck)(JNIEnv *env);                              /* 228 */
    void *slot_229_234[6];
};

#define CSIC_JNI_SLOT(name, index) \
    _Static_assert(offsetof(struct JNINativeInterface_, name) == (index) * sizeof(void *), #name " is not at table index " #index)
CSIC_JNI_SLOT(FindClass, 6);
CSIC_JNI_SLOT(ThrowNew, 14);
CSIC_JNI_SLOT(GetArrayLength, 171);
CSIC_JNI_SLOT(NewIntArray, 179);
CSIC_JNI_SLOT(NewLongArray, 180);
CSIC_JNI_SLOT(GetIntArrayElements, 187);
CSIC_JNI_SLOT(ReleaseIntArrayElements, 195);
CSIC_JNI_SLOT(GetIntArrayRegion, 203);
CSIC_JNI_SLOT(SetByteArrayRegion, 208);
CSIC_JNI_SLOT(SetIntArrayRegion, 211);
CSIC_JNI_SLOT(SetLongArrayRegion, 212);
CSIC_JNI_SLOT(GetPrimitiveArrayCritical, 222);
CSIC_JNI_SLOT(ReleasePrimitiveArrayCritical, 223);
CSIC_JNI_SLOT(ExceptionCheck, 228);

#endif
