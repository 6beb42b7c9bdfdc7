/*
 * tests/jni_stub/jni.h -- TEST INFRASTRUCTURE, not a JDK header.
 *
 * The build container has no JDK, so jni/locrec_jni.c could never be compiled or run here.  This is a
 * declaration-level stand-in for <jni.h>, written from the JNI specification's names and signatures and holding
 * ONLY the types and JNIEnv functions the shim uses; fake_jvm.c implements them over malloc.  It lets
 * tests/test_jni_shim.py compile the shim with -Wall -Werror and drive every native method on the GPU box
 * (length checks, exception mapping, copy-in / copy-out, the handle cache) without a JVM.  The layout of the
 * function table is NOT the JVM's: a shim built against this header must never be loaded into a real JVM
 * (jni/Makefile uses $JAVA_HOME/include).
 */
#ifndef LOCREC_TEST_JNI_STUB_H
#define LOCREC_TEST_JNI_STUB_H

#include <stdint.h>

#define LOCREC_JNI_STUB 1
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_COMMIT 1

typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef uint8_t jboolean;
typedef jint jsize;

struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jlongArray;
typedef jarray jintArray;
typedef jarray jdoubleArray;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *env, const char *name);
    jint (*ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    jsize (*GetArrayLength)(JNIEnv *env, jarray array);
    void (*GetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, jlong *buf);
    void (*GetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, jint *buf);
    void (*GetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize len, jdouble *buf);
    void (*SetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, const jlong *buf);
    void (*SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf);
    void (*SetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize len, const jdouble *buf);
    void *(*GetPrimitiveArrayCritical)(JNIEnv *env, jarray array, jboolean *isCopy);
    void (*ReleasePrimitiveArrayCritical)(JNIEnv *env, jarray array, void *carray, jint mode);
    jstring (*NewStringUTF)(JNIEnv *env, const char *utf);
    const char *(*GetStringUTFChars)(JNIEnv *env, jstring str, jboolean *isCopy);
    void (*ReleaseStringUTFChars)(JNIEnv *env, jstring str, const char *chars);
};

#endif
