/* TEST INFRASTRUCTURE ONLY -- NOT the JDK's jni.h.  This image has no JDK; this stub declares, with the JDK's names and
 * signatures, just the JNI types and JNIEnv functions that jni/rapid_mi355x_jni.c uses, so that the shim can be compiled
 * and type-checked against include/rapid_mi355x.h (tests/test_jni_shim.py).  With a JDK, build against the real header. */
#ifndef RAPID_TEST_JNI_STUB_H
#define RAPID_TEST_JNI_STUB_H
#include <stdint.h>
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef double jdouble;
typedef jint jsize;
struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_ABORT 2
#define JNIEXPORT
#define JNICALL
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv* env, const char* name);
    jint (*ThrowNew)(JNIEnv* env, jclass clazz, const char* msg);
    jsize (*GetArrayLength)(JNIEnv* env, jarray array);
    jintArray (*NewIntArray)(JNIEnv* env, jsize len);
    jlongArray (*NewLongArray)(JNIEnv* env, jsize len);
    void (*SetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, const jint* buf);
    void (*SetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, const jlong* buf);
    jint* (*GetIntArrayElements)(JNIEnv* env, jintArray array, jboolean* isCopy);
    jlong* (*GetLongArrayElements)(JNIEnv* env, jlongArray array, jboolean* isCopy);
    void (*ReleaseIntArrayElements)(JNIEnv* env, jintArray array, jint* elems, jint mode);
    void (*ReleaseLongArrayElements)(JNIEnv* env, jlongArray array, jlong* elems, jint mode);
    void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buf);
    jlong (*GetDirectBufferCapacity)(JNIEnv* env, jobject buf);
};
#endif
