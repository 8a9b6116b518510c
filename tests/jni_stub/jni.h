/* Minimal stand-in for the JDK's <jni.h>: ONLY the types and the JNIEnv function-table entries jni/mlease_jni.c uses, with
 * the signatures of the JNI specification (Java SE 8, chapter 4). It exists so that tests/test_jni_glue.py can type-check
 * the glue with `gcc -fsyntax-only` in an image without a JDK; it is never linked and never shipped. Member ORDER does not
 * follow the real table (irrelevant for a syntax check). */
#ifndef MLEASE_JNI_STUB_H
#define MLEASE_JNI_STUB_H
#include <stdarg.h>
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_COMMIT 1

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef uint16_t jchar;
typedef int16_t jshort;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jthrowable;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jbooleanArray;
typedef jarray jbyteArray;
typedef jarray jcharArray;
typedef jarray jshortArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jfloatArray;
typedef jarray jdoubleArray;
typedef jarray jobjectArray;
struct _jfieldID;
typedef struct _jfieldID *jfieldID;
struct _jmethodID;
typedef struct _jmethodID *jmethodID;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

#define JNI_ARRAY_FNS(T, J)                                                          \
    T *(JNICALL *Get##J##ArrayElements)(JNIEnv *env, T##Array array, jboolean *isCopy); \
    void(JNICALL *Release##J##ArrayElements)(JNIEnv *env, T##Array array, T *elems, jint mode); \
    T##Array(JNICALL *New##J##Array)(JNIEnv *env, jsize len);                        \
    void(JNICALL *Get##J##ArrayRegion)(JNIEnv *env, T##Array array, jsize start, jsize len, T *buf); \
    void(JNICALL *Set##J##ArrayRegion)(JNIEnv *env, T##Array array, jsize start, jsize len, const T *buf);

struct JNINativeInterface_ {
    jclass(JNICALL *FindClass)(JNIEnv *env, const char *name);
    jint(JNICALL *Throw)(JNIEnv *env, jthrowable obj);
    jint(JNICALL *ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    jobject(JNICALL *NewObject)(JNIEnv *env, jclass clazz, jmethodID methodID, ...);
    jclass(JNICALL *GetObjectClass)(JNIEnv *env, jobject obj);
    jmethodID(JNICALL *GetMethodID)(JNIEnv *env, jclass clazz, const char *name, const char *sig);
    jfieldID(JNICALL *GetFieldID)(JNIEnv *env, jclass clazz, const char *name, const char *sig);
    jlong(JNICALL *GetLongField)(JNIEnv *env, jobject obj, jfieldID fieldID);
    void(JNICALL *SetLongField)(JNIEnv *env, jobject obj, jfieldID fieldID, jlong val);
    void(JNICALL *SetDoubleField)(JNIEnv *env, jobject obj, jfieldID fieldID, jdouble val);
    jstring(JNICALL *NewStringUTF)(JNIEnv *env, const char *utf);
    const char *(JNICALL *GetStringUTFChars)(JNIEnv *env, jstring str, jboolean *isCopy);
    void(JNICALL *ReleaseStringUTFChars)(JNIEnv *env, jstring str, const char *chars);
    jsize(JNICALL *GetArrayLength)(JNIEnv *env, jarray array);
    jobject(JNICALL *GetObjectArrayElement)(JNIEnv *env, jobjectArray array, jsize index);
    jint(JNICALL *EnsureLocalCapacity)(JNIEnv *env, jint capacity);
    void(JNICALL *DeleteLocalRef)(JNIEnv *env, jobject localRef);
    JNI_ARRAY_FNS(jbyte, Byte)
    JNI_ARRAY_FNS(jint, Int)
    JNI_ARRAY_FNS(jlong, Long)
    JNI_ARRAY_FNS(jfloat, Float)
    JNI_ARRAY_FNS(jdouble, Double)
};
#endif
