/* fake_env.c -- a miniature JNIEnv for tests/test_jni_glue.py: enough of the JNI function table (the entries jni/mlease_jni.c uses,
 * declared in this directory's stub jni.h) to RUN the glue without a JVM. Arrays are heap blocks with a length, objects carry a
 * `handle` long field, exceptions are recorded (class name + message) instead of thrown. Test infrastructure only: never shipped,
 * never linked into the product. Built together with jni/mlease_jni.c into tests/jni_stub/libjni_fake.so by the test. */
#define _POSIX_C_SOURCE 200809L
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct _jobject {
    int kind;               /* 0 object, 1 primitive array, 2 object array, 3 class, 4 string */
    jsize len;
    size_t esize;
    void *data;
    jobject *elems;
    char name[96];
    jlong handle;
    double dfields[32];
    jlong lfields[32];
};
struct _jfieldID { char name[48]; };
struct _jmethodID { char name[48]; };

static char g_exc_class[128], g_exc_msg[700];
static int g_pending, g_pins;

static jobject new_obj(int kind) { jobject o = (jobject)calloc(1, sizeof(struct _jobject)); o->kind = kind; return o; }

/* ---- exported to the python test ---------------------------------------------------------------------------------------------- */
jobject fake_new_array(size_t esize, jsize len, const void *init)
{
    jobject o = new_obj(1);
    o->len = len; o->esize = esize;
    o->data = calloc((size_t)(len > 0 ? len : 1), esize);
    if (init && len > 0) memcpy(o->data, init, (size_t)len * esize);
    return o;
}
jobject fake_new_object_array(jsize len) { jobject o = new_obj(2); o->len = len; o->elems = (jobject *)calloc((size_t)(len > 0 ? len : 1), sizeof(jobject)); return o; }
void fake_set_element(jobject arr, jsize i, jobject v) { arr->elems[i] = v; }
void *fake_array_data(jobject a) { return a->data; }
jsize fake_array_len(jobject a) { return a->len; }
jobject fake_new_self(jlong handle) { jobject o = new_obj(0); o->handle = handle; return o; }
double fake_object_double(jobject o, int i) { return o->dfields[i]; }
jlong fake_object_long(jobject o, int i) { return o->lfields[i]; }
const char *fake_exception_class(void) { return g_pending ? g_exc_class : ""; }
const char *fake_exception_message(void) { return g_pending ? g_exc_msg : ""; }
void fake_clear(void) { g_pending = 0; g_exc_class[0] = g_exc_msg[0] = 0; }
int fake_outstanding_pins(void) { return g_pins; }

/* ---- the function table ------------------------------------------------------------------------------------------------------- */
static jclass JNICALL FindClass(JNIEnv *env, const char *name) { (void)env; jobject c = new_obj(3); snprintf(c->name, sizeof c->name, "%s", name); return c; }
static jint JNICALL ThrowNew(JNIEnv *env, jclass clazz, const char *msg)
{
    (void)env;
    if (!g_pending) { snprintf(g_exc_class, sizeof g_exc_class, "%s", clazz->name); snprintf(g_exc_msg, sizeof g_exc_msg, "%s", msg ? msg : ""); }
    g_pending = 1;
    return 0;
}
static jint JNICALL Throw(JNIEnv *env, jthrowable obj)
{
    (void)env;
    if (!g_pending) { snprintf(g_exc_class, sizeof g_exc_class, "%s", obj->name); snprintf(g_exc_msg, sizeof g_exc_msg, "%s", (const char *)obj->data ? (const char *)obj->data : ""); }
    g_pending = 1;
    return 0;
}
static jobject JNICALL NewObject(JNIEnv *env, jclass clazz, jmethodID mid, ...)
{
    (void)env; (void)mid;
    jobject o = new_obj(0);
    snprintf(o->name, sizeof o->name, "%s", clazz->name);
    /* exception constructors used by the glue: (String) and (String, Throwable): keep the message for Throw() */
    va_list ap;
    va_start(ap, mid);
    if (strstr(clazz->name, "Exception")) { jobject s = va_arg(ap, jobject); if (s && s->kind == 4) o->data = strdup((const char *)s->data); }
    va_end(ap);
    return o;
}
static jclass JNICALL GetObjectClass(JNIEnv *env, jobject obj) { (void)env; jobject c = new_obj(3); snprintf(c->name, sizeof c->name, "%s", obj->name); return c; }
static jmethodID JNICALL GetMethodID(JNIEnv *env, jclass c, const char *name, const char *sig) { (void)env; (void)c; (void)sig; jmethodID m = (jmethodID)calloc(1, sizeof(struct _jmethodID)); snprintf(m->name, sizeof m->name, "%s", name); return m; }
static jfieldID JNICALL GetFieldID(JNIEnv *env, jclass c, const char *name, const char *sig) { (void)env; (void)c; (void)sig; jfieldID f = (jfieldID)calloc(1, sizeof(struct _jfieldID)); snprintf(f->name, sizeof f->name, "%s", name); return f; }
static jlong JNICALL GetLongField(JNIEnv *env, jobject obj, jfieldID f) { (void)env; (void)f; return obj->handle; }
/* Stats fields in declaration order of jni/mlease_jni.c: doubles and longs land in slots by first use */
static int slot_of(const char *name)
{
    static char names[32][48];
    static int n;
    for (int i = 0; i < n; i++) if (!strcmp(names[i], name)) return i;
    if (n >= 32) { fprintf(stderr, "fake JNI: more than 32 Stats fields\n"); abort(); }
    snprintf(names[n], 48, "%s", name);
    return n++;
}
static void JNICALL SetLongField(JNIEnv *env, jobject obj, jfieldID f, jlong v) { (void)env; obj->lfields[slot_of(f->name)] = v; }
static void JNICALL SetDoubleField(JNIEnv *env, jobject obj, jfieldID f, jdouble v) { (void)env; obj->dfields[slot_of(f->name)] = v; }
static jstring JNICALL NewStringUTF(JNIEnv *env, const char *utf) { (void)env; jobject s = new_obj(4); s->data = strdup(utf ? utf : ""); return s; }
static const char *JNICALL GetStringUTFChars(JNIEnv *env, jstring s, jboolean *isCopy) { (void)env; if (isCopy) *isCopy = 0; g_pins++; return (const char *)s->data; }
static void JNICALL ReleaseStringUTFChars(JNIEnv *env, jstring s, const char *c) { (void)env; (void)s; (void)c; g_pins--; }
jobject fake_new_string(const char *utf) { return NewStringUTF(NULL, utf); }
const char *fake_string_chars(jobject s) { return s ? (const char *)s->data : ""; }
static jsize JNICALL GetArrayLength(JNIEnv *env, jarray a) { (void)env; return a->len; }
static jobject JNICALL GetObjectArrayElement(JNIEnv *env, jobjectArray a, jsize i) { (void)env; return (i >= 0 && i < a->len) ? a->elems[i] : NULL; }
static jint JNICALL EnsureLocalCapacity(JNIEnv *env, jint cap) { (void)env; (void)cap; return 0; }
static void JNICALL DeleteLocalRef(JNIEnv *env, jobject ref) { (void)env; (void)ref; }

#define FAKE_ARRAY_FNS(T, J)                                                                                                         \
    static T *JNICALL Get##J##ArrayElements(JNIEnv *env, T##Array a, jboolean *isCopy) { (void)env; if (isCopy) *isCopy = 0; g_pins++; return (T *)a->data; } \
    static void JNICALL Release##J##ArrayElements(JNIEnv *env, T##Array a, T *e, jint mode) { (void)env; (void)a; (void)e; (void)mode; g_pins--; }  \
    static T##Array JNICALL New##J##Array(JNIEnv *env, jsize len) { (void)env; return fake_new_array(sizeof(T), len, NULL); }          \
    static void JNICALL Get##J##ArrayRegion(JNIEnv *env, T##Array a, jsize s, jsize l, T *buf)                                         \
    { (void)env; if (s < 0 || l < 0 || s + l > a->len) { fprintf(stderr, "fake JNI: Get" #J "ArrayRegion out of bounds\n"); abort(); } memcpy(buf, (T *)a->data + s, (size_t)l * sizeof(T)); } \
    static void JNICALL Set##J##ArrayRegion(JNIEnv *env, T##Array a, jsize s, jsize l, const T *buf)                                   \
    { (void)env; if (s < 0 || l < 0 || s + l > a->len) { fprintf(stderr, "fake JNI: Set" #J "ArrayRegion out of bounds\n"); abort(); } memcpy((T *)a->data + s, buf, (size_t)l * sizeof(T)); }
FAKE_ARRAY_FNS(jbyte, Byte)
FAKE_ARRAY_FNS(jint, Int)
FAKE_ARRAY_FNS(jlong, Long)
FAKE_ARRAY_FNS(jfloat, Float)
FAKE_ARRAY_FNS(jdouble, Double)

#define TABLE_ARRAY_FNS(J) .Get##J##ArrayElements = Get##J##ArrayElements, .Release##J##ArrayElements = Release##J##ArrayElements, \
    .New##J##Array = New##J##Array, .Get##J##ArrayRegion = Get##J##ArrayRegion, .Set##J##ArrayRegion = Set##J##ArrayRegion
static const struct JNINativeInterface_ g_table = {
    .FindClass = FindClass, .Throw = Throw, .ThrowNew = ThrowNew, .NewObject = NewObject, .GetObjectClass = GetObjectClass,
    .GetMethodID = GetMethodID, .GetFieldID = GetFieldID, .GetLongField = GetLongField, .SetLongField = SetLongField,
    .SetDoubleField = SetDoubleField, .NewStringUTF = NewStringUTF, .GetStringUTFChars = GetStringUTFChars, .ReleaseStringUTFChars = ReleaseStringUTFChars, .GetArrayLength = GetArrayLength,
    .GetObjectArrayElement = GetObjectArrayElement, .EnsureLocalCapacity = EnsureLocalCapacity, .DeleteLocalRef = DeleteLocalRef,
    TABLE_ARRAY_FNS(Byte), TABLE_ARRAY_FNS(Int), TABLE_ARRAY_FNS(Long), TABLE_ARRAY_FNS(Float), TABLE_ARRAY_FNS(Double),
};
static JNIEnv g_env = &g_table;
JNIEnv *fake_env(void) { return &g_env; }
