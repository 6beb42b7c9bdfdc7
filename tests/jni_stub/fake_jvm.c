/*
 * fake_jvm.c -- TEST INFRASTRUCTURE: the JNIEnv of tests/jni_stub/jni.h over malloc, plus the few helpers
 * tests/test_jni_shim.py needs to make "Java arrays", read them back and look at the pending exception.
 * Checks a real JVM would not make for us: region accesses outside an array ABORT the test run (a write past a
 * Java array is exactly the bug class ADVICE r02 named), critical regions must be balanced, and an array must
 * not be region-copied while any critical region is open.
 */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { K_LONGS = 1, K_INTS, K_DOUBLES, K_STRING, K_CLASS };

struct _jobject {
    int kind;
    int64_t len;  /* elements (arrays) or bytes (strings, class names) */
    void *data;
    int pinned;
};

static char g_exc_class[128];
static char g_exc_msg[1024];
static int g_exc_pending;
static int g_critical_depth, g_critical_max, g_violations;

static size_t elem_of(int kind) { return kind == K_INTS ? 4 : 8; }

static void violation(const char *what)
{
    fprintf(stderr, "fake_jvm: JNI misuse: %s\n", what);
    ++g_violations;
}

static jclass f_FindClass(JNIEnv *env, const char *name)
{
    (void)env;
    struct _jobject *o = calloc(1, sizeof *o);
    o->kind = K_CLASS;
    o->data = strdup(name);
    return o; /* leaked on purpose: a handful per test */
}
static jint f_ThrowNew(JNIEnv *env, jclass c, const char *msg)
{
    (void)env;
    snprintf(g_exc_class, sizeof g_exc_class, "%s", (const char *)c->data);
    snprintf(g_exc_msg, sizeof g_exc_msg, "%s", msg ? msg : "");
    g_exc_pending = 1;
    return 0;
}
static jsize f_GetArrayLength(JNIEnv *env, jarray a)
{
    (void)env;
    return (jsize)a->len;
}
static int region_ok(jarray a, int kind, jsize start, jsize len)
{
    if (g_critical_depth) violation("JNI call inside a critical region");
    if (!a || a->kind != kind || start < 0 || len < 0 || (int64_t)start + len > a->len) {
        violation("array region out of bounds / wrong element type");
        return 0;
    }
    return 1;
}
#define REGION_FNS(NAME, KIND, T)                                                              \
    static void f_Get##NAME##ArrayRegion(JNIEnv *env, jarray a, jsize s, jsize n, T *buf)      \
    {                                                                                          \
        (void)env;                                                                             \
        if (region_ok(a, KIND, s, n)) memcpy(buf, (T *)a->data + s, (size_t)n * sizeof(T));    \
    }                                                                                          \
    static void f_Set##NAME##ArrayRegion(JNIEnv *env, jarray a, jsize s, jsize n, const T *buf)\
    {                                                                                          \
        (void)env;                                                                             \
        if (region_ok(a, KIND, s, n)) memcpy((T *)a->data + s, buf, (size_t)n * sizeof(T));    \
    }
REGION_FNS(Long, K_LONGS, jlong)
REGION_FNS(Int, K_INTS, jint)
REGION_FNS(Double, K_DOUBLES, jdouble)

static void *f_GetPrimitiveArrayCritical(JNIEnv *env, jarray a, jboolean *is_copy)
{
    (void)env;
    if (is_copy) *is_copy = 0;
    ++a->pinned;
    if (++g_critical_depth > g_critical_max) g_critical_max = g_critical_depth;
    return a->data;
}
static void f_ReleasePrimitiveArrayCritical(JNIEnv *env, jarray a, void *p, jint mode)
{
    (void)env;
    (void)mode;
    if (p != a->data || a->pinned <= 0) violation("release of an array that is not pinned");
    --a->pinned;
    --g_critical_depth;
}
static jstring f_NewStringUTF(JNIEnv *env, const char *utf)
{
    (void)env;
    struct _jobject *o = calloc(1, sizeof *o);
    o->kind = K_STRING;
    o->data = strdup(utf);
    o->len = (int64_t)strlen(utf);
    return o;
}
static const char *f_GetStringUTFChars(JNIEnv *env, jstring s, jboolean *is_copy)
{
    (void)env;
    if (is_copy) *is_copy = 0;
    return (const char *)s->data;
}
static void f_ReleaseStringUTFChars(JNIEnv *env, jstring s, const char *c)
{
    (void)env;
    if (c != (const char *)s->data) violation("ReleaseStringUTFChars with a foreign pointer");
}

static const struct JNINativeInterface_ g_table = {
    f_FindClass, f_ThrowNew, f_GetArrayLength, f_GetLongArrayRegion, f_GetIntArrayRegion, f_GetDoubleArrayRegion,
    f_SetLongArrayRegion, f_SetIntArrayRegion, f_SetDoubleArrayRegion, f_GetPrimitiveArrayCritical,
    f_ReleasePrimitiveArrayCritical, f_NewStringUTF, f_GetStringUTFChars, f_ReleaseStringUTFChars};
static JNIEnv g_env = &g_table;

/* ---- what the Python test calls ---- */
JNIEXPORT JNIEnv *fake_env(void) { return &g_env; }

JNIEXPORT jarray fake_new_array(int kind, int64_t len, const void *init)
{
    struct _jobject *o = calloc(1, sizeof *o);
    o->kind = kind;
    o->len = len;
    /* exact size + a canary after the last element: a write past the array is caught by fake_array_intact() */
    o->data = malloc((size_t)len * elem_of(kind) + 8);
    if (init) memcpy(o->data, init, (size_t)len * elem_of(kind));
    else memset(o->data, 0xEE, (size_t)len * elem_of(kind));
    memset((char *)o->data + (size_t)len * elem_of(kind), 0xA5, 8);
    return o;
}
JNIEXPORT void *fake_array_data(jarray a) { return a->data; }
JNIEXPORT int fake_array_intact(jarray a)
{
    const unsigned char *c = (const unsigned char *)a->data + (size_t)a->len * elem_of(a->kind);
    for (int i = 0; i < 8; ++i)
        if (c[i] != 0xA5) return 0;
    return a->pinned == 0;
}
JNIEXPORT void fake_free(jobject o)
{
    if (o) {
        free(o->data);
        free(o);
    }
}
JNIEXPORT jstring fake_new_string(const char *s) { return f_NewStringUTF(&g_env, s); }
JNIEXPORT const char *fake_string_chars(jstring s) { return (const char *)s->data; }
JNIEXPORT int fake_exception_pending(void) { return g_exc_pending; }
JNIEXPORT const char *fake_exception_class(void) { return g_exc_class; }
JNIEXPORT const char *fake_exception_message(void) { return g_exc_msg; }
JNIEXPORT void fake_exception_clear(void) { g_exc_pending = 0; }
JNIEXPORT int fake_critical_depth(void) { return g_critical_depth; }
JNIEXPORT int fake_critical_max(void)
{
    int m = g_critical_max;
    g_critical_max = 0;
    return m;
}
JNIEXPORT int fake_violations(void) { return g_violations; }
