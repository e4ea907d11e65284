/* TEST INFRASTRUCTURE ONLY: a fake JNIEnv (arrays and direct buffers as plain C blocks, exceptions recorded) that
 * drives jni/rapid_mi355x_jni.c the way the Java facades would -- the host-only natives of NativeFastPaxos and the wire
 * decoders, i.e. everything that runs without a GPU: five nodes propose conflicting values, one recovery timer fires,
 * and the nodes exchange nothing but the serialized RapidRequests the shim hands out.  Exit code 0 = all five decided
 * the same value and an illegal call raised the right Java exception class. */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rapid_mi355x.h"

struct _jobject {
    int kind;  /* 0 int[], 1 long[], 2 direct buffer, 3 class */
    jsize len;
    void* data;
    const char* name;
};
static char last_exception[128];

static jclass FindClass(JNIEnv* env, const char* name) {
    (void)env;
    jobject o = calloc(1, sizeof *o);
    o->kind = 3;
    o->name = name;
    return o;
}
static jint ThrowNew(JNIEnv* env, jclass c, const char* msg) {
    (void)env; (void)msg;
    snprintf(last_exception, sizeof last_exception, "%s", c->name);
    return 0;
}
static jsize GetArrayLength(JNIEnv* env, jarray a) { (void)env; return a->len; }
static jobject new_block(int kind, jsize len, size_t elem) {
    jobject o = calloc(1, sizeof *o);
    o->kind = kind;
    o->len = len;
    o->data = calloc((size_t)(len > 0 ? len : 1), elem);
    return o;
}
static jintArray NewIntArray(JNIEnv* env, jsize len) { (void)env; return new_block(0, len, sizeof(jint)); }
static jlongArray NewLongArray(JNIEnv* env, jsize len) { (void)env; return new_block(1, len, sizeof(jlong)); }
static void SetIntArrayRegion(JNIEnv* env, jintArray a, jsize s, jsize n, const jint* b) { (void)env; memcpy((jint*)a->data + s, b, sizeof(jint) * (size_t)n); }
static void SetLongArrayRegion(JNIEnv* env, jlongArray a, jsize s, jsize n, const jlong* b) { (void)env; memcpy((jlong*)a->data + s, b, sizeof(jlong) * (size_t)n); }
static jint* GetIntArrayElements(JNIEnv* env, jintArray a, jboolean* c) { (void)env; if (c) *c = JNI_FALSE; return a->data; }
static jlong* GetLongArrayElements(JNIEnv* env, jlongArray a, jboolean* c) { (void)env; if (c) *c = JNI_FALSE; return a->data; }
static void ReleaseIntArrayElements(JNIEnv* env, jintArray a, jint* e, jint m) { (void)env; (void)a; (void)e; (void)m; }
static void ReleaseLongArrayElements(JNIEnv* env, jlongArray a, jlong* e, jint m) { (void)env; (void)a; (void)e; (void)m; }
static void* GetDirectBufferAddress(JNIEnv* env, jobject b) { (void)env; return b->data; }
static jlong GetDirectBufferCapacity(JNIEnv* env, jobject b) { (void)env; return b->len; }

static const struct JNINativeInterface_ table = {FindClass, ThrowNew, GetArrayLength, NewIntArray, NewLongArray, SetIntArrayRegion,
                                                 SetLongArrayRegion, GetIntArrayElements, GetLongArrayElements, ReleaseIntArrayElements,
                                                 ReleaseLongArrayElements, GetDirectBufferAddress, GetDirectBufferCapacity};

/* the natives under test (jni/rapid_mi355x_jni.c) */
jlong Java_com_vrg_rapid_NativeFastPaxos_create(JNIEnv*, jclass, jint, jint, jlong, jint);
void Java_com_vrg_rapid_NativeFastPaxos_close(JNIEnv*, jobject, jlong);
void Java_com_vrg_rapid_NativeFastPaxos_propose(JNIEnv*, jobject, jlong, jintArray);
void Java_com_vrg_rapid_NativeFastPaxos_handle(JNIEnv*, jobject, jlong, jlong, jint, jobject, jlong, jlong);
void Java_com_vrg_rapid_NativeFastPaxos_startClassicRound(JNIEnv*, jobject, jlong);
jint Java_com_vrg_rapid_NativeFastPaxos_poll(JNIEnv*, jobject, jlong, jlong, jobject, jlongArray);
jintArray Java_com_vrg_rapid_NativeFastPaxos_decision(JNIEnv*, jobject, jlong);
jlong Java_com_vrg_rapid_NativeFastPaxos_fallbackDelayMs(JNIEnv*, jclass, jint, jlong, jdouble);
jint Java_com_vrg_rapid_NativeCutEngine_decodeRequest(JNIEnv*, jobject, jobject, jint, jlongArray);

#define N 5
struct packet { int len; unsigned char bytes[512]; };
static struct packet queue[N][256];
static int q_head[N], q_tail[N];

int main(void) {
    JNIEnv env_obj = &table;
    JNIEnv* env = &env_obj;
    const char* hosts = "n0n1n2n3n4";
    int32_t off[N + 1] = {0, 2, 4, 6, 8, 10}, ports[N] = {9000, 9001, 9002, 9003, 9004};
    rapid_endpoint_map* map = NULL;
    if (rapid_endpoint_map_create((const uint8_t*)hosts, off, ports, N, &map) != RAPID_OK) return 10;
    jlong node[N];
    for (int i = 0; i < N; ++i) node[i] = Java_com_vrg_rapid_NativeFastPaxos_create(env, NULL, i, 100 + i, 77, N);
    jobject request = new_block(2, 512, 1);
    jlongArray len1 = NewLongArray(env, 1), offlen = NewLongArray(env, 2);
#define FLUSH(i)                                                                                                 \
    for (;;) {                                                                                                   \
        const jint dest = Java_com_vrg_rapid_NativeFastPaxos_poll(env, NULL, node[i], (jlong)(intptr_t)map, request, len1); \
        if (dest == -2) break;                                                                                   \
        const int len = (int)((jlong*)len1->data)[0];                                                            \
        for (int d = 0; d < N; ++d)                                                                              \
            if (dest == -1 || dest == d) {                                                                       \
                struct packet* p = &queue[d][q_tail[d]++ % 256];                                                \
                p->len = len;                                                                                    \
                memcpy(p->bytes, request->data, (size_t)len);                                                    \
            }                                                                                                    \
    }
    for (int i = 0; i < N; ++i) {  /* two camps: no fast quorum (needs 4 of 5 identical votes) */
        jintArray prop = NewIntArray(env, 2);
        ((jint*)prop->data)[0] = i % 2;
        ((jint*)prop->data)[1] = 4;
        Java_com_vrg_rapid_NativeFastPaxos_propose(env, NULL, node[i], prop);
        FLUSH(i);
    }
    if (Java_com_vrg_rapid_NativeFastPaxos_fallbackDelayMs(env, NULL, N, 1000, 0.0) != 1000) return 11;
    Java_com_vrg_rapid_NativeFastPaxos_startClassicRound(env, NULL, node[3]);  /* its recovery timer fired first */
    FLUSH(3);
    unsigned rng = 12345;
    for (;;) {
        int pending[N], np = 0;
        for (int i = 0; i < N; ++i)
            if (q_head[i] != q_tail[i]) pending[np++] = i;
        if (!np) break;
        rng = rng * 1103515245u + 12345u;
        const int i = pending[(rng >> 16) % (unsigned)np];
        struct packet* p = &queue[i][q_head[i]++ % 256];
        jobject wire = new_block(2, p->len, 1);
        memcpy(wire->data, p->bytes, (size_t)p->len);
        const jint kind = Java_com_vrg_rapid_NativeCutEngine_decodeRequest(env, NULL, wire, p->len, offlen);
        if (kind < 5 || kind > 9) return 12;
        Java_com_vrg_rapid_NativeFastPaxos_handle(env, NULL, node[i], (jlong)(intptr_t)map, kind, wire, ((jlong*)offlen->data)[0],
                                                  ((jlong*)offlen->data)[1]);
        if (last_exception[0]) return 13;
        FLUSH(i);
    }
    jint first[2] = {-1, -1};
    for (int i = 0; i < N; ++i) {
        jintArray d = Java_com_vrg_rapid_NativeFastPaxos_decision(env, NULL, node[i]);
        if (!d || d->len != 2) return 14;
        if (i == 0) memcpy(first, d->data, sizeof first);
        if (memcmp(first, d->data, sizeof first) != 0) return 15;
    }
    if (first[1] != 4 || (first[0] != 0 && first[0] != 1)) return 16;
    /* error mapping: a consensus instance for an empty membership is an IllegalArgumentException */
    (void)Java_com_vrg_rapid_NativeFastPaxos_create(env, NULL, 0, 0, 1, 0);
    if (strcmp(last_exception, "java/lang/IllegalArgumentException") != 0) return 17;
    for (int i = 0; i < N; ++i) Java_com_vrg_rapid_NativeFastPaxos_close(env, NULL, node[i]);
    rapid_endpoint_map_destroy(map);
    printf("decided (%d, %d) at all %d nodes through the JNI shim\n", first[0], first[1], N);
    return 0;
}
