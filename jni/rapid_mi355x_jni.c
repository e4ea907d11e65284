/* JNI shim between the reference's Java and librapid_mi355x.so: unpack the arguments, call
 * the C ABI of include/rapid_mi355x.h, map error codes onto the exceptions the Java classes throw.  One function per
 * native method of the two package-private facades a maintainer adds:
 *     com.vrg.rapid.NativeCutEngine   (MembershipView, MultiNodeCutDetector, population replay, wire ingest)
 *     com.vrg.rapid.NativeFastPaxos   (FastPaxos + Paxos of one configuration)
 * Inputs are borrowed for the duration of the call (direct buffers, primitive arrays), outputs are copied into fresh
 * Java arrays, nothing calls back into the JVM, there is no global state.  Every native call happens on the reference's
 * single protocol executor thread (SharedResources.java:53): one caller at a time per handle, as the library requires.
 *
 * Build (needs a JDK, which this repository's build image does not have -- tests compile it against a stub jni.h):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jni/rapid_mi355x_jni.c \
 *       -Lrapid_amd -lrapid_mi355x -o librapid_mi355x_jni.so */
#include <jni.h>
#include <stdlib.h>
#include <string.h>

#include "rapid_mi355x.h"

#define MAX_LIST 4096

static void throw_for(JNIEnv* env, const char* detail, int rc) {
    const char* cls = "java/lang/IllegalStateException";
    switch (rc) {
        case RAPID_EINVAL: cls = "java/lang/IllegalArgumentException"; break;                               /* MultiNodeCutDetector.java:52-55 */
        case RAPID_ENODE_EXISTS: cls = "com/vrg/rapid/MembershipView$NodeAlreadyInRingException"; break;    /* MembershipView.java:502-506 */
        case RAPID_ENODE_MISSING: cls = "com/vrg/rapid/MembershipView$NodeNotInRingException"; break;       /* :508-512 */
        case RAPID_EUUID_SEEN: cls = "com/vrg/rapid/MembershipView$UUIDAlreadySeenException"; break;        /* :514-519 */
        default: break;
    }
    (*env)->ThrowNew(env, (*env)->FindClass(env, cls), detail ? detail : "");
}
#define ENGINE(h) ((rapid_engine*)(intptr_t)(h))
#define CHECK(h, rc)                                                 \
    do {                                                             \
        const int rc_ = (rc);                                        \
        if (rc_ != RAPID_OK) throw_for(env, rapid_last_error(ENGINE(h)), rc_); \
    } while (0)

static void throw_iae(JNIEnv* env, const char* what) {
    (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), what);
}
/* [off, off + len) of a direct buffer, or NULL after throwing IllegalArgumentException: a bug in the facade becomes a Java
 * exception, not a read or write outside the buffer in native code */
static uint8_t* direct_region(JNIEnv* env, jobject buf, jlong off, jlong len, const char* what) {
    uint8_t* base = buf ? (uint8_t*)(*env)->GetDirectBufferAddress(env, buf) : NULL;
    const jlong cap = buf ? (*env)->GetDirectBufferCapacity(env, buf) : -1;
    if (!base || cap < 0 || off < 0 || len < 0 || off > cap || len > cap - off) {
        throw_iae(env, what);
        return NULL;
    }
    return base + off;
}

static jintArray to_java(JNIEnv* env, const int32_t* v, int32_t n) {
    jintArray a = (*env)->NewIntArray(env, n);
    if (a) (*env)->SetIntArrayRegion(env, a, 0, n, (const jint*)v);
    return a;
}

/* ---------------------------------------------------------------- NativeCutEngine: lifecycle + MembershipView */
JNIEXPORT jlong JNICALL Java_com_vrg_rapid_NativeCutEngine_create(JNIEnv* env, jclass cls, jint nMax, jint K, jint H, jint L,
                                                                  jint deviceId) {
    (void)cls;
    rapid_engine_config cfg = {nMax, K, H, L, deviceId, 0};
    rapid_engine* h = NULL;
    const int rc = rapid_engine_create(&cfg, &h);
    if (rc != RAPID_OK) throw_for(env, "rapid_engine_create", rc);  /* K/H/L as the MultiNodeCutDetector constructor checks them */
    return (jlong)(intptr_t)h;
}

/* once after create: a known-answer view + round on h's device; RAPID_EDEVICE -> IllegalStateException with the first difference */
JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeCutEngine_selfTest(JNIEnv* env, jobject self, jlong h) {
    (void)self;
    CHECK(h, rapid_engine_self_test(ENGINE(h)));
}

JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeCutEngine_close(JNIEnv* env, jobject self, jlong h) {
    (void)env; (void)self;
    rapid_engine_destroy(ENGINE(h));
}

/* hostnames: direct buffer with all hostname bytes back to back; hostOff has n + 1 entries */
JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeCutEngine_viewBuild(JNIEnv* env, jobject self, jlong h, jobject hostnames,
                                                                    jintArray hostOff, jintArray ports, jlongArray idHi,
                                                                    jlongArray idLo, jintArray members) {
    (void)self;
    const jsize n = (*env)->GetArrayLength(env, ports), nm = (*env)->GetArrayLength(env, members);
    jint* off = (*env)->GetIntArrayElements(env, hostOff, NULL);
    jint* po = (*env)->GetIntArrayElements(env, ports, NULL);
    jlong* hi = (*env)->GetLongArrayElements(env, idHi, NULL);
    jlong* lo = (*env)->GetLongArrayElements(env, idLo, NULL);
    jint* mem = (*env)->GetIntArrayElements(env, members, NULL);
    const int rc = rapid_view_build(ENGINE(h), (const uint8_t*)(*env)->GetDirectBufferAddress(env, hostnames), off, po,
                                    (const int64_t*)hi, (const int64_t*)lo, n, mem, nm, NULL, NULL, 0);
    (*env)->ReleaseIntArrayElements(env, members, mem, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, idLo, lo, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, idHi, hi, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, ports, po, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, hostOff, off, JNI_ABORT);
    CHECK(h, rc);
}

JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeCutEngine_ringAdd(JNIEnv* env, jobject self, jlong h, jint node, jlong idHi,
                                                                  jlong idLo) {
    (void)self;
    CHECK(h, rapid_view_ring_add(ENGINE(h), node, idHi, idLo));
}

JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeCutEngine_ringDelete(JNIEnv* env, jobject self, jlong h, jint node) {
    (void)self;
    CHECK(h, rapid_view_ring_delete(ENGINE(h), node));
}

typedef int (*list_fn)(rapid_engine*, int32_t, int32_t*, int32_t, int32_t*);
static jintArray node_list(JNIEnv* env, jlong h, jint node, list_fn fn) {
    int32_t out[RAPID_MAX_K], n = 0;
    const int rc = fn(ENGINE(h), node, out, RAPID_MAX_K, &n);
    if (rc != RAPID_OK) {
        throw_for(env, rapid_last_error(ENGINE(h)), rc);
        return NULL;
    }
    return to_java(env, out, n);
}
JNIEXPORT jintArray JNICALL Java_com_vrg_rapid_NativeCutEngine_observersOf(JNIEnv* env, jobject self, jlong h, jint node) {
    (void)self;
    return node_list(env, h, node, rapid_view_observers);
}
JNIEXPORT jintArray JNICALL Java_com_vrg_rapid_NativeCutEngine_subjectsOf(JNIEnv* env, jobject self, jlong h, jint node) {
    (void)self;
    return node_list(env, h, node, rapid_view_subjects);
}
JNIEXPORT jintArray JNICALL Java_com_vrg_rapid_NativeCutEngine_expectedObserversOf(JNIEnv* env, jobject self, jlong h, jint node) {
    (void)self;
    return node_list(env, h, node, rapid_view_expected_observers);
}

/* int[] q4AtRisk(long h, int[] hotSubjects): the members among them whose observers, memoised by the reference when they were
 * first in flux (MembershipView.cachedObservers), differ from today's -- empty: quirk Q4 cannot fire at any receiver this round */
JNIEXPORT jintArray JNICALL Java_com_vrg_rapid_NativeCutEngine_q4AtRisk(JNIEnv* env, jobject self, jlong h, jintArray hot) {
    (void)self;
    const jsize n = (*env)->GetArrayLength(env, hot);
    jint* in = (*env)->GetIntArrayElements(env, hot, NULL);
    if (in == NULL) return NULL;
    jintArray result = NULL;
    int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    if (out != NULL) {
        int32_t m = 0;
        const int rc = rapid_view_q4_at_risk(ENGINE(h), (const int32_t*)in, (int32_t)n, out, (int32_t)n, &m);
        if (rc != RAPID_OK) throw_for(env, rapid_last_error(ENGINE(h)), rc);
        else result = to_java(env, out, m);
        free(out);
    }
    (*env)->ReleaseIntArrayElements(env, hot, in, JNI_ABORT);
    return result;
}

JNIEXPORT jlong JNICALL Java_com_vrg_rapid_NativeCutEngine_configurationId(JNIEnv* env, jobject self, jlong h) {
    (void)self;
    int64_t id = 0;
    CHECK(h, rapid_view_config_id(ENGINE(h), &id));
    return id;
}

/* ---------------------------------------------------------------- NativeCutEngine: MultiNodeCutDetector */
JNIEXPORT jlong JNICALL Java_com_vrg_rapid_NativeCutEngine_cdCreate(JNIEnv* env, jobject self, jlong h, jint K, jint H, jint L) {
    (void)self;
    rapid_cd* cd = NULL;
    CHECK(h, rapid_cd_create(ENGINE(h), K, H, L, &cd));
    return (jlong)(intptr_t)cd;
}

/* packedAlerts: direct buffer of n 20-byte rapid_alert_record; returns the proposed endpoints of all n alerts */
JNIEXPORT jintArray JNICALL Java_com_vrg_rapid_NativeCutEngine_cdAggregate(JNIEnv* env, jobject self, jlong cd, jobject packedAlerts,
                                                                           jint n) {
    (void)self;
    const rapid_alert_record* recs = (const rapid_alert_record*)(*env)->GetDirectBufferAddress(env, packedAlerts);
    int32_t out[MAX_LIST], n_out = 0;
    int32_t* counts = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    const int rc = counts ? rapid_cd_aggregate((rapid_cd*)(intptr_t)cd, recs, n, out, MAX_LIST, counts, &n_out) : RAPID_ECAPACITY;
    free(counts);
    if (rc != RAPID_OK) {
        throw_for(env, "rapid_cd_aggregate", rc);
        return NULL;
    }
    return to_java(env, out, n_out);
}

JNIEXPORT jintArray JNICALL Java_com_vrg_rapid_NativeCutEngine_cdInvalidate(JNIEnv* env, jobject self, jlong cd) {
    (void)self;
    int32_t out[MAX_LIST], n_out = 0;
    const int rc = rapid_cd_invalidate((rapid_cd*)(intptr_t)cd, out, MAX_LIST, &n_out);
    if (rc != RAPID_OK) {
        throw_for(env, "rapid_cd_invalidate", rc);
        return NULL;
    }
    return to_java(env, out, n_out);
}

JNIEXPORT jint JNICALL Java_com_vrg_rapid_NativeCutEngine_cdNumProposals(JNIEnv* env, jobject self, jlong cd) {
    (void)self;
    int32_t n = 0;
    const int rc = rapid_cd_num_proposals((rapid_cd*)(intptr_t)cd, &n);
    if (rc != RAPID_OK) throw_for(env, "rapid_cd_num_proposals", rc);
    return n;
}

JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeCutEngine_cdClear(JNIEnv* env, jobject self, jlong cd) {
    (void)self;
    const int rc = rapid_cd_clear((rapid_cd*)(intptr_t)cd);
    if (rc != RAPID_OK) throw_for(env, "rapid_cd_clear", rc);
}

/* ---------------------------------------------------------------- NativeCutEngine: population replay */
JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeCutEngine_simLoadStreams(JNIEnv* env, jobject self, jlong h, jobject records,
                                                                         jlongArray recOff) {
    (void)self;
    const jsize n = (*env)->GetArrayLength(env, recOff);
    jlong* off = (*env)->GetLongArrayElements(env, recOff, NULL);
    const int rc = rapid_sim_load_streams(ENGINE(h), (const rapid_alert_record*)(*env)->GetDirectBufferAddress(env, records),
                                          (const int64_t*)off, n - 1);
    (*env)->ReleaseLongArrayElements(env, recOff, off, JNI_ABORT);
    CHECK(h, rc);
}

JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeCutEngine_simSetAlertSet(JNIEnv* env, jobject self, jlong h, jobject alerts, jlong n) {
    (void)self;
    CHECK(h, rapid_sim_set_alert_set(ENGINE(h), (const rapid_alert_record*)(*env)->GetDirectBufferAddress(env, alerts), n));
}

/* void simSetAlertSetDevice(long h, long dAlerts, long alertsBytes, long n): the round's distinct alerts already in device memory
 * (alertsBytes readable bytes covering them), read in place */
JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeCutEngine_simSetAlertSetDevice(JNIEnv* env, jobject self, jlong h, jlong dAlerts, jlong alertsBytes,
                                                                               jlong n) {
    (void)self;
    if (alertsBytes < 0) {
        throw_iae(env, "alertsBytes < 0");
        return;
    }
    CHECK(h, rapid_sim_set_alert_set_device(ENGINE(h), (const void*)(intptr_t)dAlerts, (uint64_t)alertsBytes, (int64_t)n));
}

/* void simGenerate(long h, ByteBuffer alerts, long[] batchOff, int[] batchKeep, int[] receivers, long seed, boolean boundary): the
 * round's deliveries made on the device (rapid_sim_generate); `alerts` = the round's distinct alerts, packed, in batch order (a
 * direct buffer); batchKeep = per-batch delivery thresholds as unsigned 32-bit values (null: every batch reaches every receiver) */
JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeCutEngine_simGenerate(JNIEnv* env, jobject self, jlong h, jobject alerts, jlongArray batchOff,
                                                                      jintArray batchKeep, jintArray receivers, jlong seed, jboolean boundary) {
    (void)self;
    const jsize nb = (*env)->GetArrayLength(env, batchOff), nr = (*env)->GetArrayLength(env, receivers);
    jlong* off = (*env)->GetLongArrayElements(env, batchOff, NULL);
    jint* rx = (*env)->GetIntArrayElements(env, receivers, NULL);
    jint* keep = batchKeep != NULL ? (*env)->GetIntArrayElements(env, batchKeep, NULL) : NULL;
    int rc = RAPID_EINVAL;
    if (off != NULL && rx != NULL && nb >= 1 && (batchKeep == NULL || (keep != NULL && (*env)->GetArrayLength(env, batchKeep) == nb - 1)))
        rc = rapid_sim_generate(ENGINE(h), (const rapid_alert_record*)(*env)->GetDirectBufferAddress(env, alerts), (const int64_t*)off, (int32_t)(nb - 1),
                                (const uint32_t*)keep, (const int32_t*)rx, (int32_t)nr, (uint64_t)seed, boundary ? RAPID_GEN_BOUNDARY : RAPID_GEN_RESOLVED);
    if (keep != NULL) (*env)->ReleaseIntArrayElements(env, batchKeep, keep, JNI_ABORT);
    if (rx != NULL) (*env)->ReleaseIntArrayElements(env, receivers, rx, JNI_ABORT);
    if (off != NULL) (*env)->ReleaseLongArrayElements(env, batchOff, off, JNI_ABORT);
    CHECK(h, rc);
}

/* void simAttachStreams(long h, long dRecords, long recordsBytes, long dRecOff, int nReceivers): a round's deliveries that already
 * lie in device memory (addresses as handed out by the producer on the same GPU), tallied in place (rapid_sim_attach_streams_device) */
JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeCutEngine_simAttachStreams(JNIEnv* env, jobject self, jlong h, jlong dRecords, jlong recordsBytes,
                                                                           jlong dRecOff, jint nReceivers) {
    (void)self;
    CHECK(h, rapid_sim_attach_streams_device(ENGINE(h), (const void*)(intptr_t)dRecords, (uint64_t)recordsBytes, (const int64_t*)(intptr_t)dRecOff,
                                             (int32_t)nReceivers));
}

/* -> {decided, cutSize, votesWinner, quorum, newConfigId} */
JNIEXPORT jlongArray JNICALL Java_com_vrg_rapid_NativeCutEngine_simRound(JNIEnv* env, jobject self, jlong h, jboolean apply) {
    (void)self;
    rapid_round_result rr;
    int64_t cfg = 0;
    const int rc = rapid_sim_round(ENGINE(h), apply ? 1 : 0, &rr, &cfg);
    if (rc != RAPID_OK) {
        throw_for(env, rapid_last_error(ENGINE(h)), rc);
        return NULL;
    }
    const jlong v[5] = {rr.decided, rr.cut_size, rr.votes_winner, rr.quorum, cfg};
    jlongArray a = (*env)->NewLongArray(env, 5);
    if (a) (*env)->SetLongArrayRegion(env, a, 0, 5, v);
    return a;
}

/* a round whose deliveries and distinct alerts lie in device memory, in ONE crossing (rapid_sim_round_device) */
JNIEXPORT jlongArray JNICALL Java_com_vrg_rapid_NativeCutEngine_simRoundDevice(JNIEnv* env, jobject self, jlong h, jlong dRecords, jlong recordsBytes,
                                                                               jlong dRecOff, jint nReceivers, jlong dAlerts, jlong alertsBytes,
                                                                               jlong nAlerts, jint trust, jboolean apply) {
    (void)self;
    if (recordsBytes < 0 || alertsBytes < 0 || nAlerts < 0 || nReceivers < 0) {
        throw_iae(env, "negative size");
        return NULL;
    }
    rapid_round_result rr;
    int64_t cfg = 0;
    const int rc = rapid_sim_round_device(ENGINE(h), (const void*)(intptr_t)dRecords, (uint64_t)recordsBytes, (const int64_t*)(intptr_t)dRecOff,
                                          nReceivers, (const void*)(intptr_t)dAlerts, (uint64_t)alertsBytes, nAlerts, trust, apply ? 1 : 0, &rr, &cfg);
    if (rc != RAPID_OK) {
        throw_for(env, rapid_last_error(ENGINE(h)), rc);
        return NULL;
    }
    const jlong v[5] = {rr.decided, rr.cut_size, rr.votes_winner, rr.quorum, cfg};
    jlongArray a = (*env)->NewLongArray(env, 5);
    if (a) (*env)->SetLongArrayRegion(env, a, 0, 5, v);
    return a;
}

JNIEXPORT jintArray JNICALL Java_com_vrg_rapid_NativeCutEngine_simDecidedCut(JNIEnv* env, jobject self, jlong h) {
    (void)self;
    int32_t out[MAX_LIST], n = 0;
    const int rc = rapid_sim_decided_cut(ENGINE(h), out, MAX_LIST, &n);
    if (rc != RAPID_OK) {
        throw_for(env, rapid_last_error(ENGINE(h)), rc);
        return NULL;
    }
    return to_java(env, out, n);
}

/* ---------------------------------------------------------------- wire ingest */
/* -> content case of the RapidRequest in wire[0, len); payloadOffLen receives {offset, length} of its payload */
JNIEXPORT jint JNICALL Java_com_vrg_rapid_NativeCutEngine_decodeRequest(JNIEnv* env, jobject self, jobject wire, jint len,
                                                                        jlongArray payloadOffLen) {
    (void)self;
    int32_t kind = 0;
    int64_t ol[2] = {0, 0};
    const uint8_t* in = direct_region(env, wire, 0, len, "wire: length outside the buffer");
    if (!in) return -1;
    if (!payloadOffLen || (*env)->GetArrayLength(env, payloadOffLen) < 2) {
        throw_iae(env, "payloadOffLen must hold two longs");
        return -1;
    }
    const int rc = rapid_decode_request(in, len, &kind, &ol[0], &ol[1]);
    if (rc != RAPID_OK) {
        throw_for(env, "rapid_decode_request", rc);
        return -1;
    }
    (*env)->SetLongArrayRegion(env, payloadOffLen, 0, 2, (const jlong*)ol);
    return kind;
}

/* -> number of alerts; recordsOut: direct buffer with room for idHi.length records (20 bytes each); idLo at least as long as idHi */
JNIEXPORT jint JNICALL Java_com_vrg_rapid_NativeCutEngine_decodeBatchedAlerts(JNIEnv* env, jobject self, jlong map, jobject wire,
                                                                              jlong off, jlong len, jint K, jobject recordsOut,
                                                                              jlongArray idHi, jlongArray idLo) {
    (void)self;
    if (!idHi || !idLo) {
        throw_iae(env, "idHi / idLo");
        return -1;
    }
    const jsize cap = (*env)->GetArrayLength(env, idHi);
    if ((*env)->GetArrayLength(env, idLo) < cap) {
        throw_iae(env, "idLo is shorter than idHi");
        return -1;
    }
    const uint8_t* in = direct_region(env, wire, off, len, "wire: offset / length outside the buffer");
    if (!in) return -1;
    uint8_t* out = direct_region(env, recordsOut, 0, (jlong)cap * 20, "recordsOut: smaller than idHi.length records");
    if (!out) return -1;
    jlong* hi = (*env)->GetLongArrayElements(env, idHi, NULL);
    jlong* lo = (*env)->GetLongArrayElements(env, idLo, NULL);
    if (!hi || !lo) {
        if (lo) (*env)->ReleaseLongArrayElements(env, idLo, lo, JNI_ABORT);
        if (hi) (*env)->ReleaseLongArrayElements(env, idHi, hi, JNI_ABORT);
        throw_iae(env, "idHi / idLo not accessible");
        return -1;
    }
    int32_t n = 0, sender = -1;
    const int rc = rapid_decode_batched_alerts((const rapid_endpoint_map*)(intptr_t)map, in, len, K, (rapid_alert_record*)out,
                                               (int64_t*)hi, (int64_t*)lo, cap, &n, &sender);
    (*env)->ReleaseLongArrayElements(env, idLo, lo, 0);
    (*env)->ReleaseLongArrayElements(env, idHi, hi, 0);
    if (rc != RAPID_OK) {
        throw_for(env, "rapid_decode_batched_alerts", rc);
        return -1;
    }
    return n;
}

/* ---------------------------------------------------------------- NativeFastPaxos */
JNIEXPORT jlong JNICALL Java_com_vrg_rapid_NativeFastPaxos_create(JNIEnv* env, jclass cls, jint myIndex, jint rankIndex,
                                                                  jlong configurationId, jint membershipSize) {
    (void)cls;
    rapid_consensus* c = NULL;
    const int rc = rapid_consensus_create(myIndex, rankIndex, configurationId, membershipSize, &c);
    if (rc != RAPID_OK) throw_for(env, "rapid_consensus_create", rc);
    return (jlong)(intptr_t)c;
}

JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeFastPaxos_close(JNIEnv* env, jobject self, jlong h) {
    (void)env; (void)self;
    rapid_consensus_destroy((rapid_consensus*)(intptr_t)h);
}

JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeFastPaxos_propose(JNIEnv* env, jobject self, jlong h, jintArray proposal) {
    (void)self;
    const jsize n = (*env)->GetArrayLength(env, proposal);
    jint* p = (*env)->GetIntArrayElements(env, proposal, NULL);
    const int rc = rapid_consensus_propose((rapid_consensus*)(intptr_t)h, p, n);
    (*env)->ReleaseIntArrayElements(env, proposal, p, JNI_ABORT);
    if (rc != RAPID_OK) throw_for(env, "rapid_consensus_propose", rc);
}

/* one received consensus message: the payload of a RapidRequest of content case 5..9 inside `wire` */
JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeFastPaxos_handle(JNIEnv* env, jobject self, jlong h, jlong endpointMap,
                                                                 jint contentCase, jobject wire, jlong off, jlong len) {
    (void)self;
    const uint8_t* in = direct_region(env, wire, off, len, "wire: offset / length outside the buffer");
    if (!in) return;
    rapid_consensus_msg msg;
    /* a serialized Endpoint takes at least two bytes: len / 2 bounds the value list of any message, however large the cluster */
    const int32_t cap = (int32_t)(len / 2 + 1 < MAX_LIST ? MAX_LIST : len / 2 + 1);
    int32_t stack_eps[MAX_LIST];
    int32_t* eps = cap > MAX_LIST ? (int32_t*)malloc(sizeof(int32_t) * (size_t)cap) : stack_eps;
    if (!eps) {
        throw_for(env, "out of memory", RAPID_ESTATE);
        return;
    }
    int rc = rapid_decode_consensus_message((const rapid_endpoint_map*)(intptr_t)endpointMap, contentCase, in, len, &msg, eps, cap);
    if (rc == RAPID_OK) rc = rapid_consensus_handle((rapid_consensus*)(intptr_t)h, &msg, eps);
    if (eps != stack_eps) free(eps);
    if (rc != RAPID_OK) throw_for(env, "consensus message", rc);
}

JNIEXPORT void JNICALL Java_com_vrg_rapid_NativeFastPaxos_startClassicRound(JNIEnv* env, jobject self, jlong h) {
    (void)self;
    const int rc = rapid_consensus_start_classic_round((rapid_consensus*)(intptr_t)h);
    if (rc != RAPID_OK) throw_for(env, "rapid_consensus_start_classic_round", rc);
}

/* next outgoing message as a serialized RapidRequest in requestOut; -> its destination (-1 = broadcast), -2 if none;
 * lenOut[0] receives the number of bytes (native int poll(long h, long endpointMap, ByteBuffer requestOut, long[] lenOut)) */
JNIEXPORT jint JNICALL Java_com_vrg_rapid_NativeFastPaxos_poll(JNIEnv* env, jobject self, jlong h, jlong endpointMap,
                                                               jobject requestOut, jlongArray lenOut) {
    (void)self;
    if (!lenOut || (*env)->GetArrayLength(env, lenOut) < 1) {
        throw_iae(env, "lenOut must hold one long");
        return -2;
    }
    const jlong out_cap = requestOut ? (*env)->GetDirectBufferCapacity(env, requestOut) : -1;
    uint8_t* out = direct_region(env, requestOut, 0, out_cap < 0 ? 0 : out_cap, "requestOut must be a direct buffer");
    if (!out) return -2;
    rapid_consensus_msg msg;
    int32_t stack_eps[MAX_LIST], got = 0;
    int32_t* eps = stack_eps;
    int rc = rapid_consensus_poll((rapid_consensus*)(intptr_t)h, &msg, eps, MAX_LIST, &got);
    if (rc == RAPID_ECAPACITY && msg.n_endpoints > MAX_LIST) {  /* the message stays queued: take it with a list of its own size */
        eps = (int32_t*)malloc(sizeof(int32_t) * (size_t)msg.n_endpoints);
        if (!eps) {
            throw_for(env, "out of memory", RAPID_ESTATE);
            return -2;
        }
        rc = rapid_consensus_poll((rapid_consensus*)(intptr_t)h, &msg, eps, msg.n_endpoints, &got);
    }
    if (rc == RAPID_OK && !got) {
        if (eps != stack_eps) free(eps);
        return -2;
    }
    int64_t len = 0;
    if (rc == RAPID_OK)
        rc = rapid_encode_consensus_request((const rapid_endpoint_map*)(intptr_t)endpointMap, &msg, eps, out, out_cap, &len);
    if (eps != stack_eps) free(eps);
    if (rc != RAPID_OK) {
        throw_for(env, "outgoing consensus message", rc);
        return -2;
    }
    (*env)->SetLongArrayRegion(env, lenOut, 0, 1, (const jlong*)&len);
    return msg.dest;
}

/* the decided value (what onDecide receives), or null while there is none */
JNIEXPORT jintArray JNICALL Java_com_vrg_rapid_NativeFastPaxos_decision(JNIEnv* env, jobject self, jlong h) {
    (void)self;
    int32_t stack_out[MAX_LIST], n = 0;
    int32_t* out = stack_out;
    int rc = rapid_consensus_decision((rapid_consensus*)(intptr_t)h, out, MAX_LIST, &n);
    if (rc == RAPID_ECAPACITY && n > MAX_LIST) {
        out = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
        if (!out) {
            throw_for(env, "out of memory", RAPID_ESTATE);
            return NULL;
        }
        rc = rapid_consensus_decision((rapid_consensus*)(intptr_t)h, out, n, &n);
    }
    jintArray res = NULL;
    if (rc == RAPID_OK)
        res = to_java(env, out, n);
    else if (rc != RAPID_ESTATE)
        throw_for(env, "rapid_consensus_decision", rc);
    if (out != stack_out) free(out);
    return res;
}

JNIEXPORT jlong JNICALL Java_com_vrg_rapid_NativeFastPaxos_fallbackDelayMs(JNIEnv* env, jclass cls, jint membershipSize,
                                                                           jlong baseDelayMs, jdouble u) {
    (void)cls;
    int64_t d = 0;
    const int rc = rapid_consensus_fallback_delay_ms(membershipSize, baseDelayMs, u, &d);
    if (rc != RAPID_OK) throw_for(env, "rapid_consensus_fallback_delay_ms", rc);
    return d;
}
