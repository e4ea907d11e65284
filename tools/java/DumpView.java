// SOURCE ONLY -- not compiled in this repository (no JDK in the build image).  Closes the "parity unpinned" gap of
// SURVEY.md 8c(4): ring orders and configuration ids of the REAL MembershipView (OpenHFT XXH64, MembershipView.java:360-388,
// 544-587) for the inputs of SURVEY Appendix C and for the committed golden fixtures, in the format
// oracle/check_java_dump.py reads.
//
// How to run (in a checkout of lalithsuresh/rapid with Java 8 + Maven):
//   cp tools/java/DumpView.java rapid/src/test/java/com/vrg/rapid/          (package-private classes: same package)
//   mvn -q -pl rapid test-compile dependency:build-classpath -Dmdep.outputFile=cp.txt
//   java -cp rapid/target/classes:rapid/target/test-classes:$(cat rapid/cp.txt) com.vrg.rapid.DumpView > java_view_dump.txt
//   python oracle/check_java_dump.py java_view_dump.txt        (back in this repository: compares with oracle/)
package com.vrg.rapid;

import com.vrg.rapid.pb.Endpoint;
import com.vrg.rapid.pb.NodeId;

import java.util.ArrayList;
import java.util.List;

public final class DumpView {
    private static final long SEED_IDS = 0x5241504944L; // "RAPID", rapid_amd/scenarios.py

    private DumpView() {
    }

    /** splitmix64 of (seed + (idx + 1) * golden), as rapid_amd/scenarios.py:splitmix64. */
    private static long splitmix64(final long seed, final long idx) {
        long z = seed + (idx + 1) * 0x9E3779B97F4A7C15L;
        z = (z ^ (z >>> 30)) * 0xBF58476D1CE4E5B9L;
        z = (z ^ (z >>> 27)) * 0x94D049BB133111EBL;
        return z ^ (z >>> 31);
    }

    private static void dump(final String name, final int k, final List<Endpoint> endpoints, final List<NodeId> ids) {
        final MembershipView view = new MembershipView(k);
        for (int i = 0; i < endpoints.size(); i++) {
            view.ringAdd(endpoints.get(i), ids.get(i));
        }
        System.out.println("view " + name + " K " + k + " n " + endpoints.size());
        System.out.println("config_id " + view.getCurrentConfigurationId());
        for (int ring = 0; ring < k; ring++) {
            final StringBuilder sb = new StringBuilder("ring " + ring);
            for (final Endpoint e : view.getRing(ring)) {
                sb.append(' ').append(e.getHostname().toStringUtf8()).append(':').append(e.getPort());
            }
            System.out.println(sb);
        }
        // observers of the first endpoint (KAT-7 of CutDetectionTest uses them)
        final StringBuilder sb = new StringBuilder("observers_of_first");
        for (final Endpoint e : view.getObserversOf(endpoints.get(0))) {
            sb.append(' ').append(e.getHostname().toStringUtf8()).append(':').append(e.getPort());
        }
        System.out.println(sb);
        System.out.println("end");
    }

    public static void main(final String[] args) {
        // SURVEY Appendix C: 127.0.0.1:{1234..1239}, NodeIds (i, i)
        final List<Endpoint> a = new ArrayList<>();
        final List<NodeId> aid = new ArrayList<>();
        for (int p = 1234; p <= 1239; p++) {
            a.add(Utils.hostFromParts("127.0.0.1", p));
            aid.add(NodeId.newBuilder().setHigh(p - 1234).setLow(p - 1234).build());
        }
        dump("appendix_c_six_ports", 10, a, aid);
        dump("appendix_c_single", 10, a.subList(0, 1), aid.subList(0, 1));
        // KAT-7 view of CutDetectionTest.java:254-301: 127.0.0.2:2..31, NodeIds (i + 1, i + 1)
        final List<Endpoint> b = new ArrayList<>();
        final List<NodeId> bid = new ArrayList<>();
        for (int i = 0; i < 30; i++) {
            b.add(Utils.hostFromParts("127.0.0.2", 2 + i));
            bid.add(NodeId.newBuilder().setHigh(i + 1).setLow(i + 1).build());
        }
        dump("kat7_thirty_nodes", 10, b, bid);
        // the populations of tests/golden/*.npz: 10.a.b.c:5000 with splitmix64 NodeIds (rapid_amd/scenarios.py:Population.make)
        for (final int[] nk : new int[][] {{50, 3}, {300, 10}, {400, 10}}) {
            final List<Endpoint> g = new ArrayList<>();
            final List<NodeId> gid = new ArrayList<>();
            for (int j = 0; j < nk[0]; j++) {
                g.add(Utils.hostFromParts("10." + ((j >> 16) & 255) + "." + ((j >> 8) & 255) + "." + (j & 255), 5000));
                gid.add(NodeId.newBuilder().setHigh(splitmix64(SEED_IDS, 2L * j)).setLow(splitmix64(SEED_IDS, 2L * j + 1)).build());
            }
            dump("golden_n" + nk[0] + "_k" + nk[1], nk[1], g, gid);
        }
    }
}
