// SOURCE ONLY -- not compiled in this repository (no JDK in the build image).  Times the REAL MultiNodeCutDetector
// (MultiNodeCutDetector.java:76-128, 137-164) on a record file written by scripts/export_records.py -- the same per-receiver
// alert streams bench.py replays on the GPU -- so that the CPU baseline next to the GPU number can be the reference itself
// instead of its restatement (bench.py's cpu_baseline.kind would then be "reference").  It also prints, per receiver, the
// batch that announced a proposal and the proposal size, for a bit-for-bit comparison with the engine's results
// (python oracle/check_java_dump.py --cuts java_cuts.txt <records file>).
//
// How to run (in a checkout of lalithsuresh/rapid with Java 8 + Maven):
//   python scripts/export_records.py C3b 64 records_c3b_64rx.bin                (here: 64 receivers of BASELINE configs[2])
//   cp tools/java/CutDetectorBench.java rapid/src/test/java/com/vrg/rapid/
//   mvn -q -pl rapid test-compile dependency:build-classpath -Dmdep.outputFile=cp.txt
//   java -cp rapid/target/classes:rapid/target/test-classes:$(cat rapid/cp.txt) com.vrg.rapid.CutDetectorBench records_c3b_64rx.bin > java_cuts.txt
//
// File format (little endian): magic "RAPIDREC", int32 n_nodes, K, H, L, int64 config_id, int32 n_members, int32 members[],
// per node: int32 hostname length, hostname bytes, int32 port, int64 id_hi, id_lo; int32 n_receivers, int64 rec_off[n_receivers + 1];
// then the 20-byte records of include/rapid_mi355x.h (int64 cfg_id, uint32 src, uint32 dst, uint16 ring_mask, uint8 status, uint8 flags).
package com.vrg.rapid;

import com.vrg.rapid.pb.AlertMessage;
import com.vrg.rapid.pb.EdgeStatus;
import com.vrg.rapid.pb.Endpoint;
import com.vrg.rapid.pb.NodeId;

import java.io.DataInputStream;
import java.io.FileInputStream;
import java.io.IOException;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.channels.FileChannel;
import java.util.ArrayList;
import java.util.HashSet;
import java.util.List;
import java.util.Set;

public final class CutDetectorBench {
    private CutDetectorBench() {
    }

    public static void main(final String[] args) throws IOException {
        final FileChannel ch = new FileInputStream(args[0]).getChannel();
        final ByteBuffer f = ch.map(FileChannel.MapMode.READ_ONLY, 0, ch.size()).order(ByteOrder.LITTLE_ENDIAN);
        final byte[] magic = new byte[8];
        f.get(magic);
        if (!new String(magic, "US-ASCII").equals("RAPIDREC")) {
            throw new IOException("not a record file");
        }
        final int nNodes = f.getInt();
        final int k = f.getInt();
        final int h = f.getInt();
        final int l = f.getInt();
        final long configId = f.getLong();
        final int nMembers = f.getInt();
        final int[] members = new int[nMembers];
        for (int i = 0; i < nMembers; i++) {
            members[i] = f.getInt();
        }
        final Endpoint[] endpoints = new Endpoint[nNodes];
        final NodeId[] ids = new NodeId[nNodes];
        for (int i = 0; i < nNodes; i++) {
            final byte[] host = new byte[f.getInt()];
            f.get(host);
            endpoints[i] = Utils.hostFromParts(new String(host, "UTF-8"), f.getInt());
            ids[i] = NodeId.newBuilder().setHigh(f.getLong()).setLow(f.getLong()).build();
        }
        final MembershipView view = new MembershipView(k);
        for (final int m : members) {
            view.ringAdd(endpoints[m], ids[m]);
        }
        if (view.getCurrentConfigurationId() != configId) {
            System.err.println("WARNING: configuration id " + view.getCurrentConfigurationId() + " != file's " + configId
                    + " (the restated ring hash differs from the real one: report this)");
        }
        final int nReceivers = f.getInt();
        final long[] recOff = new long[nReceivers + 1];
        for (int i = 0; i <= nReceivers; i++) {
            recOff[i] = f.getLong();
        }
        final int base = f.position();
        long batches = 0;
        final long t0 = System.nanoTime();
        for (int r = 0; r < nReceivers; r++) {
            // one receiver = MembershipService.handleMessage(BatchedAlertMessage) over its stream (MembershipService.java:300-354)
            final MultiNodeCutDetector cd = new MultiNodeCutDetector(k, h, l);
            final Set<Endpoint> proposal = new HashSet<>();
            int emitBatch = -1;
            int batch = 0;
            boolean announced = false;
            final List<AlertMessage> current = new ArrayList<>();
            for (long i = recOff[r]; i < recOff[r + 1]; i++) {
                final int p = base + (int) (i * 20);
                final long cfg = f.getLong(p);
                final int src = f.getInt(p + 8);
                final int dst = f.getInt(p + 12);
                final int mask = f.getShort(p + 16) & 0xFFFF;
                final int status = f.get(p + 18) & 0xFF;
                final int flags = f.get(p + 19) & 0xFF;
                final AlertMessage.Builder b = AlertMessage.newBuilder().setEdgeSrc(endpoints[src]).setEdgeDst(endpoints[dst])
                        .setEdgeStatus(status == 1 ? EdgeStatus.DOWN : EdgeStatus.UP).setConfigurationId(cfg);
                for (int ring = 0; ring < k; ring++) {
                    if (((mask >> ring) & 1) != 0) {
                        b.addRingNumber(ring);
                    }
                }
                current.add(b.build());
                final boolean last = (flags & 1) != 0 || i + 1 == recOff[r + 1];
                if (!last) {
                    continue;
                }
                if (!announced) {
                    // filterAlertMessages (MembershipService.java:644-675) + aggregate + invalidate (:318-335)
                    for (final AlertMessage msg : current) {
                        if (msg.getConfigurationId() != configId) {
                            continue;
                        }
                        final boolean present = view.isHostPresent(msg.getEdgeDst());
                        if ((msg.getEdgeStatus() == EdgeStatus.UP && present) || (msg.getEdgeStatus() == EdgeStatus.DOWN && !present)) {
                            continue;
                        }
                        proposal.addAll(cd.aggregateForProposal(msg));
                    }
                    proposal.addAll(cd.invalidateFailingEdges(view));
                    if (!proposal.isEmpty()) {
                        announced = true;
                        emitBatch = batch;
                    }
                }
                current.clear();
                batch++;
                batches++;
            }
            System.out.println("receiver " + r + " emit_batch " + emitBatch + " num_proposals " + cd.getNumProposals() + " cut_size "
                    + proposal.size());
        }
        final double s = (System.nanoTime() - t0) * 1e-9;
        System.out.println("alert_batches " + batches + " seconds " + s + " alert_batches_per_s " + (batches / s) + " threads 1");
    }
}
