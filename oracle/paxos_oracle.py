"""TEST INFRASTRUCTURE ONLY -- never imported by the product (rapid_amd/), only by tests/.

A line-by-line restatement, in plain Python, of the reference's consensus classes for ONE configuration change:

  Paxos      <- rapid/src/main/java/com/vrg/rapid/Paxos.java:57-328      (classic rounds + coordinator rule)
  FastPaxos  <- rapid/src/main/java/com/vrg/rapid/FastPaxos.java:62-204  (fast round, dispatch, fallback)

Endpoints are node indices, a value (List<Endpoint>) is a tuple of node indices (ORDERED: the Java compares lists),
a Rank (rapid.proto:133-137) is the tuple (round, nodeIndex) and compares like Paxos.compareRanks (:333-339), which is
Python's tuple order.  Messages are small dataclasses with the fields of rapid.proto:124-169.  Instead of an
IBroadcaster / IMessagingClient the classes take two callables: broadcast(msg) and send(dest, msg).

One thing cannot be restated: Paxos.startPhase1a takes the coordinator's half of its rank from `myAddr.hashCode()`
(:102), the hash of a generated protobuf message, which mixes in the identity hash of the descriptor and so differs
from JVM to JVM.  The protocol only needs it to be distinct per node; here it is a constructor argument.

Pinned by tests/test_paxos_oracle.py against the cases of rapid/src/test/java/com/vrg/rapid/PaxosTests.java.
"""
import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Tuple

Rank = Tuple[int, int]
Value = Tuple[int, ...]

FAST_ROUND_PHASE2B, PHASE1A, PHASE1B, PHASE2A, PHASE2B = 5, 6, 7, 8, 9  # RapidRequest.content field numbers, rapid.proto:28-32


@dataclass(frozen=True)
class FastRoundPhase2bMessage:  # rapid.proto:124-129
    sender: int
    configurationId: int
    endpoints: Value
    kind: int = FAST_ROUND_PHASE2B


@dataclass(frozen=True)
class Phase1aMessage:  # rapid.proto:139-144
    sender: int
    configurationId: int
    rank: Rank
    kind: int = PHASE1A


@dataclass(frozen=True)
class Phase1bMessage:  # rapid.proto:146-153
    sender: int
    configurationId: int
    rnd: Rank
    vrnd: Rank
    vval: Value
    kind: int = PHASE1B


@dataclass(frozen=True)
class Phase2aMessage:  # rapid.proto:155-161
    sender: int
    configurationId: int
    rnd: Rank
    vval: Value
    kind: int = PHASE2A


@dataclass(frozen=True)
class Phase2bMessage:  # rapid.proto:163-169
    sender: int
    configurationId: int
    rnd: Rank
    endpoints: Value
    kind: int = PHASE2B


class Paxos:
    """Paxos.java.  Field names are the Java's."""

    def __init__(self, myAddr: int, configurationId: int, N: int, send: Callable, broadcast: Callable,
                 onDecide: Callable, myAddrHashCode: int):
        # :57-70
        self.myAddr = myAddr
        self.configurationId = configurationId
        self.N = N
        self.send = send
        self.broadcast = broadcast
        self.onDecide = onDecide
        self.myAddrHashCode = myAddrHashCode
        self.crnd: Rank = (0, 0)
        self.rnd: Rank = (0, 0)
        self.vrnd: Rank = (0, 0)
        self.vval: Value = ()
        self.cval: Value = ()
        self.phase1bMessages: List[Phase1bMessage] = []
        self.acceptResponses: Dict[Rank, Dict[int, Phase2bMessage]] = {}
        self.decided = False

    def startPhase1a(self, round_: int):  # :98-111
        if self.crnd[0] > round_:
            return
        self.crnd = (round_, self.myAddrHashCode)
        self.broadcast(Phase1aMessage(sender=self.myAddr, configurationId=self.configurationId, rank=self.crnd))

    def handlePhase1aMessage(self, m: Phase1aMessage):  # :118-148
        if m.configurationId != self.configurationId:
            return
        if self.rnd < m.rank:
            self.rnd = m.rank
        else:
            return  # rejected: lower or equal rank
        self.send(m.sender, Phase1bMessage(sender=self.myAddr, configurationId=self.configurationId, rnd=self.rnd,
                                           vrnd=self.vrnd, vval=self.vval))

    def handlePhase1bMessage(self, m: Phase1bMessage):  # :156-188
        if m.configurationId != self.configurationId:
            return
        if self.crnd != m.rnd:  # compareRanks(crnd, m.rnd) != 0
            return
        self.phase1bMessages.append(m)  # a List: a message delivered twice counts twice
        if len(self.phase1bMessages) > self.N // 2:
            chosenProposal = self.selectProposalUsingCoordinatorRule(self.phase1bMessages)
            if self.crnd == m.rnd and len(self.cval) == 0 and len(chosenProposal) > 0:
                self.cval = chosenProposal
                self.broadcast(Phase2aMessage(sender=self.myAddr, configurationId=self.configurationId, rnd=self.crnd,
                                              vval=chosenProposal))

    def handlePhase2aMessage(self, m: Phase2aMessage):  # :195-216
        if m.configurationId != self.configurationId:
            return
        if self.rnd <= m.rnd and self.vrnd != m.rnd:
            self.rnd = m.rnd
            self.vrnd = m.rnd
            self.vval = m.vval
            self.broadcast(Phase2bMessage(sender=self.myAddr, configurationId=self.configurationId, rnd=m.rnd,
                                          endpoints=self.vval))

    def handlePhase2bMessage(self, m: Phase2bMessage):  # :223-238
        if m.configurationId != self.configurationId:
            return
        inRnd = self.acceptResponses.setdefault(m.rnd, {})
        inRnd[m.sender] = m
        if len(inRnd) > self.N // 2 and not self.decided:
            self.onDecide(m.endpoints)
            self.decided = True

    def registerFastRoundVote(self, vote: Value):  # :246-259
        if self.rnd[0] > 1:
            return
        self.rnd = (1, 1)
        self.vrnd = self.rnd
        self.vval = tuple(vote)

    def selectProposalUsingCoordinatorRule(self, phase1bMessages: List[Phase1bMessage]) -> Value:  # :271-328
        if not phase1bMessages:
            raise ValueError("phase1bMessages was empty")  # IllegalArgumentException, :274
        maxVrndSoFar = max(m.vrnd for m in phase1bMessages)
        collectedVvals = [m.vval for m in phase1bMessages if m.vrnd == maxVrndSoFar and len(m.vval) > 0]
        setOfCollectedVvals = set(collectedVvals)
        chosenProposal = None
        if len(setOfCollectedVvals) == 1:  # :287-289
            chosenProposal = collectedVvals[0]
        elif len(collectedVvals) > 1:  # :293-307
            counters: Dict[Value, int] = {}
            for value in collectedVvals:
                if value not in counters:
                    counters[value] = 0
                count = counters[value]
                if count + 1 > self.N // 4:
                    chosenProposal = value
                    break
                counters[value] = count + 1
        if chosenProposal is None:  # :319-326: any proposed value, in arrival order; possibly the empty list
            chosenProposal = next((m.vval for m in phase1bMessages if len(m.vval) > 0), ())
        return chosenProposal


class FastPaxos:
    """FastPaxos.java.  The ScheduledExecutorService is the caller's: propose() returns, the caller decides when (or
    whether) startClassicPaxosRound() fires."""

    def __init__(self, myAddr: int, configurationId: int, membershipSize: int, send: Callable, broadcast: Callable,
                 onDecide: Callable, myAddrHashCode: int):
        # :62-87
        self.myAddr = myAddr
        self.configurationId = configurationId
        self.membershipSize = membershipSize
        self.broadcast = broadcast
        self.jitterRate = 1.0 / membershipSize
        self.votesReceived = set()
        self.votesPerProposal: Dict[Value, int] = {}
        self.decided = False
        self.decision = None
        self._onDecide = onDecide
        self.paxos = Paxos(myAddr, configurationId, membershipSize, send, broadcast, self._onDecidedWrapped, myAddrHashCode)

    def _onDecidedWrapped(self, hosts: Value):  # :78-85
        assert not self.decided  # the Java asserts it as well
        self.decided = True
        self.decision = tuple(hosts)
        self._onDecide(tuple(hosts))

    def propose(self, proposal: Value):  # :95-110 (without the timer)
        self.paxos.registerFastRoundVote(tuple(proposal))
        self.broadcast(FastRoundPhase2bMessage(sender=self.myAddr, configurationId=self.configurationId,
                                               endpoints=tuple(proposal)))

    def handleFastRoundProposal(self, m: FastRoundPhase2bMessage):  # :125-156
        if m.configurationId != self.configurationId:
            return
        if m.sender in self.votesReceived:
            return
        if self.decided:
            return
        self.votesReceived.add(m.sender)
        count = self.votesPerProposal.get(m.endpoints, 0) + 1
        self.votesPerProposal[m.endpoints] = count
        F = int(math.floor((self.membershipSize - 1) / 4.0))
        if len(self.votesReceived) >= self.membershipSize - F:
            if count >= self.membershipSize - F:
                self._onDecidedWrapped(m.endpoints)

    def handleMessages(self, m):  # :163-185
        if m.kind == FAST_ROUND_PHASE2B:
            self.handleFastRoundProposal(m)
        elif m.kind == PHASE1A:
            self.paxos.handlePhase1aMessage(m)
        elif m.kind == PHASE1B:
            self.paxos.handlePhase1bMessage(m)
        elif m.kind == PHASE2A:
            self.paxos.handlePhase2aMessage(m)
        elif m.kind == PHASE2B:
            self.paxos.handlePhase2bMessage(m)
        else:
            raise ValueError("Unexpected message case")  # IllegalArgumentException, :181

    def startClassicPaxosRound(self):  # :190-196
        if not self.decided:
            self.paxos.startPhase1a(2)

    def getRandomDelayMs(self, u: float, baseDelayMs: int) -> int:  # :201-204, u = nextDouble() in [0, 1)
        jitter = int(-1000 * math.log(1 - u) / self.jitterRate)
        return jitter + baseDelayMs


@dataclass
class Network:
    """What PaxosTests' DirectBroadcaster / DirectMessagingClient do (PaxosTests.java:403-476): every node has ONE
    single-threaded executor, so messages to a node are handled in the order they were sent to it; different nodes run
    concurrently -- here: interleaved by a seeded generator.  `drop` holds message kinds the broadcaster swallows."""
    n: int
    configurationId: int = 1
    seed: int = 0
    hash_codes: List[int] = None
    live: List[int] = None  # nodes that exist as instances; others are silent (crashed)
    drop: set = field(default_factory=set)
    membershipSize: int = None

    def __post_init__(self):
        import random
        self.rng = random.Random(self.seed)
        self.live = list(range(self.n)) if self.live is None else list(self.live)
        self.membershipSize = self.n if self.membershipSize is None else self.membershipSize
        hc = self.hash_codes or [1000 + 7 * i for i in range(self.n)]
        self.queues: Dict[int, List] = {i: [] for i in self.live}
        self.decisions: List[Tuple[int, Value]] = []
        self.log: List[Tuple[int, object]] = []  # (destination, message) in delivery order
        self.nodes: Dict[int, FastPaxos] = {}
        for i in self.live:
            self.nodes[i] = FastPaxos(i, self.configurationId, self.membershipSize, self._send, self._broadcast,
                                      (lambda v, i=i: self.decisions.append((i, v))), hc[i])

    def _send(self, dest, m):
        if dest in self.queues:
            self.queues[dest].append(m)

    def _broadcast(self, m):
        if m.kind in self.drop:
            return
        for dest in self.live:  # to every member, the sender included (UnicastToAllBroadcaster.java:46-53)
            self.queues[dest].append(m)

    def pending(self):
        return [i for i in self.live if self.queues[i]]

    def deliver_one(self, dest):
        m = self.queues[dest].pop(0)
        self.log.append((dest, m))
        self.nodes[dest].handleMessages(m)
        return m

    def run(self, max_steps=10_000_000):
        steps = 0
        while steps < max_steps:
            p = self.pending()
            if not p:
                return steps
            self.deliver_one(self.rng.choice(p))
            steps += 1
        raise RuntimeError("no quiescence")


class TimedNetwork:
    """The same classes on a time line: a message sent at time t to node j is handled at t + delay(sender, j); handlers
    run in time order (ties: lower destination, then send order) and send at the time they run.  For the protocol-time
    checks of tests/test_timeline.py; delay(sender, receiver) -> integer milliseconds."""

    def __init__(self, membershipSize, live, configurationId, delay, hash_codes=None):
        import heapq
        self._heapq = heapq
        self.live = list(live)
        self.delay = delay
        self.now = 0
        self._events = []
        self._seq = 0
        self.decision_ms = {}
        self.decisions = {}
        self.nodes = {}
        for i in self.live:
            self.nodes[i] = FastPaxos(i, configurationId, membershipSize, (lambda dest, m, i=i: self._post(i, dest, m)),
                                      (lambda m, i=i: [self._post(i, dest, m) for dest in self.live]),
                                      (lambda v, i=i: self._decided(i, v)), (hash_codes or {}).get(i, i + 2))

    def _post(self, sender, dest, m):
        if dest in self.nodes:
            self._heapq.heappush(self._events, (self.now + int(self.delay(sender, dest)), dest, self._seq, m))
            self._seq += 1

    def _decided(self, i, v):
        self.decisions[i] = v
        self.decision_ms[i] = self.now

    def at(self, t, fn):
        """run fn() (e.g. a node's propose or startClassicPaxosRound) at time t"""
        self._heapq.heappush(self._events, (t, -1, self._seq, fn))
        self._seq += 1

    def run(self):
        while self._events:
            t, dest, _, what = self._heapq.heappop(self._events)
            self.now = t
            if dest < 0:
                what()
            else:
                self.nodes[dest].handleMessages(what)
