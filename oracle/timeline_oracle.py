"""TEST INFRASTRUCTURE ONLY -- never imported by the product (rapid_amd/), only by tests/.

The alert PRODUCER side of the reference as a literal discrete-event simulation: the timers and
counters that turn "node s crashed at time t" into BatchedAlertMessages on the wire,

  PingPongFailureDetector  <- rapid/src/main/java/com/vrg/rapid/monitoring/impl/PingPongFailureDetector.java:39-126
  MembershipService        <- rapid/src/main/java/com/vrg/rapid/MembershipService.java:144-148 (batcher job),
                              :572-581 (enqueueAlertMessage), :472-495 (edgeFailureNotification),
                              :613-637 (AlertBatcher.run), :697-706 (one detector per entry of getSubjectsOf)

PARITY UNPINNED: the reference holds no known-answer test for these timers (its detectors are only exercised end to
end, through ClusterTest with real sleeps) and cannot be run here, so this restatement rests on the cited statements
and constants alone.

Restated class by class, field by field, with a heap of timed events in place of the ScheduledExecutorService and
System.currentTimeMillis().  Time is integer milliseconds.  What the Java leaves to chance is fixed here, and in the
product, as follows: events with the same timestamp run in the order  probe callbacks < detector ticks (ring order
within a node) < batcher ticks;  a probe to a subject that has crashed by the time it is sent fails `probe_fail_ms`
later (connection refused: small; gRPC probe timeout: DEFAULT_GRPC_PROBE_TIMEOUT = 1000, GrpcClient.java:59); a probe
sent earlier succeeds.
"""
import heapq

FAILURE_THRESHOLD = 10  # PingPongFailureDetector.java:41
DOWN = 1

PRIO_CALLBACK, PRIO_DETECTOR, PRIO_BATCHER = 0, 1, 2


class Clock:
    def __init__(self):
        self.now = 0
        self._heap = []
        self._seq = 0

    def at(self, t, prio, fn):
        heapq.heappush(self._heap, (t, prio, self._seq, fn))
        self._seq += 1

    def run_until(self, t_end):
        while self._heap and self._heap[0][0] <= t_end:
            t, _, _, fn = heapq.heappop(self._heap)
            self.now = t
            fn()

    def fixed_rate(self, t0, period, prio, fn, t_end):
        """scheduleAtFixedRate(fn, initialDelay, period): runs at t0, t0 + period, ..."""
        def tick(t=t0):
            fn()
            if t + period <= t_end:
                self.at(t + period, prio, lambda: tick(t + period))
        self.at(t0, prio, tick)


class PingPongFailureDetector:
    """:39-126.  `probe(subject, on_failure)` stands for rpcClient.sendMessageBestEffort + ProbeCallback."""

    def __init__(self, subject, probe, notifier):
        self.subject = subject
        self.probe = probe
        self.notifier = notifier
        self.failureCount = 0
        self.notified = False

    def hasFailed(self):  # :70-72
        return self.failureCount >= FAILURE_THRESHOLD

    def run(self):  # :75-85
        if self.hasFailed() and not self.notified:
            self.notified = True
            self.notifier()
        else:
            self.probe(self.subject, self.handleProbeOnFailure)

    def handleProbeOnFailure(self):  # :122-125
        self.failureCount += 1


class Node:
    """The timers of one MembershipService."""

    def __init__(self, sim, me, start_ms):
        self.sim, self.me = sim, me
        self.sendQueue = []
        self.lastEnqueueTimestamp = -1  # :98
        clock, p = sim.clock, sim.params
        # :147-148 -- alertBatcherJob = scheduleAtFixedRate(new AlertBatcher(), 0, batchingWindow)
        clock.fixed_rate(start_ms, p["batching_window_ms"], PRIO_BATCHER, self.alertBatcherRun, sim.t_end)
        # :697-706 -- one detector per entry of getSubjectsOf(myAddr), duplicates included, initial delay 0
        for k, subject in enumerate(sim.subjects_of(me)):
            fd = PingPongFailureDetector(subject, self.probe, (lambda s=subject: self.edgeFailureNotification(s)))
            clock.fixed_rate(start_ms, p["fd_interval_ms"], PRIO_DETECTOR, fd.run, sim.t_end)

    def alive(self):
        return self.sim.clock.now < self.sim.crash_ms[self.me]

    def probe(self, subject, on_failure):
        if not self.alive():
            return  # a crashed node runs nothing
        sim = self.sim
        if sim.clock.now >= sim.crash_ms[subject]:
            sim.clock.at(sim.clock.now + sim.params["probe_fail_ms"], PRIO_CALLBACK,
                         lambda: on_failure() if self.alive() else None)

    def edgeFailureNotification(self, subject):  # :472-495 (the configuration does not change during the run)
        if not self.alive():
            return
        msg = (self.me, subject, DOWN, tuple(self.sim.ring_numbers(self.me, subject)))
        self.lastEnqueueTimestamp = self.sim.clock.now  # enqueueAlertMessage, :572-581
        self.sendQueue.append(msg)

    def alertBatcherRun(self):  # :613-637
        if not self.alive():
            return
        now = self.sim.clock.now
        if self.sendQueue and self.lastEnqueueTimestamp > 0 and (now - self.lastEnqueueTimestamp) > self.sim.params["batching_window_ms"]:
            messages, self.sendQueue = self.sendQueue, []
            self.sim.broadcasts.append((now, self.me, messages))


class ProducerSimulation:
    """subj[n][k] = subject of n on ring k (getSubjectsOf, MembershipView.java:267-282); crash_ms[n] = when n stops
    (None = never); start_ms[n] = when n's MembershipService was constructed (the phase of all its timers)."""

    def __init__(self, subj, crash_ms, start_ms, t_end, fd_interval_ms=1000, batching_window_ms=100, probe_fail_ms=1):
        self.subj = subj
        self.crash_ms = [float("inf") if c is None else c for c in crash_ms]
        self.params = dict(fd_interval_ms=fd_interval_ms, batching_window_ms=batching_window_ms, probe_fail_ms=probe_fail_ms)
        self.t_end = t_end
        self.clock = Clock()
        self.broadcasts = []  # (send time, sender, [(src, dst, status, ring numbers)])
        self.nodes = [Node(self, i, start_ms[i]) for i in range(len(subj))]

    def subjects_of(self, n):
        return [int(s) for s in self.subj[n]]

    def ring_numbers(self, observer, subject):  # getRingNumbers, MembershipView.java:397-418
        return [k for k, s in enumerate(self.subj[observer]) if int(s) == subject]

    def run(self):
        self.clock.run_until(self.t_end)
        return sorted(self.broadcasts, key=lambda b: (b[0], b[1]))
