"""Test configuration.

`-m gpu` runs are ISOLATED: the collecting process never touches the device; every test MODULE's GPU tests run in a child
`python -m pytest <node ids>` whose per-test reports come back through a JSON-lines file and are replayed here, so the
terminal output and the final counts are those of an ordinary run.  A device fault (`Memory access fault by GPU`, SIGABRT)
kills the child, not the session: the test that was running is reported as failed with the child's last output, a new
child takes the rest of the module, and the other modules run as if nothing had happened.  (GPUTEST_r05: one fault in the first
test erased the record of 58 tests.)  For the same reason a GPU run does not stop at the first failure even under `-x`: a
failure still fails the run (exit code 1), the tail still says how many tests passed.
RAPID_GPU_NO_ISOLATION=1 runs everything in-process (debuggers, rocprofv3 around one pytest process)."""
import json
import os
import subprocess
import sys
import tempfile
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHILD_ENV = "RAPID_GPU_CHILD_REPORT"  # set in a child: the JSON-lines file its reports go to
MODULE_TIMEOUT_S = 1500


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config._rapid_isolate = _wants_isolation(config)
    if config._rapid_isolate:
        config.option.maxfail = 0  # (see the module docstring)


def _wants_isolation(config):
    if os.environ.get(CHILD_ENV) or os.environ.get("RAPID_GPU_NO_ISOLATION"):
        return False
    expr = (config.getoption("markexpr", "") or "").replace(" ", "")
    return expr == "gpu"


# ---- child side: every report of every phase, one JSON line each, flushed to disk before the next test starts ------------------
def pytest_runtest_logreport(report):
    path = os.environ.get(CHILD_ENV)
    if not path:
        return
    rec = {"nodeid": report.nodeid, "when": report.when, "outcome": report.outcome, "duration": getattr(report, "duration", 0.0),
           "longrepr": None if report.longrepr is None else str(report.longrepr)[-6000:],
           "skip": list(report.longrepr) if report.skipped and isinstance(report.longrepr, tuple) else None,
           "sections": [(a, b[-2000:]) for a, b in report.sections][-4:]}
    with open(path, "a") as f:
        f.write(json.dumps(rec) + "\n")
        f.flush()
        os.fsync(f.fileno())


# ---- parent side -----------------------------------------------------------------------------------------------------------------
def _run_child(nodeids, report_path):
    env = dict(os.environ)
    env[CHILD_ENV] = report_path
    # (-s: the child does not capture -- what the runtime prints when it aborts the process, "Memory access fault by GPU ...", goes to
    # the child's real stderr and so into `out`; captured, it would die with the child's capture buffers)
    cmd = [sys.executable, "-m", "pytest", "-q", "-s", "-m", "gpu", "-p", "no:cacheprovider", "--no-header"] + nodeids
    t0 = time.time()
    try:
        p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=MODULE_TIMEOUT_S)
        rc, out = p.returncode, p.stdout
    except subprocess.TimeoutExpired as e:
        rc, out = -999, (e.stdout or "") if isinstance(e.stdout, str) else (e.stdout or b"").decode(errors="replace")
    return rc, out, time.time() - t0


def _replay(item, recs):
    from _pytest.reports import TestReport
    item.ihook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
    for rec in recs:
        longrepr = rec["longrepr"]
        if rec["outcome"] == "skipped" and rec.get("skip"):
            longrepr = tuple(rec["skip"])
        rep = TestReport(nodeid=item.nodeid, location=item.location, keywords={k: 1 for k in item.keywords}, outcome=rec["outcome"],
                         longrepr=longrepr, when=rec["when"], sections=[tuple(s) for s in rec.get("sections") or []],
                         duration=rec.get("duration", 0.0))
        item.ihook.pytest_runtest_logreport(report=rep)
    item.ihook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)


def _crash_records(item, rc, out, why):
    sig = " (signal %d)" % -rc if rc < 0 and rc != -999 else ""
    lines = [l for l in out.splitlines() if l.strip()]
    # what names the cause first (the runtime's fault line, the interpreter's verdict, the frames inside this repository), then the tail
    key = [l for l in lines if "Memory access fault" in l or "Fatal Python error" in l or "HSA_STATUS" in l or
           ('File "' in l and os.sep + "site-packages" + os.sep not in l and "dist-packages" not in l and "/usr/lib/python" not in l)][:12]
    tail = "\n".join(lines)[-2000:]
    msg = "%s: the child process running this test exited with code %d%s before reporting it.\n%s\n---- child output (tail) ----\n%s" % (
        why, rc, sig, "\n".join(key), tail)
    return [{"nodeid": item.nodeid, "when": "call", "outcome": "failed", "longrepr": msg, "duration": 0.0}]


@pytest.hookimpl(tryfirst=True)
def pytest_runtestloop(session):
    config = session.config
    if not getattr(config, "_rapid_isolate", False) or config.option.collectonly:
        return None
    if session.testsfailed and not config.option.continue_on_collection_errors:
        raise session.Interrupted("%d error%s during collection" % (session.testsfailed, "s" if session.testsfailed != 1 else ""))
    modules, order = {}, []
    for item in session.items:
        key = str(item.fspath)
        if key not in modules:
            modules[key] = []
            order.append(key)
        modules[key].append(item)
    tmpdir = tempfile.mkdtemp(prefix="rapid_gpu_reports_")
    tw = config.get_terminal_writer()
    for mi, key in enumerate(order):
        pending = list(modules[key])
        attempt = 0
        while pending:
            attempt += 1
            report_path = os.path.join(tmpdir, "m%d_a%d.jsonl" % (mi, attempt))
            rc, out, secs = _run_child([it.nodeid for it in pending], report_path)
            by_node = {}
            if os.path.exists(report_path):
                for line in open(report_path):
                    try:
                        rec = json.loads(line)
                    except ValueError:
                        continue  # (a line cut short by the child's death)
                    by_node.setdefault(rec["nodeid"], []).append(rec)
            rest = []
            crashed = False
            for it in pending:
                recs = by_node.get(it.nodeid, [])
                complete = any(r["when"] == "teardown" for r in recs) or any(r["when"] == "setup" and r["outcome"] != "passed" for r in recs)
                if complete:
                    _replay(it, recs)
                elif not crashed:
                    # the first test without a complete record is the one the child died in (or never got to, if it died between two)
                    crashed = True
                    why = "timed out after %d s" % MODULE_TIMEOUT_S if rc == -999 else "DEVICE FAULT or crash"
                    if rc == 0:
                        why = "child finished without running this test"
                    _replay(it, [r for r in recs if r["when"] == "setup" and r["outcome"] == "passed"] + _crash_records(it, rc, out, why))
                    tw.line()
                    tw.line("[gpu isolation] %s: child died (rc %d) in %s after %.1f s; the rest of the module runs in a new child" %
                            (os.path.basename(key), rc, it.nodeid, secs), red=True)
                else:
                    rest.append(it)
            pending = rest if crashed else []
            if attempt > 50:  # (a module that kills every child: give up on it, not on the session)
                for it in pending:
                    _replay(it, _crash_records(it, rc, out, "not run: this module's children kept dying"))
                pending = []
    return True
