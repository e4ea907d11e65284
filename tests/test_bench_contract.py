"""bench.py's side of the driver contract, as far as it can be checked without a GPU: the command line the driver uses parses to the
documented defaults, the JSON line carries every key the contract names (read off the source: the line itself needs a device), and
the CPU baseline leg -- the one part of bench.py that runs on the host -- returns what the line promises on a small sample."""
import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline"}
ROOFLINE_KEYS = {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def _dict_keys(tree, name):
    """String keys of the dict literal assigned to `name` anywhere in the module."""
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == name for t in node.targets) and isinstance(node.value, ast.Dict):
            return {k.value for k in node.value.keys if isinstance(k, ast.Constant) and isinstance(k.value, str)}
    return set()


def test_json_line_names_every_contract_key():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    out = _dict_keys(tree, "out")
    assert CONTRACT_KEYS <= out, sorted(CONTRACT_KEYS - out)
    assert {"time_to_stable_cut_ms", "n_ranks_seen", "parity_checked", "repetitions"} <= out
    assert ROOFLINE_KEYS <= _dict_keys(tree, "roofline")


def test_driver_command_lines_parse(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.config == "C3b"  # no flags: N = 1 and a K / W that finish within minutes
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "50", "--warmup", "5"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 50, 5)


def test_cpu_baseline_leg_on_a_small_sample():
    import bench
    from rapid_amd import scenarios as S
    from tests.helpers import oracle_view
    n, K, H, L = 600, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K, list(range(n)))
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario("C2", subj, cfg, n=n, f=6, H=H, L=L)
    nb = np.array([int(np.count_nonzero(sc.records["flags"][sc.rec_off[r]:sc.rec_off[r + 1]] & 1)) for r in range(len(sc.rec_off) - 1)])
    nb = np.maximum(nb, 1)
    base, opt = bench.cpu_baseline(pop, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off, nb, 0.5)
    for d in (base, opt):
        assert set(d) == {"value", "unit", "cores", "kind", "sample"} and d["kind"] == "port" and d["unit"] == "alert-batches/s" and d["value"] > 0
    assert base["cores"] == 1 and opt["cores"] == (os.cpu_count() or 1)


def test_gpus_n_without_a_launcher_starts_its_own_ranks(monkeypatch):
    """`python3 bench.py --gpus N` with WORLD_SIZE unset: the process replaces itself by torch.distributed.run with one rank per GPU
    (the exec is mocked here), rendezvous on 127.0.0.1 with a free port, the original flags passed through, dmabuf IPC selected;
    fewer visible devices than ranks asked for is a non-zero exit, not a line from fewer ranks."""
    import pytest

    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2", "--config", "C4"])
    args = bench.parse()
    seen = {}

    def fake_execve(path, cmd, env):
        seen.update(path=path, cmd=cmd, env=env)

    with pytest.raises(SystemExit):  # (the mocked exec returns)
        bench.self_launch(args, execve=fake_execve, device_count=lambda: 8)
    cmd = seen["cmd"]
    assert seen["path"] == sys.executable and cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) <= 65535
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2", "--config", "C4"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    with pytest.raises(SystemExit) as e:
        bench.self_launch(args, execve=fake_execve, device_count=lambda: 1)
    assert "4" in str(e.value) and "1" in str(e.value)
    # main() takes that path exactly when nobody launched the ranks
    called = []
    monkeypatch.setattr(bench, "self_launch", lambda a: called.append(a.gpus) or (_ for _ in ()).throw(SystemExit(0)))
    with pytest.raises(SystemExit):
        bench.main()
    assert called == [4]
    monkeypatch.setenv("WORLD_SIZE", "2")  # launched, but with another number of ranks: refused, not re-launched
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert called == [4] and "WORLD_SIZE=2" in str(e.value)


def test_the_line_verifies_itself_against_the_oracle():
    """bench.py's untimed parity leg: a sample of receivers spread over a stream set, the kernel's fingerprint restated on the oracle's
    lists, and the comparison itself -- clean results pass, a single wrong field of a single receiver is counted."""
    import bench
    from oracle import pyoracle as O
    from rapid_amd import scenarios as S
    from tests.helpers import oracle_view, proposal_fingerprints
    n, K, H, L = 500, 10, 9, 4
    pop = S.Population.make(n)
    reg, view = oracle_view(pop, K, list(range(n)))
    obs, subj, member = view.tables(n)
    cfg = view.getCurrentConfigurationId()
    sc = S.build_scenario("C2", subj, cfg, n=n, f=5, H=H, L=L)
    idx, recs, off = bench.take_sample(sc.records, sc.rec_off, 24)
    assert len(idx) == 24 and idx[0] == 0 and idx[-1] == len(sc.rec_off) - 2 and len(np.unique(idx)) == 24
    for j, r in enumerate(idx):
        assert np.array_equal(recs[off[j]:off[j + 1]], sc.records[sc.rec_off[r]:sc.rec_off[r + 1]])
    fe, fn, fo, fpp = O.fast_sim_run(n, K, H, L, cfg, obs, subj, member, sc.records, sc.rec_off)
    fp = proposal_fingerprints(fo, fpp, fe >= 0)
    assert np.array_equal(bench.fingerprints_of(fo, fpp, fe >= 0), fp)
    res = [fe.copy(), fn.copy(), np.diff(fo).astype(np.int32), fp.copy()]
    assert bench.parity_mismatches(res, (idx, recs, off), n, K, H, L, cfg, obs, subj, member) == 0
    res[3][idx[5]] ^= np.uint64(1)
    res[0][idx[9]] += 1
    assert bench.parity_mismatches(res, (idx, recs, off), n, K, H, L, cfg, obs, subj, member) == 2
