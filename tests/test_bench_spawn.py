"""CPU checks of `python bench.py --gpus N` BEFORE it ever meets an 8-GPU node: the rank-spawn path with
a stubbed device count, the launcher contract (WORLD_SIZE must equal --gpus), and the defaults of the command line — a typo here
must not cost the only multi-GPU lease."""
import importlib.util
import json
import os
import sys

import pytest

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _FakeProc:
    started = []

    def __init__(self, argv, env=None, stdout=None, **kw):
        self.argv, self.env, self.stdout, self.code = argv, env, stdout, 0
        _FakeProc.started.append(self)

    def poll(self):
        return self.code

    def kill(self):
        pass

    def wait(self):
        return self.code


def test_gpus8_spawns_eight_ranks_over_loopback(monkeypatch, capsys):
    bench = _bench()
    import subprocess
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "Popen", _FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RC_BENCH_SHARE_GPU", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    _FakeProc.started = []
    args = bench.parse()
    assert (args.gpus, args.steps, args.warmup) == (8, 20, 5)
    assert bench.spawn_ranks(args) == 0
    procs = _FakeProc.started
    assert len(procs) == 8
    ports = {p.env["MASTER_PORT"] for p in procs}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    for r, p in enumerate(procs):
        assert p.env["RANK"] == str(r) and p.env["LOCAL_RANK"] == str(r) and p.env["WORLD_SIZE"] == "8"
        assert p.env["MASTER_ADDR"] == "127.0.0.1"
        assert p.argv[0] == sys.executable and os.path.samefile(p.argv[1], os.path.join(ROOT, "bench.py"))
        assert p.argv[2:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
        assert (p.stdout is None) == (r == 0)                  # rank 0 alone owns the JSON line
    # a rank that dies: its exit code is the command's, the others are taken down
    _FakeProc.started = []

    class _Dies(_FakeProc):
        def __init__(self, argv, env=None, stdout=None, **kw):
            super().__init__(argv, env=env, stdout=stdout)
            self.code = 3 if env["RANK"] == "5" else None
            self.killed = False

        def kill(self):
            self.killed = True
            self.code = -9
    monkeypatch.setattr(subprocess, "Popen", _Dies)
    assert bench.spawn_ranks(args) == 3
    assert all(p.killed for p in _FakeProc.started if p.env["RANK"] != "5")


def test_fewer_gpus_than_asked_prints_one_skipped_line(monkeypatch, capsys):
    bench = _bench()
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("RC_BENCH_SHARE_GPU", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    assert bench.spawn_ranks(bench.parse()) == 0
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["value"] is None and "skipped" in line


def test_launcher_contract_and_defaults(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and a.adc_k == 1000 and a.batch == 49152
    # under a launcher (WORLD_SIZE set) --gpus must equal the world size: refused before any GPU work
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    # stub what main() imports before the check, so that the test needs neither a GPU nor the HIP library
    saved = os.dup(1)                                           # main() points fd 1 at stderr for the libraries it loads
    try:
        with pytest.raises(SystemExit) as e:
            bench.main()
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    assert "WORLD_SIZE=4" in str(e.value)
    # the recipe's global batch divides by every rank count the driver launches, and the fixtures the N > 1 self-check reads exist
    for n in (1, 2, 4, 8):
        assert a.batch % n == 0
    for f in ("headline_b49152_m48_sample.npz", "recipe8_b49152_m48_sample.npz"):
        assert os.path.exists(os.path.join(ROOT, "tests", "golden", f))
