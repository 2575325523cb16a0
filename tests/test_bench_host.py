"""Host logic of bench.py and of the strong-scaling box generator (no GPU): the self-launch of
`python bench.py --gpus N`, the cut of ONE lattice into rank blocks, and the algorithmic byte counts."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from .conftest import ROOT


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def bench():
    return _load("emdee_bench", os.path.join(ROOT, "bench.py"))


def test_self_launch_starts_a_child_torchrun(bench, monkeypatch):
    """`python bench.py --gpus 8` (the form the driver used at N = 1) must start the ranks itself, as a CHILD process,
    before anything touches the GPU (VERDICT r1 missing #2); a failed native run is repeated on the torch driver, which
    is told why it runs (its line carries `degraded`)."""
    import io
    import subprocess
    seen = []

    class FakeChild:
        pid = 12345

        def __init__(self, cmd, env=None, start_new_session=False, stdout=None, text=None):
            seen.append((cmd, env, start_new_session))
            self.stdout = io.StringIO("")

        def wait(self, timeout=None):
            return 7 if len(seen) == 1 else 0
    monkeypatch.setattr(subprocess, "Popen", FakeChild)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("EMDEE_BENCH_DEADLINE_AT", raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0                                   # second attempt's exit code is relayed
    cmd, env, new_session = seen[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and new_session
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert float(env["EMDEE_BENCH_DEADLINE_AT"]) < bench.T_PROCESS_START + 540.0     # the ranks inherit ONE deadline, below the driver's 600 s
    assert len(seen) == 2 and seen[1][0][-4:-1] == ["--dd", "torch", "--degraded"]    # native run failed (status 7): torch driver next
    assert "status 7" in seen[1][0][-1]


def _launch(mode, *flags, timeout=120):
    """bench.py --gpus 2 as the driver starts it, with tests/helpers/fake_bench_ranks.py standing in for the ranks."""
    import json
    import subprocess
    import time
    env = dict(os.environ, FAKE_RANKS_MODE=mode, EMDEE_BENCH_RANK_SCRIPT=os.path.join(ROOT, "tests", "helpers", "fake_bench_ranks.py"))
    env.pop("WORLD_SIZE", None)
    env.pop("EMDEE_BENCH_DEADLINE_AT", None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *flags], capture_output=True, text=True,
                       timeout=timeout, env=env)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines, time.monotonic() - t0


def test_a_stalled_native_run_ends_as_one_degraded_line_inside_the_deadline():
    """a native run that never finishes is stopped when its share of the ONE deadline is used up, the torch
    driver gets the rest, and its line says why it exists.  Nothing runs past the deadline."""
    r, lines, took = _launch("stall", "--deadline", "45", "--launch-timeout", "8", "--retry-min", "10")
    assert r.returncode == 0, r.stderr[-1500:]
    assert len(lines) == 1 and "status 124" in lines[0]["degraded"]
    assert took < 45.0
    assert "once more with --dd torch" in r.stderr


def test_no_second_attempt_without_time_for_it():
    r, lines, took = _launch("fail", "--deadline", "30", "--retry-min", "200")
    assert r.returncode != 0 and lines == [] and "no second attempt" in r.stderr     # (torch.distributed.run reports a failed rank as status 1)
    assert took < 30.0


def test_a_failed_native_run_is_repeated_and_marked():
    r, lines, _ = _launch("fail", "--deadline", "60", "--retry-min", "5")
    assert r.returncode == 0 and len(lines) == 1 and "status" in lines[0]["degraded"]
    # the ranks of both attempts were given the same absolute deadline, inside the launcher's own
    assert lines[0]["deadline_at"] > 0


def test_a_printed_line_is_never_followed_by_a_second_run():
    """ADVICE r2 (bench.py:92): rank 0 prints the line and then hangs (a final barrier nobody reaches): the launcher stops
    the ranks at the deadline, reports success -- the measurement is out -- and does NOT start the torch driver."""
    r, lines, took = _launch("late", "--deadline", "25", "--retry-min", "1")
    assert r.returncode == 0 and len(lines) == 1 and "degraded" not in lines[0] and lines[0]["exit_status"] == 124
    assert "once more" not in r.stderr and took < 25.0


def test_a_crash_after_the_line_shows_in_the_line():
    """ADVICE r3 (bench.py:222): the measurement is complete once its line exists, so the launcher still reports success and
    starts nothing again -- but the rank group's abnormal end is in the run's records: top-level exit_status != 0 (a normal
    run carries exit_status 0), and a note on stderr."""
    r, lines, _ = _launch("crash", "--deadline", "60", "--retry-min", "1")
    assert r.returncode == 0 and len(lines) == 1 and lines[0]["exit_status"] != 0
    assert "AFTER its result line" in r.stderr and "once more" not in r.stderr
    r, lines, _ = _launch("twice", "--deadline", "60")
    assert lines[0]["exit_status"] == 0


def test_only_one_line_reaches_stdout():
    r, lines, _ = _launch("twice", "--deadline", "60")
    assert r.returncode == 0 and len(lines) == 1


def test_defaults_are_the_baseline_config(bench, monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert (a.gpus, a.cells, a.precision, a.rc, a.scaling) == (1, 136, "f64", 2.5, "strong")
    assert bench.algorithmic_bytes_per_atom_step(8, 2.5) == pytest.approx(377.4, abs=0.05)
    assert bench.algorithmic_bytes_per_atom_step(4, 2.5) == pytest.approx(293.4, abs=0.05)
    assert bench.algorithmic_bytes_per_atom_step(8, 3.5) == pytest.approx(742.7, abs=0.05)


@pytest.mark.parametrize("world", [1, 2, 4, 8, 6])
@pytest.mark.parametrize("cells", [8, 9])
def test_strong_scaling_blocks_tile_the_lattice(emdee_synthetic, world, cells):
    """Every atom of the ONE cells^3 box is generated by exactly one rank, whatever the rank count, with the
    coordinates the undivided generator gives it."""
    domain = _load("emdee_domain_host", os.path.join(ROOT, "emdee.jl_amd", "domain.py"))
    syn = emdee_synthetic
    grid = domain.rank_grid(world)
    ref, L = syn.fcc_positions(cells)
    seen = np.zeros(ref.shape[0], dtype=np.int32)
    for rank in range(world):
        coords = (rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1]))
        ncells, lo, n = domain.lattice_block(cells, grid, coords, "strong")
        assert ncells == (cells,) * 3
        pos, gid, lengths = syn.fcc_block(ncells, lo, n)
        assert lengths[0] == pytest.approx(L)
        seen[gid] += 1
        assert np.array_equal(pos, ref[gid])
    assert (seen == 1).all()
    # weak scaling keeps the per-rank block and grows the lattice
    ncells, lo, n = domain.lattice_block(cells, grid, (0, 0, 0), "weak")
    assert ncells == tuple(cells * g for g in grid) and n == [cells] * 3


# ---------------------------------------------------------------------------------------------------------------------
# the connectivity probe of the native decomposition (emdee.jl_amd/dd_probe.py, dd.probe_over_rccl): on a box without a
# GPU every child fails, and what is tested is the agreement -- all ranks return False together, quickly, with the
# child's reason in the note, and no child is left running
def _probe_worker(rank, world, port, out_dir):
    import json
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package
        pkg = load_package()
        t0 = time.monotonic()
        ok, note = pkg.dd.probe_over_rccl(world, rank, 0, dist, timeout=90.0, cells=12)
        with open(os.path.join(out_dir, "probe%d.json" % rank), "w") as f:
            json.dump({"ok": ok, "note": note, "seconds": time.monotonic() - t0}, f)
    finally:
        dist.destroy_process_group()


def test_probe_agrees_on_failure_without_a_gpu(tmp_path):
    import json
    import socket
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("CPU-side behaviour: with a GPU the probe is covered by tests/test_gpu_bench.py")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_probe_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [json.load(open(tmp_path / ("probe%d.json" % r))) for r in range(2)]
    assert [r["ok"] for r in res] == [False, False]
    assert all(r["seconds"] < 80.0 for r in res)
    assert any("probe" in r["note"] for r in res)


def test_a_line_only_carries_traffic_measured_on_its_own_configuration():
    """profiles/traffic.json is keyed by (atoms per GPU, dtype, rc, one or two species) and bench.py
    attaches an entry on an exact match only; an entry below the line's algorithmic bytes is refused.  Checked on the
    committed file for every BASELINE configuration bench.py can be asked for."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    entries = bench.load_traffic_entries()
    assert entries, "profiles/traffic.json holds the headline configuration at least"
    keys = [(int(e["atoms"]), e["dtype"], float(e.get("rc", 2.5)), bool(e.get("mixture", False))) for e in entries]
    assert len(set(keys)) == len(keys)
    N = 10061824
    nbar = lambda rc: (4.0 / 3.0) * 3.141592653589793 * rc ** 3 * 0.8
    alg = lambda w, rc: bench.traffic_floor(N, w, rc)            # 21 words per atom + one uint16 entry per in-cutoff neighbour
    assert alg(8, 2.5) == pytest.approx(N * (21 * 8 + 2 * nbar(2.5)))
    head = bench.pick_traffic(entries, N, "f64", 2.5, False, alg(8, 2.5))
    assert head is not None and head["lj_force_nbr_bytes_per_launch"] >= N * (21 * 8 + 4 * nbar(2.5))   # fp64: above SURVEY's figure too
    for atoms, dtype, rc, mix, w in ((N, "f64", 3.5, True, 8), (N, "f32", 2.5, False, 4), (1000188, "f64", 2.5, False, 8),
                                     (N, "f64", 2.5, True, 8), (N, "f64", 3.5, False, 8), (100615028, "f64", 2.5, False, 8)):
        t = bench.pick_traffic(entries, atoms, dtype, rc, mix, bench.traffic_floor(atoms, w, rc))
        if t is not None:
            assert (int(t["atoms"]), t["dtype"], float(t["rc"]), bool(t["mixture"])) == (atoms, dtype, rc, mix)
            assert t["lj_force_nbr_bytes_per_launch"] >= bench.traffic_floor(atoms, w, rc)
    # the round-3 mistake, replayed: the single-species figure offered to the mixture line is not taken
    only_head = [dict(head)]
    assert bench.pick_traffic(only_head, N, "f64", 3.5, True, alg(8, 3.5)) is None
    assert bench.pick_traffic(only_head, N, "f32", 2.5, False, alg(4, 2.5)) is None
    # and an entry that claims less than the algorithmic bytes is refused even on its own key
    low = [dict(head, lj_force_nbr_bytes_per_launch=int(0.9 * alg(8, 2.5)))]     # (what round 3 attached to the mixture line was 0.56 of its algorithmic bytes)
    assert bench.pick_traffic(low, N, "f64", 2.5, False, alg(8, 2.5)) is None


def test_a_line_only_carries_the_issue_floor_of_its_own_kernel():
    """VERDICT r4 missing #3: profiles/valu.json is keyed like traffic.json -- (atoms per GPU, dtype, rc, one or two species) --
    and bench.py attaches `valu_issue` on an exact match only: the instruction count of the single-species fp64 kernel is not
    offered to the fp32 line or to the mixture's typed kernel (round 4 dropped `valu_issue` for mixtures instead of
    measuring theirs)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    entries = bench.load_valu_entries()
    assert entries, "profiles/valu.json holds the headline configuration at least"
    keys = [(int(e["atoms"]), e["dtype"], float(e.get("rc", 2.5)), bool(e.get("mixture", False))) for e in entries]
    assert len(set(keys)) == len(keys)
    N = 10061824
    head = bench.pick_valu(entries, N, "f64", 2.5, False)
    assert head is not None and 0.5 < head["issue_floor_ms"] < 1.5 and head["valu_insts_per_launch"] > 1e8
    for atoms, dtype, rc, mix in ((N, "f64", 3.5, True), (N, "f32", 2.5, False), (1000188, "f64", 2.5, False), (N, "f64", 2.5, True)):
        t = bench.pick_valu(entries, atoms, dtype, rc, mix)
        if t is not None:
            assert (int(t["atoms"]), t["dtype"], float(t["rc"]), bool(t["mixture"])) == (atoms, dtype, rc, mix)
            assert t["issue_floor_ms"] > 0 and ("k_typed" in t["kernel"]) == mix
    only_head = [dict(head)]
    assert bench.pick_valu(only_head, N, "f64", 3.5, True) is None and bench.pick_valu(only_head, N, "f32", 2.5, False) is None
    assert bench.pick_valu([dict(head, issue_floor_ms=None)], N, "f64", 2.5, False) is None
