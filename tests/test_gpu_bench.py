"""bench.py end to end on the GPU box, at sizes that take seconds: the single-GPU line, and the native decomposition
(all domains in this process) with the second, larger `target_box` measurement that N > 1 runs carry."""
import json
import os
import subprocess
import sys

import pytest

from .conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    d = _bench("--cells", "12", "--steps", "10", "--warmup", "4")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["dtype"] == "f64" and d["config"]["atoms"] == 4 * 12 ** 3
    assert d["value"] == pytest.approx(1e3 / d["ms_per_step"], rel=1e-9)
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert d["energy_per_atom"]["kinetic"] > 0.5


def test_native_decomposition_line_and_target_box():
    d = _bench("--domains", "2", "--cells", "12", "--target-cells", "16", "--steps", "8", "--warmup", "4", "--no-cpu-baseline")
    assert d["config"]["parallelism"] == "dd2x1x1" and d["config"]["decomposition"].startswith("native")
    assert d["scaling"] == "strong" and d["config"]["atoms"] == 4 * 12 ** 3
    t = d["target_box"]
    assert "error" not in t, t
    assert t["atoms"] == 4 * 16 ** 3 and t["steps_per_sec"] > 0 and t["energy_per_atom"]["kinetic"] > 0.5
    pr = d["per_rank"]["ranks"]                                    # both in-process domains report their phases
    assert len(pr) == 2 and sum(r["atoms_owned"] for r in pr) == 4 * 12 ** 3
    assert all(r["force_interior_ms"] > 0 and r["halo_ms"] > 0 and r["ghost_fraction"] > 0.05 for r in pr)
    one = _bench("--cells", "12", "--steps", "8", "--warmup", "4", "--no-cpu-baseline")
    # the decomposed box is the same physical system: same energies per atom after the same number of steps
    assert d["energy_per_atom"]["potential"] == pytest.approx(one["energy_per_atom"]["potential"], rel=1e-9)
    assert d["energy_per_atom"]["kinetic"] == pytest.approx(one["energy_per_atom"]["kinetic"], rel=1e-9)


def test_probe_child_speaks_the_protocol():
    """dd_probe.py on its own (one rank: the in-process transport): `ID <hex>` first, `OK ...` last, exit code 0."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "emdee.jl_amd", "dd_probe.py"), "--world", "1", "--rank", "0",
                        "--cells", "12", "--steps", "12"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.split("\n")
    assert lines[0].startswith("ID ") and len(lines[0].split()[1]) == 256
    ok = [l for l in lines if l.startswith("OK ")]
    assert len(ok) == 1 and int(ok[0].split()[1]) == 4 * 12 ** 3 and int(ok[0].split()[4]) >= 2


def test_two_ranks_on_one_gpu_fall_back_to_the_torch_driver():
    """`bench.py --gpus 2` started the way the driver starts it (no launcher), both ranks on this box's one GPU.  RCCL
    refuses two ranks on one device, so the connectivity probe fails on every rank, all of them agree, and the line
    comes from the torch.distributed driver -- the fallback the multi-GPU run depends on, exercised end to end."""
    d = _bench("--gpus", "2", "--share-gpu", "--backend", "gloo", "--cells", "12", "--target-cells", "0", "--steps", "8",
               "--warmup", "4", "--probe-timeout", "120")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["atoms"] == 4 * 12 ** 3
    assert d["config"]["decomposition"].startswith("torch"), d["config"]
    assert d["config"]["decomposition_probe"].startswith("failed"), d["config"]
    assert "probe" in d.get("degraded", ""), d.get("degraded")      # the line says that, and why, the native decomposition did not produce it
    assert "cpu_baseline" not in d or d["cpu_baseline"] is None
    assert d["energy_per_atom"]["kinetic"] > 0.5


def test_two_ranks_over_rccl_on_one_gpu():
    """The native decomposition with two real ranks: `--rccl-loopback` gives each rank its own NCCL_HOSTID, RCCL then
    accepts both on this box's one device and carries the halo over its TCP transport.  The probe must pass, the line
    must come from emdee_dd_* over RCCL send/recv, the second (target) box must be measured, and the physics must be
    that of the undivided box."""
    d = _bench("--gpus", "2", "--share-gpu", "--rccl-loopback", "--cells", "16", "--target-cells", "20",
               "--steps", "8", "--warmup", "4", "--probe-timeout", "240", "--halo-trial-steps", "2")   # torch.distributed on its default backend: RCCL too
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["atoms"] == 4 * 16 ** 3
    assert d["config"]["decomposition"].startswith("native") and "RCCL" in d["config"]["decomposition"], d["config"]
    assert d["config"]["decomposition_probe"].startswith("OK "), d["config"]
    assert "degraded" not in d
    h = d["config"]["halo_exchange"]                               # both forms of the step were tried before the timed run
    assert h["chosen"] in ("overlapped", "in order") and set(h["trial_ms_per_step"]) == {"overlapped", "in order"}
    t = d["target_box"]
    assert "error" not in t, t
    assert t["atoms"] == 4 * 20 ** 3 and t["steps_per_sec"] > 0
    # where each rank's step went: both ranks report, interior + boundary launches (or one launch over all
    # bricks in the in-order form), the halo on its stream, the rebuilds' device and wall-clock time, read-backs, ghost share
    pr = d["per_rank"]["ranks"]
    assert [r["rank"] for r in pr] == [0, 1] and sum(r["atoms_owned"] for r in pr) == 4 * 16 ** 3
    for r in pr:
        assert 0.05 < r["ghost_fraction"] < 0.8 and r["force_interior_ms"] > 0 and r["halo_ms"] > 0
        assert (r["force_boundary_ms"] > 0) == (h["chosen"] == "overlapped")
        assert r["rebuilds"] >= 1 and r["rebuild_wall_ms"] > r["rebuild_device_ms"] > 0 and r["readbacks"] >= r["rebuilds"]
        assert r["readback_wall_ms"] < d["ms_per_step"] and r["force_interior_ms"] + r["force_boundary_ms"] < d["ms_per_step"]
        # a rebuild in the engines' own order costs TWO blocking read-backs: the batch's request words and the build's words, which
        # carry the counts of both exchanges along (round 4: three; round 2: five)
        if r["rebuilds_in_engine_order"] == r["rebuilds"]:
            assert r["readbacks_per_rebuild"] <= 2.0, r
    assert any(r["rebuilds_in_engine_order"] == r["rebuilds"] for r in pr), pr
    # the two trials took 2 x (2 + 2) untimed steps: the undivided run gets them as warm-up
    one = _bench("--cells", "16", "--steps", "8", "--warmup", "12", "--no-cpu-baseline")
    assert d["energy_per_atom"]["potential"] == pytest.approx(one["energy_per_atom"]["potential"], rel=1e-9)
    assert d["energy_per_atom"]["kinetic"] == pytest.approx(one["energy_per_atom"]["kinetic"], rel=1e-9)


def test_target_box_leg_cannot_lose_the_headline_line():
    """--target-timeout 0: the watchdog of the second (target) box fires at once; the line must still be printed, once,
    with the leg marked as timed out, and the ranks must leave with status 0."""
    d = _bench("--gpus", "2", "--share-gpu", "--rccl-loopback", "--backend", "gloo", "--cells", "16", "--target-cells", "20",
               "--steps", "8", "--warmup", "4", "--probe-timeout", "240", "--target-timeout", "0")
    assert d["config"]["decomposition"].startswith("native")
    assert "timed out" in d["target_box"].get("error", ""), d["target_box"]
    assert d["value"] > 0 and d["energy_per_atom"]["kinetic"] > 0.5


def test_worked_examples_run(tmp_path):
    """examples/lj_fluid.py and examples/lj_fluid_decomposed.py (README.md "worked examples") end to end on small boxes: NVT -> NVE,
    XYZ frame, checkpoint restart, the operator on caller arrays; the same fluid cut into four in-process domains."""
    import subprocess
    import sys
    from .conftest import ROOT
    for script, args in (("lj_fluid.py", ["10"]), ("lj_fluid_decomposed.py", ["4", "12"])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)] + args, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, script + ": " + r.stdout[-600:] + r.stderr[-1200:]
        assert "relative energy change" in r.stdout or "energy" in r.stdout, r.stdout[-600:]
