"""bench.py on the GPU box: the contract fields of the one-line JSON, and a two-rank rehearsal of the N > 1
control flow (sharded seeds, asynchronous gather, max-over-ranks timing).  RCCL refuses two ranks on one device,
so the rehearsal uses the gloo backend with both ranks on GPU 0 (PDOG_BENCH_BACKEND); the RCCL leg itself only
runs on the driver's multi-GPU node.  bench.py is started as a child process (never exec'd from this one)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _run(cmd, extra_env=None):
    env = dict(os.environ, **(extra_env or {}))
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    r = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "256"])
    for k in CONTRACT + ("cpu_baseline",):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["unit"] == "frames/s" and r["vs_baseline"] is None
    assert r["value"] > 0 and abs(r["value"] - 256 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-6 * r["value"]
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb


def test_bench_two_rank_rehearsal():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "128"],
             {"PDOG_BENCH_BACKEND": "gloo"})
    for k in CONTRACT:
        assert k in r, k
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and "cpu_baseline" not in r
    assert abs(r["value"] - 2 * 128 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-6 * r["value"]


def test_plain_c_client_tracks_a_clip(tmp_path):
    """examples/track_clip.c: the reference's frame loop over the C ABI with no Python in between (compiled with gcc,
    run as a child process); exit code 0 = every position within a pixel of the moving disc."""
    sys.path.insert(0, ROOT)
    from pawsometracker_jl_amd import _lib
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "track_clip")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "track_clip.c"),
                           "-o", exe, "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    p = subprocess.run([exe, "80"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stderr[-500:], p.stdout[-200:])
    assert len(p.stdout.split()) == 160
