"""bench.py's static contract (no GPU): every workload names a registered env and an oracle class, the algorithmic bytes
are SURVEY 8(d)'s figures, and the CPU-baseline leg produces the fields the driver reads."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_workloads_are_consistent(oracle):
    import gym_xarm_amd
    b = _bench()
    assert set(b.WORKLOADS) == {"pnp", "reach", "handover", "stack", "handover2"} == set(b.WORKLOAD_NAMES) == set(b.SUBSTEPS)
    survey_bytes = {"pnp": 452, "reach": 336, "handover": 648, "stack": 1040, "handover2": 836}    # SURVEY.md 8(d) (+ its rule for N = 2)
    default_envs = {"pnp": 65536, "reach": 4096, "handover": 16384, "stack": 8192, "handover2": 16384}     # BASELINE.json configs, per GPU
    for w, (env_id, E, act_dim, nbytes, kernel, cls, sample, cfg) in b.WORKLOADS.items():
        assert env_id in gym_xarm_amd.registered_ids()
        assert hasattr(oracle, cls)
        assert nbytes == survey_bytes[w] and E == default_envs[w]
        assert act_dim == (4 if w in ("pnp", "reach") else 8)
    assert b.HBM_PEAK_GBS == 8000.0


def test_cpu_baseline_leg_reports_the_contract_fields(oracle, monkeypatch):
    b = _bench()
    monkeypatch.setitem(b.WORKLOADS, "reach", b.WORKLOADS["reach"][:6] + ((4, 3),) + b.WORKLOADS["reach"][7:])   # tiny sample
    monkeypatch.setattr(os, "cpu_count", lambda: 2)
    out = b.cpu_baseline("reach")
    assert set(out) >= {"value", "unit", "cores", "kind", "sample"}
    assert out["kind"] == "port" and out["cores"] == 2 and out["value"] > 0 and "XarmReach-v0" in out["sample"]


def test_cpu_baseline_sample_is_what_baseline_md_asks_for(oracle):
    """the oracle over 4 096 envs on all cores; the sample string says how it is threaded"""
    b = _bench()
    for w, row in b.WORKLOADS.items():
        assert row[6][0] == 4096, w
    import inspect
    src = inspect.getsource(b.cpu_baseline)
    assert "not OpenMP" in src and "ThreadPoolExecutor" in src


def test_json_line_carries_the_library_identity_and_refuses_variants(tmp_path):
    """xarm_version() / library path / pipeline / solver constants come from the handle, and a TIMING VARIANT
    or XARM_HIP_LIB without --allow-variant-lib ends the run non-zero (checked on the stand-in env: bench.py's own logic)"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, XARM_BENCH_DEVICE="cpu", XARM_BENCH_ENV_FACTORY="tests.bench_stub:make", PYTHONPATH=ROOT)
    env.pop("XARM_HIP_LIB", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--repeats", "1", "--envs-per-gpu", "8", "--no-extras", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    lib = j["library"]
    assert set(lib) >= {"version", "path", "variant", "pipeline", "reset_coop_limit", "step_coop_limit"} and lib["variant"] is False
    assert j["config"]["solver_iterations"] == 50 == lib["pipeline"]["solver_iterations"]
    assert "avg_ms_is" in j["roofline"]["kernels"]["reset"]
    bad = subprocess.run(cmd, env=dict(env, XARM_HIP_LIB="/nonexistent/lib_variant.so"), capture_output=True, text=True, cwd=ROOT)
    assert bad.returncode != 0 and "not the product build" in bad.stderr
    ok = subprocess.run(cmd + ["--allow-variant-lib"], env=dict(env, XARM_HIP_LIB="/nonexistent/lib_variant.so"), capture_output=True, text=True, cwd=ROOT)
    assert ok.returncode == 0 and json.loads([l for l in ok.stdout.splitlines() if l.startswith("{")][-1])["library"]["variant"] is True
