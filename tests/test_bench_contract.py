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
