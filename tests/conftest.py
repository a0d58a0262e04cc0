import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def parity():
    from oracle import parity as PR
    return PR


@pytest.fixture(scope="session")
def golden_reward():
    return np.load(os.path.join(GOLDEN, "pnp_reward_reference.npz"))


@pytest.fixture(scope="session")
def golden_rollout():
    return np.load(os.path.join(GOLDEN, "pnp_oracle_rollout.npz"))


class HostCore:
    """g++ instantiation of csrc/xarm_core.h (tests/hostbuild) - CPU-side unit tests only."""

    def __init__(self):
        d = os.path.join(ROOT, "tests", "hostbuild")
        so = os.path.join(d, "libxarm_host.so")
        srcs = [os.path.join(d, "xarm_host.cpp")] + [os.path.join(ROOT, "gym_xarm_amd", "csrc", f) for f in (
            "xarm_core.h", "xarm7_pd_model.h", "xarm_reach_core.h", "xarm7_reach_model.h", "xarm_handover_core.h", "xarm_handover2_core.h", "xarm_stack_core.h",
            "xarm_coop_core.h", "xarm_reach_coop_core.h", "xarm_handover_coop_core.h")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas",
                                   "-o", so, srcs[0]])
        self.L = C.CDLL(so)

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.POINTER(C.c_double))

    @staticmethod
    def _u8(a):
        return a.ctypes.data_as(C.POINTER(C.c_uint8))

    @staticmethod
    def _cfg(seed=0, off=0, igr=0.0, ggr=0.0, gs=0, rt=0):
        return (C.c_uint64(seed), C.c_int64(off), C.c_double(igr), C.c_double(ggr), C.c_int(gs), C.c_int(rt))

    def init(self, E, f32=1, **kw):
        st = np.zeros((E, 54))
        self.L.xh_init(C.c_int(f32), *self._cfg(**kw), C.c_int64(E), self._p(st))
        return st

    def step(self, state, actions, f32=1, **kw):
        E = state.shape[0]
        st = np.array(state, dtype=np.float64, copy=True)
        a = np.ascontiguousarray(actions, dtype=np.float64)
        obs, ag, dg = np.zeros((E, 24)), np.zeros((E, 3)), np.zeros((E, 3))
        rew, done, succ = np.zeros(E), np.zeros(E, np.uint8), np.zeros(E, np.uint8)
        self.L.xh_step(C.c_int(f32), *self._cfg(**kw), C.c_int64(E), self._p(st), self._p(a), self._p(obs),
                       self._p(ag), self._p(dg), self._p(rew), self._u8(done), self._u8(succ))
        return st, obs, ag, dg, rew, done, succ

    def step_fast(self, state, actions, f32=1, **kw):
        """xk::env_step_fast: the pad-free fast step; ok[e] = 0 -> a finger-pad row was active, row e is returned untouched"""
        E = state.shape[0]
        st = np.array(state, dtype=np.float64, copy=True)
        a = np.ascontiguousarray(actions, dtype=np.float64)
        obs, ag, dg = np.zeros((E, 24)), np.zeros((E, 3)), np.zeros((E, 3))
        rew, done, succ, ok = np.zeros(E), np.zeros(E, np.uint8), np.zeros(E, np.uint8), np.zeros(E, np.uint8)
        self.L.xh_step_fast(C.c_int(f32), *self._cfg(**kw), C.c_int64(E), self._p(st), self._p(a), self._p(obs),
                            self._p(ag), self._p(dg), self._p(rew), self._u8(done), self._u8(succ), self._u8(ok))
        return st, obs, ag, dg, rew, done, succ, ok.astype(bool)

    def coop_step(self, state, actions, f32=1, **kw):
        """xc::env_step: the 16-lanes-per-env impulse-space core (csrc/xarm_coop_core.h)"""
        E = state.shape[0]
        st = np.array(state, dtype=np.float64, copy=True)
        a = np.ascontiguousarray(actions, dtype=np.float64)
        obs, ag, dg = np.zeros((E, 24)), np.zeros((E, 3)), np.zeros((E, 3))
        rew, done, succ = np.zeros(E), np.zeros(E, np.uint8), np.zeros(E, np.uint8)
        self.L.xh_coop_step(C.c_int(f32), *self._cfg(**kw), C.c_int64(E), self._p(st), self._p(a), self._p(obs),
                            self._p(ag), self._p(dg), self._p(rew), self._u8(done), self._u8(succ))
        return st, obs, ag, dg, rew, done, succ

    def coop_reset(self, state, mask=None, f32=1, **kw):
        E = state.shape[0]
        st = np.array(state, dtype=np.float64, copy=True)
        obs, ag, dg = np.zeros((E, 24)), np.zeros((E, 3)), np.zeros((E, 3))
        mk = None if mask is None else self._u8(np.ascontiguousarray(mask, dtype=np.uint8))
        self.L.xh_coop_reset(C.c_int(f32), *self._cfg(**kw), C.c_int64(E), self._p(st), mk, self._p(obs), self._p(ag), self._p(dg))
        return st, obs, ag, dg

    def coop_substep(self, state, qt, n, f32=1):
        st = np.array(state, dtype=np.float64, copy=True)
        self.L.xh_coop_substep(C.c_int(f32), C.c_int64(st.shape[0]), self._p(st), self._p(np.ascontiguousarray(qt, dtype=np.float64)), C.c_int(n))
        return st

    def step_lazy(self, state, actions, f32=1, **kw):
        """lazy auto-reset step; the returned `done` array holds the phase (0 / 1 / 2)"""
        E = state.shape[0]
        st = np.array(state, dtype=np.float64, copy=True)
        a = np.ascontiguousarray(actions, dtype=np.float64)
        obs, ag, dg = np.zeros((E, 24)), np.zeros((E, 3)), np.zeros((E, 3))
        rew, done, succ = np.zeros(E), np.zeros(E, np.uint8), np.zeros(E, np.uint8)
        self.L.xh_step_lazy(C.c_int(f32), *self._cfg(**kw), C.c_int64(E), self._p(st), self._p(a), self._p(obs),
                            self._p(ag), self._p(dg), self._p(rew), self._u8(done), self._u8(succ))
        return st, obs, ag, dg, rew, done, succ

    def reset(self, state, mask=None, f32=1, **kw):
        E = state.shape[0]
        st = np.array(state, dtype=np.float64, copy=True)
        obs, ag, dg = np.zeros((E, 24)), np.zeros((E, 3)), np.zeros((E, 3))
        mk = None if mask is None else self._u8(np.ascontiguousarray(mask, dtype=np.uint8))
        self.L.xh_reset(C.c_int(f32), *self._cfg(**kw), C.c_int64(E), self._p(st), mk, self._p(obs), self._p(ag), self._p(dg))
        return st, obs, ag, dg

    def substep(self, state, qt, n, f32=1):
        st = np.array(state, dtype=np.float64, copy=True)
        self.L.xh_substep(C.c_int(f32), C.c_int64(st.shape[0]), self._p(st), self._p(np.ascontiguousarray(qt, dtype=np.float64)), C.c_int(n))
        return st

    def reach_init(self, E, f32=1, seed=0, off=0, rt=0):
        st = np.zeros((E, 45))
        self.L.xh_reach_init(C.c_int(f32), C.c_uint64(seed), C.c_int64(off), C.c_int(rt), C.c_int64(E), self._p(st))
        return st

    def reach_step(self, state, actions, f32=1, seed=0, off=0, rt=0, coop=False):
        """coop=True: the 16-lanes-per-env core (csrc/xarm_reach_coop_core.h)"""
        E = state.shape[0]
        st = np.array(state, dtype=np.float64, copy=True)
        a = np.ascontiguousarray(actions, dtype=np.float64)
        obs, ag, dg = np.zeros((E, 8)), np.zeros((E, 3)), np.zeros((E, 3))
        rew, done, succ, fut = np.zeros(E), np.zeros(E, np.uint8), np.zeros(E, np.uint8), np.zeros(E, np.int32)
        (self.L.xh_reach_coop_step if coop else self.L.xh_reach_step)(C.c_int(f32), C.c_uint64(seed), C.c_int64(off), C.c_int(rt), C.c_int64(E), self._p(st), self._p(a),
                             self._p(obs), self._p(ag), self._p(dg), self._p(rew), self._u8(done), self._u8(succ),
                             fut.ctypes.data_as(C.POINTER(C.c_int32)))
        return st, obs, ag, dg, rew, done, succ, fut

    def reach_reset(self, state, mask=None, f32=1, seed=0, off=0, rt=0, coop=False):
        E = state.shape[0]
        st = np.array(state, dtype=np.float64, copy=True)
        obs, ag, dg = np.zeros((E, 8)), np.zeros((E, 3)), np.zeros((E, 3))
        mk = None if mask is None else self._u8(np.ascontiguousarray(mask, dtype=np.uint8))
        (self.L.xh_reach_coop_reset if coop else self.L.xh_reach_reset)(C.c_int(f32), C.c_uint64(seed), C.c_int64(off), C.c_int(rt), C.c_int64(E), self._p(st), mk,
                              self._p(obs), self._p(ag), self._p(dg))
        return st, obs, ag, dg

    def st_init(self, E, f32=1, seed=0, off=0, rt=0):
        st = np.zeros((E, 136))
        self.L.xh_st_init(C.c_int(f32), C.c_uint64(seed), C.c_int64(off), C.c_int(rt), C.c_int64(E), self._p(st))
        return st

    def st_step(self, state, actions, f32=1, seed=0, off=0, rt=0):
        E = state.shape[0]
        st = np.array(state, dtype=np.float64, copy=True)
        a = np.ascontiguousarray(actions, dtype=np.float64)
        obs, ag, dg = np.zeros((E, 55)), np.zeros((E, 9)), np.zeros((E, 9))
        rew, done, succ = np.zeros(E), np.zeros(E, np.uint8), np.zeros(E, np.uint8)
        self.L.xh_st_step(C.c_int(f32), C.c_uint64(seed), C.c_int64(off), C.c_int(rt), C.c_int64(E), self._p(st), self._p(a),
                          self._p(obs), self._p(ag), self._p(dg), self._p(rew), self._u8(done), self._u8(succ))
        return st, obs, ag, dg, rew, done, succ

    def st_reset(self, state, mask=None, f32=1, seed=0, off=0, rt=0):
        E = state.shape[0]
        st = np.array(state, dtype=np.float64, copy=True)
        obs, ag, dg = np.zeros((E, 55)), np.zeros((E, 9)), np.zeros((E, 9))
        mk = None if mask is None else self._u8(np.ascontiguousarray(mask, dtype=np.uint8))
        self.L.xh_st_reset(C.c_int(f32), C.c_uint64(seed), C.c_int64(off), C.c_int(rt), C.c_int64(E), self._p(st), mk,
                           self._p(obs), self._p(ag), self._p(dg))
        return st, obs, ag, dg

    def cube_cube(self, pA, RA, pB, RB, h=0.025, margin=0.005, f32=0):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (pA, RA, pB, RB)]
        pts, nrm, dist = np.zeros((4, 3)), np.zeros(3), np.zeros(4)
        self.L.xh_cube_cube.restype = C.c_int
        n = self.L.xh_cube_cube(C.c_int(f32), self._p(a[0]), self._p(a[1]), self._p(a[2]), self._p(a[3]), C.c_double(h), C.c_double(margin),
                                self._p(pts), self._p(nrm), self._p(dist))
        return pts[:n].copy(), nrm, dist[:n].copy()

    def ho_init(self, E, f32=1, seed=0, off=0, ssr=0.5, gs=1):
        st = np.zeros((E, 76))
        self.L.xh_ho_init(C.c_int(f32), C.c_uint64(seed), C.c_int64(off), C.c_double(ssr), C.c_int(gs), C.c_int64(E), self._p(st))
        return st

    def ho_step(self, state, actions, f32=1, seed=0, off=0, ssr=0.5, gs=1, rt=0, use_stand=0):
        E = state.shape[0]
        self.L.xh_ho_set_reward_type(C.c_int(rt))
        self.L.xh_ho_set_use_stand(C.c_int(use_stand))
        st = np.array(state, dtype=np.float64, copy=True)
        a = np.ascontiguousarray(actions, dtype=np.float64)
        obs, ag, dg = np.zeros((E, 29)), np.zeros((E, 3)), np.zeros((E, 3))
        rew, done, succ = np.zeros(E), np.zeros(E, np.uint8), np.zeros(E, np.uint8)
        self.L.xh_ho_step(C.c_int(f32), C.c_uint64(seed), C.c_int64(off), C.c_double(ssr), C.c_int(gs), C.c_int64(E), self._p(st),
                          self._p(a), self._p(obs), self._p(ag), self._p(dg), self._p(rew), self._u8(done), self._u8(succ))
        return st, obs, ag, dg, rew, done, succ

    def ho_reset(self, state, mask=None, f32=1, seed=0, off=0, ssr=0.5, gs=1):
        E = state.shape[0]
        st = np.array(state, dtype=np.float64, copy=True)
        obs, ag, dg = np.zeros((E, 29)), np.zeros((E, 3)), np.zeros((E, 3))
        mk = None if mask is None else self._u8(np.ascontiguousarray(mask, dtype=np.uint8))
        self.L.xh_ho_reset(C.c_int(f32), C.c_uint64(seed), C.c_int64(off), C.c_double(ssr), C.c_int(gs), C.c_int64(E), self._p(st), mk,
                           self._p(obs), self._p(ag), self._p(dg))
        return st, obs, ag, dg

    def hoc_step(self, state, actions, f32=1, seed=0, off=0, ssr=0.5, gs=1, rt=0, use_stand=0, mode="coop", stages=3):
        """XarmHandover.step on the cooperative rows (csrc/xarm_handover_coop_core.h; mode 'coop', 'coupled' = every substep
        forced through the coupled sweep) or on the pad-free fast lane pair (mode 'fast'); returns ok[e] last: False = a pad
        row was active during the fast step and row e came back untouched.  mode 'staged': the staged pipeline of xarm_step
        with `stages` fast stages; the last return value is then 0 (finished on the fast path) or 1 + the stage that handed off"""
        E = state.shape[0]
        self.L.xh_ho_set_reward_type(C.c_int(rt))
        self.L.xh_ho_set_use_stand(C.c_int(use_stand))
        st = np.array(state, dtype=np.float64, copy=True)
        a = np.ascontiguousarray(actions, dtype=np.float64)
        obs, ag, dg = np.zeros((E, 29)), np.zeros((E, 3)), np.zeros((E, 3))
        rew, done, succ, ok = np.zeros(E), np.zeros(E, np.uint8), np.zeros(E, np.uint8), np.ones(E, np.uint8)
        self.L.xh_hoc_step(C.c_int(f32), C.c_int({"coop": 0, "coupled": 2, "fast": 4, "staged": 4 + int(stages)}[mode]), C.c_uint64(seed), C.c_int64(off), C.c_double(ssr), C.c_int(gs),
                           C.c_int64(E), self._p(st), self._p(a), self._p(obs), self._p(ag), self._p(dg), self._p(rew), self._u8(done), self._u8(succ), self._u8(ok))
        return st, obs, ag, dg, rew, done, succ, (ok.astype(np.int64) if mode == "staged" else ok.astype(bool))

    def hoc_reset(self, state, mask=None, f32=1, seed=0, off=0, ssr=0.5, gs=1, use_stand=0, forced=False):
        E = state.shape[0]
        self.L.xh_ho_set_use_stand(C.c_int(use_stand))
        st = np.array(state, dtype=np.float64, copy=True)
        obs, ag, dg = np.zeros((E, 29)), np.zeros((E, 3)), np.zeros((E, 3))
        mk = None if mask is None else self._u8(np.ascontiguousarray(mask, dtype=np.uint8))
        self.L.xh_hoc_reset(C.c_int(f32), C.c_int(1 if forced else 0), C.c_uint64(seed), C.c_int64(off), C.c_double(ssr), C.c_int(gs), C.c_int64(E), self._p(st), mk,
                            self._p(obs), self._p(ag), self._p(dg))
        return st, obs, ag, dg

    # ---- Handover with num_obj = 2 (csrc/xarm_handover2_core.h)
    def ho2_init(self, E, f32=1, seed=0, off=0, ssr=0.5, gs=1):
        st = np.zeros((E, 100))
        self.L.xh_ho2_init(C.c_int(f32), C.c_uint64(seed), C.c_int64(off), C.c_double(ssr), C.c_int(gs), C.c_int64(E), self._p(st))
        return st

    def ho2_step(self, state, actions, f32=1, seed=0, off=0, ssr=0.5, gs=1, use_stand=0):
        E = state.shape[0]
        self.L.xh_ho_set_use_stand(C.c_int(use_stand))
        st = np.array(state, dtype=np.float64, copy=True)
        a = np.ascontiguousarray(actions, dtype=np.float64)
        obs, ag, dg = np.zeros((E, 42)), np.zeros((E, 6)), np.zeros((E, 6))
        rew, done, succ = np.zeros(E), np.zeros(E, np.uint8), np.zeros(E, np.uint8)
        self.L.xh_ho2_step(C.c_int(f32), C.c_uint64(seed), C.c_int64(off), C.c_double(ssr), C.c_int(gs), C.c_int64(E), self._p(st),
                           self._p(a), self._p(obs), self._p(ag), self._p(dg), self._p(rew), self._u8(done), self._u8(succ))
        return st, obs, ag, dg, rew, done, succ

    def ho2_reset(self, state, mask=None, f32=1, seed=0, off=0, ssr=0.5, gs=1):
        E = state.shape[0]
        st = np.array(state, dtype=np.float64, copy=True)
        obs, ag, dg = np.zeros((E, 42)), np.zeros((E, 6)), np.zeros((E, 6))
        mk = None if mask is None else self._u8(np.ascontiguousarray(mask, dtype=np.uint8))
        self.L.xh_ho2_reset(C.c_int(f32), C.c_uint64(seed), C.c_int64(off), C.c_double(ssr), C.c_int(gs), C.c_int64(E), self._p(st), mk,
                            self._p(obs), self._p(ag), self._p(dg))
        return st, obs, ag, dg

    def class_order(self, key, group=32):
        """(order, aligned): the StackTower step kernel's visiting order for these class keys"""
        k = np.ascontiguousarray(key, dtype=np.uint8)
        order = np.zeros(k.shape[0], np.int32)
        self.L.xh_class_order.restype = C.c_int
        rc = self.L.xh_class_order(k.ctypes.data_as(C.c_void_p), C.c_int64(k.shape[0]), C.c_int(group), order.ctypes.data_as(C.c_void_p))
        assert rc >= 0, "class_slot produced an out-of-range or duplicate slot"
        return order, bool(rc)

    def box_box(self, pA, RA, hA, pB, RB, hB, margin=0.005, f32=0):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (pA, RA, hA, pB, RB, hB)]
        pts, nrm, dist = np.zeros((4, 3)), np.zeros(3), np.zeros(4)
        self.L.xh_box_box.restype = C.c_int
        n = self.L.xh_box_box(C.c_int(f32), *[self._p(x) for x in a], C.c_double(margin), self._p(pts), self._p(nrm), self._p(dist))
        return pts[:n].copy(), nrm, dist[:n].copy()

    def ik(self, q, target, f32=1):
        out = np.zeros(9)
        self.L.xh_ik(C.c_int(f32), self._p(np.ascontiguousarray(q, dtype=np.float64)),
                     self._p(np.ascontiguousarray(target, dtype=np.float64)), self._p(out))
        return out


class OracleTorchEnv:
    """the CPU oracle behind the torch VecEnv call surface the scripted policies use (reset / step / get_state over CPU
    tensors, auto_reset off) - lets gym_xarm_amd/policies.py run unchanged on the oracle and on the HIP env"""

    def __init__(self, ora):
        import torch
        self.torch, self.ora, self.num_envs, self.device = torch, ora, ora.E, torch.device("cpu")

    def _obs(self, o):
        t = self.torch
        return {"observation": t.tensor(o[0], dtype=t.float32), "achieved_goal": t.tensor(o[1], dtype=t.float32),
                "desired_goal": t.tensor(o[2], dtype=t.float32)}

    def reset(self):
        return self._obs(self.ora.reset())

    def step(self, a):
        t = self.torch
        o = self.ora.step(a.double().numpy())
        return self._obs(o), t.tensor(o[3], dtype=t.float32), t.tensor(o[4]), {"is_success": t.tensor(o[5])}

    def get_state(self):
        return self.torch.tensor(self.ora.get_state(), dtype=self.torch.float32)


class ShardedOracleHandover:
    """W OracleHandover instances on W threads (ctypes releases the GIL): the live fixtures of the Handover GPU tests cost
    ~65 000 oracle env steps each"""

    def __init__(self, oracle, E, W=8, **kw):
        from concurrent.futures import ThreadPoolExecutor
        assert E % W == 0
        self.n, self.W = E // W, W
        self.o = [oracle.OracleHandover(self.n, env_id_offset=k * self.n, **kw) for k in range(W)]
        self.ex = ThreadPoolExecutor(W)

    def _map(self, fn):
        return list(self.ex.map(fn, range(self.W)))

    def reset(self):
        return np.concatenate([r[0] for r in self._map(lambda k: self.o[k].reset())])

    def get_state(self):
        return np.concatenate([o.get_state() for o in self.o])

    def step_from(self, st, a):
        """(outputs of step, next state) from the injected state"""
        def run(k):
            sl = slice(k * self.n, (k + 1) * self.n)
            self.o[k].set_state(st[sl])
            out = self.o[k].step(a[sl])
            return out, self.o[k].get_state()
        r = self._map(run)
        return [np.concatenate([x[0][i] for x in r]) for i in range(6)], np.concatenate([x[1] for x in r])

    def sens(self, st, a, nxt, cont, quats, seed, draws=(1e-6, 1e-6, 1e-6, 1e-7, 1e-7, 1e-7)):
        """the oracle's own response to a perturbation of the input state, per env.  Six draws at two amplitudes: a stick held
        by sliding pads answers a 1e-7 perturbation with 0.2 in float64 on transitions where two draws at 1e-6 saw 3e-3
        (tests/tools/ho_outliers.py) - the probe has to sample the contact discontinuities it is there to detect"""
        sens = np.zeros(st.shape[0])
        for j, eps in enumerate(draws):
            sp = st.copy()
            sp[:, cont] += np.random.default_rng(100 * seed + j).uniform(-eps, eps, size=(st.shape[0], cont.size))
            for q in quats:
                sp[:, q] /= np.linalg.norm(sp[:, q], axis=1, keepdims=True)
            sens = np.maximum(sens, np.abs(self.step_from(sp, a)[1][:, cont] - nxt[:, cont]).max(axis=1))
        return sens


@pytest.fixture(scope="session")
def sharded_handover():
    return ShardedOracleHandover


@pytest.fixture(scope="session")
def oracle_torch_env():
    return OracleTorchEnv


@pytest.fixture(scope="session")
def hostcore():
    return HostCore()
