"""AddressSanitizer / UndefinedBehaviorSanitizer run of the kernel core and of the oracle on the CPU
(GPU ASan is not available on the pool): both are compiled with -fsanitize=address,undefined into
stand-alone executables that replay a reset + a few steps of PickAndPlace and Reach."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MAIN = r'''
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
extern "C" {
void xh_init(int, uint64_t, int64_t, double, double, int, int, int64_t, double*);
void xh_reset(int, uint64_t, int64_t, double, double, int, int, int64_t, double*, const uint8_t*, double*, double*, double*);
void xh_step(int, uint64_t, int64_t, double, double, int, int, int64_t, double*, const double*, double*, double*, double*, double*, uint8_t*, uint8_t*);
void xh_reach_init(int, uint64_t, int64_t, int, int64_t, double*);
void xh_reach_reset(int, uint64_t, int64_t, int, int64_t, double*, const uint8_t*, double*, double*, double*);
void xh_reach_step(int, uint64_t, int64_t, int, int64_t, double*, const double*, double*, double*, double*, double*, uint8_t*, uint8_t*, int32_t*);
void xh_ho_init(int, uint64_t, int64_t, double, int, int64_t, double*);
void xh_ho_reset(int, uint64_t, int64_t, double, int, int64_t, double*, const uint8_t*, double*, double*, double*);
void xh_ho_step(int, uint64_t, int64_t, double, int, int64_t, double*, const double*, double*, double*, double*, double*, uint8_t*, uint8_t*);
void xh_st_init(int, uint64_t, int64_t, int, int64_t, double*);
void xh_st_reset(int, uint64_t, int64_t, int, int64_t, double*, const uint8_t*, double*, double*, double*);
void xh_st_step(int, uint64_t, int64_t, int, int64_t, double*, const double*, double*, double*, double*, double*, uint8_t*, uint8_t*);
void xh_coop_step(int, uint64_t, int64_t, double, double, int, int, int64_t, double*, const double*, double*, double*, double*, double*, uint8_t*, uint8_t*);
void xh_coop_reset(int, uint64_t, int64_t, double, double, int, int, int64_t, double*, const uint8_t*, double*, double*, double*);
void xh_ho_set_use_stand(int);
}
int main() {
    const int64_t E = 3;
    double *st = (double*)calloc(E * 54, 8), *obs = (double*)calloc(E * 24, 8), *ag = (double*)calloc(E * 3, 8), *dg = (double*)calloc(E * 3, 8);
    double *rew = (double*)calloc(E, 8), *act = (double*)calloc(E * 4, 8);
    uint8_t *done = (uint8_t*)calloc(E, 1), *succ = (uint8_t*)calloc(E, 1);
    int32_t *fut = (int32_t*)calloc(E, 4);
    for (int f32 = 0; f32 < 2; f32++) {
        xh_init(f32, 3, 0, 0.3, 0.3, 0, 2, E, st);
        xh_reset(f32, 3, 0, 0.3, 0.3, 0, 2, E, st, 0, obs, ag, dg);
        for (int t = 0; t < 2; t++) {
            for (int k = 0; k < E * 4; k++) act[k] = ((t * 7 + k * 13) % 21) / 10.0 - 1.0;
            xh_step(f32, 3, 0, 0.3, 0.3, 0, 2, E, st, act, obs, ag, dg, rew, done, succ);
        }
        // the cooperative (16 lanes per env) core: one step and one reset of one env (all four row-set instantiations
        // are compiled; init_grasp_rate 0.3 puts objects between the fingers in some envs)
        xh_coop_step(f32, 3, 0, 0.3, 0.3, 0, 2, 1, st, act, obs, ag, dg, rew, done, succ);
        xh_coop_reset(f32, 3, 0, 0.3, 0.3, 0, 2, 1, st, 0, obs, ag, dg);
        double *rs = (double*)calloc(E * 45, 8), *ro = (double*)calloc(E * 8, 8);
        xh_reach_init(f32, 1, 0, 2, E, rs);
        xh_reach_reset(f32, 1, 0, 2, E, rs, 0, ro, ag, dg);
        for (int t = 0; t < 2; t++) xh_reach_step(f32, 1, 0, 2, E, rs, act, ro, ag, dg, rew, done, succ, fut);
        free(rs); free(ro);
        // StackTower: two lanes (threads) per env; env 0 gets two overlapping cubes so that the cube/cube manifold
        // (clipped polygon in the lane's LDS columns) and its solver rows run under the sanitizers too
        const int64_t ES = 2;
        double *ss = (double*)calloc(ES * 136, 8), *so = (double*)calloc(ES * 55, 8), *sg = (double*)calloc(ES * 9, 8), *sd = (double*)calloc(ES * 9, 8);
        double *sa = (double*)calloc(ES * 8, 8);
        xh_st_init(f32, 5, 0, 0, ES, ss);
        xh_st_reset(f32, 5, 0, 0, ES, ss, 0, so, sg, sd);
        ss[57] = ss[54] + 0.03; ss[58] = ss[55] + 0.004; ss[59] = ss[56];
        for (int t = 0; t < 2; t++) {
            for (int k = 0; k < ES * 8; k++) sa[k] = ((t * 5 + k * 11) % 21) / 10.0 - 1.0;
            xh_st_step(f32, 5, 0, 0, ES, ss, sa, so, sg, sd, rew, done, succ);
        }
        // Handover: one env, reset (6 ticks) + one step (15 ticks)
        double *hs = (double*)calloc(76, 8), *ho = (double*)calloc(29, 8);
        xh_ho_init(f32, 9, 0, 0.5, 1, 1, hs);
        xh_ho_reset(f32, 9, 0, 0.5, 1, 1, hs, 0, ho, ag, dg);
        xh_ho_step(f32, 9, 0, 0.5, 1, 1, hs, sa, ho, ag, dg, rew, done, succ);
        xh_ho_set_use_stand(1);              // the stand's support points (HandoverStandScene)
        hs[38] = hs[51]; hs[39] = hs[52]; hs[40] = hs[53] + 0.002;
        xh_ho_step(f32, 9, 0, 0.5, 1, 1, hs, sa, ho, ag, dg, rew, done, succ);
        xh_ho_set_use_stand(0);
        free(hs); free(ho);
        free(ss); free(so); free(sg); free(sd); free(sa);
    }
    printf("ok %f\n", st[0] + obs[0]);
    free(st); free(obs); free(ag); free(dg); free(rew); free(act); free(done); free(succ); free(fut);
    return 0;
}
'''

ORACLE_MAIN = r'''
#include <stdio.h>
#include <string.h>
#include "xarm_oracle.h"
int main(void) { unsigned int o[4]; xo_philox(1, 2, 3, 4, 5, o); printf("ok %u %d\n", o[0], xo_state_dim()); return 0; }
'''


def _have_sanitizers():
    return subprocess.run("echo 'int main(){return 0;}' | g++ -x c++ - -fsanitize=address,undefined -o /tmp/_san_probe && /tmp/_san_probe",
                          shell=True, capture_output=True).returncode == 0


@pytest.mark.skipif(not _have_sanitizers(), reason="libasan/libubsan not available")
def test_kernel_core_under_asan_ubsan(tmp_path):
    main = tmp_path / "main.cpp"
    main.write_text(MAIN)
    exe = tmp_path / "san_core"
    # -g1: line tables only (full debug info of the fully inlined templates triples the compile time)
    # -O0: the fully unrolled templates take 2 min to compile at -O1 under the sanitizers, 25 s at -O0
    subprocess.check_call(["g++", "-O0", "-g1", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-Wno-unknown-pragmas", "-o", str(exe), str(main), os.path.join(ROOT, "tests", "hostbuild", "xarm_host.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


@pytest.mark.skipif(not _have_sanitizers(), reason="libasan/libubsan not available")
def test_oracle_under_asan_ubsan(tmp_path):
    """the oracle is driven through a sanitized shared object from a sanitized python-free harness"""
    main = tmp_path / "omain.c"
    main.write_text(ORACLE_MAIN)
    exe = tmp_path / "san_oracle"
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c99", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "oracle"), "-o", str(exe), str(main),
                           os.path.join(ROOT, "oracle", "xarm_oracle.c"), "-lm"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
    # and a full oracle episode under ASan via LD_PRELOAD-free ctypes is covered by building the .so with UBSan only
    so = tmp_path / "libxo_ubsan.so"
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c99", "-fPIC", "-shared", "-fsanitize=undefined", "-fno-sanitize-recover=all",
                           "-o", str(so), os.path.join(ROOT, "oracle", "xarm_oracle.c"), "-lm"])
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import oracle as O\n"
            "O.LIB_PATH = %r; O.build_lib = lambda force=False: None\n"
            "import numpy as np\n"
            "e = O.OraclePnP(3, seed=2, init_grasp_rate=0.5); e.reset()\n"
            "[e.step(np.random.default_rng(k).uniform(-1.5, 1.5, (3, 4))) for k in range(5)]\n"
            "r = O.OracleReach(3, seed=2, reward_type='dense_diff'); r.reset(); r.step(np.ones((3, 4)))\n"
            "h = O.OracleHandover(2, seed=2); h.reset(); h.step(np.full((2, 8), 0.5))\n"
            "t = O.OracleStackTower(2, seed=2); t.reset(); s = t.get_state(); s[0, 57:60] = s[0, 54:57] + [0.03, 0.004, 0]\n"
            "s[1, 57:60] = s[1, 54:57] + [0.001, 0.002, 0.05]; s[1, 67:71] = [0, 0, 0.38, 0.925]; t.set_state(s)\n"
            "[t.step(np.random.default_rng(k).uniform(-1, 1, (2, 8))) for k in range(3)]\n"
            "print('ok')\n") % (ROOT, str(so))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
