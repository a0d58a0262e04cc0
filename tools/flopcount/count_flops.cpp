// count_flops.cpp - floating-point operation count of one env step of the kernel core.
//
// gym_xarm_amd/csrc/xarm_core.h is a template on its scalar type.  Here it is instantiated with a scalar that counts
// every add / subtract / multiply / divide and every sqrt / sin / cos / atan2 / tanh it performs (the operations a
// fused multiply-add performs count as two), and stepped over a batch of environments - SURVEY.md 8(d)'s "counted
// FLOPs from an instrumented scalar type" for the roofline's fp32-vector ceiling.  The numbers go to
// profiles/flop_count.json through tools/count_flops.py.  Development tool: never part of the product.
#define XARM_HOST_BUILD 1
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

namespace fc {
static long long n_arith = 0, n_div = 0, n_special = 0;
struct Cnt {
    float v;
    Cnt() : v(0) {}
    Cnt(double x) : v((float)x) {}
    Cnt(float x) : v(x) {}
    Cnt(int x) : v((float)x) {}
    Cnt(unsigned x) : v((float)x) {}
    Cnt(long x) : v((float)x) {}
    Cnt(long long x) : v((float)x) {}
    explicit operator int() const { return (int)v; }
    explicit operator long() const { return (long)v; }
    explicit operator long long() const { return (long long)v; }
    explicit operator double() const { return v; }
    explicit operator float() const { return v; }
};
inline Cnt operator+(Cnt a, Cnt b) { n_arith++; return Cnt(a.v + b.v); }
inline Cnt operator-(Cnt a, Cnt b) { n_arith++; return Cnt(a.v - b.v); }
inline Cnt operator*(Cnt a, Cnt b) { n_arith++; return Cnt(a.v * b.v); }
inline Cnt operator/(Cnt a, Cnt b) { n_div++; return Cnt(a.v / b.v); }
inline Cnt operator-(Cnt a) { return Cnt(-a.v); }
inline Cnt &operator+=(Cnt &a, Cnt b) { a = a + b; return a; }
inline Cnt &operator-=(Cnt &a, Cnt b) { a = a - b; return a; }
inline Cnt &operator*=(Cnt &a, Cnt b) { a = a * b; return a; }
inline bool operator<(Cnt a, Cnt b) { return a.v < b.v; }
inline bool operator>(Cnt a, Cnt b) { return a.v > b.v; }
inline bool operator<=(Cnt a, Cnt b) { return a.v <= b.v; }
inline bool operator>=(Cnt a, Cnt b) { return a.v >= b.v; }
inline bool operator==(Cnt a, Cnt b) { return a.v == b.v; }
inline bool operator!=(Cnt a, Cnt b) { return a.v != b.v; }
inline Cnt xsqrt(Cnt x) { n_special++; return Cnt(sqrtf(x.v)); }
inline void xsincos(Cnt x, Cnt &s, Cnt &c) { n_special += 2; s = Cnt(sinf(x.v)); c = Cnt(cosf(x.v)); }
inline Cnt xsin(Cnt x) { n_special++; return Cnt(sinf(x.v)); }
inline Cnt xcos(Cnt x) { n_special++; return Cnt(cosf(x.v)); }
inline Cnt xatan2(Cnt y, Cnt x) { n_special++; return Cnt(atan2f(y.v, x.v)); }
inline Cnt xasin(Cnt x) { n_special++; return Cnt(asinf(x.v)); }
inline Cnt xtanh(Cnt x) { n_special++; return Cnt(tanhf(x.v)); }
inline Cnt xabs(Cnt x) { return Cnt(fabsf(x.v)); }
inline Cnt xremainder(Cnt x, Cnt y) { n_special++; return Cnt(remainderf(x.v, y.v)); }
inline Cnt xpow(Cnt x, Cnt y) { n_special++; return Cnt(powf(x.v, y.v)); }
}
// the core looks its math helpers up unqualified: make the counting overloads visible inside namespace xk
namespace xk { using fc::xsqrt; using fc::xsincos; using fc::xsin; using fc::xcos; using fc::xatan2; using fc::xasin; using fc::xtanh; using fc::xabs; using fc::xremainder; using fc::xpow; }
#include "../../gym_xarm_amd/csrc/xarm_core.h"

template <typename T> struct HostLds { T *base; T &operator[](int i) const { return base[i]; } };

int main(int argc, char **argv) {
    using fc::Cnt;
    const int E = argc > 1 ? atoi(argv[1]) : 256, STEPS = argc > 2 ? atoi(argv[2]) : 20;
    xk::EnvCfg cfg;
    cfg.seed = 0; cfg.env_id_offset = 0; cfg.init_grasp_rate = 0; cfg.goal_ground_rate = 0; cfg.goal_shape = 0; cfg.reward_type = 0;
    long long reset_ops[3] = {0, 0, 0}, step_ops[3] = {0, 0, 0}, step_min = 1ll << 60, step_max = 0;
    long long n_steps = 0, n_contact_steps = 0;
    uint32_t rng = 12345u;
    for (int e = 0; e < E; e++) {
        xk::EnvState<Cnt> s;
        Cnt lds[xk::LDS_FLOATS];
        HostLds<Cnt> L{lds};
        xk::env_init<Cnt>(cfg, e, s);
        fc::n_arith = fc::n_div = fc::n_special = 0;
        xk::env_reset<Cnt>(cfg, e, s, L);
        reset_ops[0] += fc::n_arith; reset_ops[1] += fc::n_div; reset_ops[2] += fc::n_special;
        for (int k = 0; k < STEPS; k++) {
            Cnt a[4], obs[xk::OBS_DIM], r;
            for (int j = 0; j < 4; j++) { rng = rng * 1664525u + 1013904223u; a[j] = Cnt((double)(rng >> 8) / 8388608.0 - 1.0); }
            bool d, su;
            fc::n_arith = fc::n_div = fc::n_special = 0;
            xk::env_step<Cnt>(cfg, s, a, obs, r, d, su, L);
            const long long tot = fc::n_arith + fc::n_div + fc::n_special;
            step_ops[0] += fc::n_arith; step_ops[1] += fc::n_div; step_ops[2] += fc::n_special;
            step_min = tot < step_min ? tot : step_min; step_max = tot > step_max ? tot : step_max;
            n_steps++;
            bool touching = false;
            for (int j = 0; j < 8; j++) touching = touching || s.lam_p[j].v != 0.f;
            n_contact_steps += touching;
            if (d) xk::env_reset<Cnt>(cfg, e, s, L);
        }
    }
    printf("{\"envs\": %d, \"steps_per_env\": %d, \"flops_per_env_step\": %.1f, \"add_sub_mul_per_env_step\": %.1f, \"div_per_env_step\": %.1f, "
           "\"sqrt_sincos_etc_per_env_step\": %.1f, \"min_env_step\": %lld, \"max_env_step\": %lld, \"share_of_steps_with_pad_contact\": %.4f, "
           "\"flops_per_env_reset\": %.1f}\n",
           E, STEPS, (double)(step_ops[0] + step_ops[1] + step_ops[2]) / n_steps, (double)step_ops[0] / n_steps, (double)step_ops[1] / n_steps,
           (double)step_ops[2] / n_steps, step_min, step_max, (double)n_contact_steps / n_steps,
           (double)(reset_ops[0] + reset_ops[1] + reset_ops[2]) / E);
    return 0;
}
