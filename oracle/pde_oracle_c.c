/*
 * C restatement of the K1 hot path (implicit ADI diffusion layer), forward AND a closed-form
 * backward.  TEST INFRASTRUCTURE — NOT A PRODUCT PATH: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this; the product never does.
 *
 * Why it exists next to oracle/pde_oracle.py (which is pinned bitwise against the reference's own
 * vectors and gets its gradients from autograd): this file derives the gradients by hand
 *  with the PLAIN one-sided Thomas factorisation, so it is an independent
 * check of the adjoint mathematics the HIP kernels implement, and it is fast enough to check
 * full-size cases on the GPU box's host.  tests/test_oracle_c.py pins it against pde_oracle.py
 * (hence, transitively, against tests/golden).
 *
 * Follows, in arithmetic order:
 *   get_alpha_beta_at_time   mnist_test.py:33-42   cifar10.py:53-63
 *   smooth_coefficients      mnist_test.py:135-149
 *   diffuse_x / diffuse_y    mnist_test.py:67-133
 *   thomas_solver_batch      mnist_test.py:151-198
 * Compiled twice: REAL=double (oracle_*_f64) and REAL=float (oracle_*_f32).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL double
#define SUF(x) x##_f64
#endif

#define MAXN 64

typedef struct { int axis; double delta, h2, t; } OSweep;

/* coefficient of one line of one sweep: kap[i], pass[i] (clamp lets the gradient through) */
static void line_coeff(const REAL* base, const REAL* slope, int stride, int N, const OSweep* sw, int smooth3,
                       int has_max, REAL cmax, REAL eps, REAL* kap, int* pass) {
    REAL th[MAXN];
    const REAL t = (REAL)sw->t;
    for (int i = 0; i < N; ++i) {
        REAL v = base[i * stride] + slope[i * stride] * t;
        pass[i] = (v >= eps) && (!has_max || v <= cmax);
        if (v < eps) v = eps;
        if (has_max && v > cmax) v = cmax;
        th[i] = v;
    }
    const REAL third = (REAL)1 / (REAL)3;
    for (int i = 0; i < N; ++i) {
        REAL v = th[i];
        if (smooth3) v = (th[i > 0 ? i - 1 : 0] * third + th[i] * third) + th[i + 1 < N ? i + 1 : N - 1] * third;
        kap[i] = (v * (REAL)sw->delta) / (REAL)sw->h2;
    }
}

/* (A + eps I) x = d, reference recurrences; keeps c* and 1/den for the adjoint */
static void thomas(const REAL* kap, int N, REAL eps, const REAL* d, int ds, REAL* x, int xs, REAL* cs, REAL* den) {
    REAL dst[MAXN];
    den[0] = ((REAL)1 + kap[0]) + eps;
    cs[0] = -kap[0] / den[0];
    dst[0] = d[0] / den[0];
    for (int i = 1; i < N; ++i) {
        const REAL b = (i == N - 1) ? (REAL)1 + kap[i] : (REAL)1 + (REAL)2 * kap[i];
        den[i] = b - (-kap[i]) * cs[i - 1] + eps;
        cs[i] = -kap[i] / den[i];
        dst[i] = (d[i * ds] - (-kap[i]) * dst[i - 1]) / den[i];
    }
    x[(N - 1) * xs] = dst[N - 1];
    for (int i = N - 2; i >= 0; --i) x[i * xs] = dst[i] - cs[i] * x[(i + 1) * xs];
}

/* (A + eps I)^T g = r with the factors of thomas(): U^T z = r, L^T g = z */
static void thomas_adj(const REAL* kap, int N, const REAL* cs, const REAL* den, const REAL* r, int rs, REAL* g, int gs) {
    REAL z[MAXN];
    z[0] = r[0];
    for (int i = 1; i < N; ++i) z[i] = r[i * rs] - cs[i - 1] * z[i - 1];
    g[(N - 1) * gs] = z[N - 1] / den[N - 1];
    for (int i = N - 2; i >= 0; --i) g[i * gs] = (z[i] + kap[i + 1] * g[(i + 1) * gs]) / den[i];
}

static void sweep_plane(const REAL* ab, const REAL* bb, const REAL* as, const REAL* bs, int N, const OSweep* sw,
                        int smooth3, int has_max, REAL cmax, REAL eps, const REAL* in, REAL* out) {
    REAL kap[MAXN], cs[MAXN], den[MAXN];
    int pass[MAXN];
    for (int l = 0; l < N; ++l) {
        if (sw->axis == 0) {
            line_coeff(ab + l * N, as + l * N, 1, N, sw, smooth3, has_max, cmax, eps, kap, pass);
            thomas(kap, N, eps, in + l * N, 1, out + l * N, 1, cs, den);
        } else {
            line_coeff(bb + l, bs + l, N, N, sw, smooth3, has_max, cmax, eps, kap, pass);
            thomas(kap, N, eps, in + l, N, out + l, N, cs, den);
        }
    }
}

/* u, y: (B,C,N,N); parameters (C,N,N).  states (optional): (B,C,S,N,N) outputs of every sweep. */
int SUF(oracle_adi_forward)(int B, int C, int N, int S, const OSweep* sw, int smooth3, int has_max, double cmax,
                            double eps, const REAL* ab, const REAL* bb, const REAL* as, const REAL* bs,
                            const REAL* u, REAL* y, REAL* states) {
    if (N > MAXN || N < 2) return -1;
    const size_t P = (size_t)N * N;
#pragma omp parallel for schedule(static)
    for (long pc = 0; pc < (long)B * C; ++pc) {
        const int c = (int)(pc % C);
        REAL cur[MAXN * MAXN], nxt[MAXN * MAXN];
        memcpy(cur, u + pc * P, P * sizeof(REAL));
        for (int s = 0; s < S; ++s) {
            sweep_plane(ab + c * P, bb + c * P, as + c * P, bs + c * P, N, &sw[s], smooth3, has_max, (REAL)cmax,
                        (REAL)eps, cur, nxt);
            memcpy(cur, nxt, P * sizeof(REAL));
            if (states) memcpy(states + ((size_t)pc * S + s) * P, cur, P * sizeof(REAL));
        }
        memcpy(y + pc * P, cur, P * sizeof(REAL));
    }
    return 0;
}

/* Exact reverse mode.  For a sweep with output x and upstream r:  g = (A+eps I)^-T r is the
 * gradient of the sweep's input;  dL/dkap_i = -g_i (L x)_i with L the Neumann second difference;
 * dL/dtheta~ = dL/dkap * delta/h2; the 3-tap average is undone with its transpose; the clamp lets
 * it through where eps <= base+slope*t <= max;  base += that, slope += t * that. */
int SUF(oracle_adi_backward)(int B, int C, int N, int S, const OSweep* sw, int smooth3, int has_max, double cmax,
                             double eps, const REAL* ab, const REAL* bb, const REAL* as, const REAL* bs,
                             const REAL* u, const REAL* gy, REAL* gu, REAL* g_ab, REAL* g_bb, REAL* g_as,
                             REAL* g_bs) {
    if (N > MAXN || N < 2) return -1;
    const size_t P = (size_t)N * N;
    const size_t PC = P * C;
    memset(g_ab, 0, PC * sizeof(REAL)); memset(g_bb, 0, PC * sizeof(REAL));
    memset(g_as, 0, PC * sizeof(REAL)); memset(g_bs, 0, PC * sizeof(REAL));
    int fail = 0;
#pragma omp parallel
    {
        REAL* st = (REAL*)malloc((size_t)(S + 1) * P * sizeof(REAL));
        double* acc = (double*)calloc(4 * PC, sizeof(double));     /* thread-local, reduced below */
        if (!st || !acc) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(static)
            for (long pc = 0; pc < (long)B * C; ++pc) {
                const int c = (int)(pc % C);
                const REAL *pab = ab + c * P, *pbb = bb + c * P, *pas = as + c * P, *pbs = bs + c * P;
                memcpy(st, u + pc * P, P * sizeof(REAL));
                for (int s = 0; s < S; ++s)
                    sweep_plane(pab, pbb, pas, pbs, N, &sw[s], smooth3, has_max, (REAL)cmax, (REAL)eps, st + s * P,
                                st + (s + 1) * P);
                REAL r[MAXN * MAXN], g[MAXN * MAXN];
                memcpy(r, gy + pc * P, P * sizeof(REAL));
                for (int s = S - 1; s >= 0; --s) {
                    const REAL* x = st + (size_t)(s + 1) * P;
                    const int ax = sw[s].axis;
                    const REAL w = (REAL)sw[s].delta / (REAL)sw[s].h2, t = (REAL)sw[s].t;
                    double* gb = acc + (ax == 0 ? 0 : 1) * PC + (size_t)c * P;
                    double* gs = acc + (ax == 0 ? 2 : 3) * PC + (size_t)c * P;
                    for (int l = 0; l < N; ++l) {
                        REAL kap[MAXN], cs[MAXN], den[MAXN], tmp[MAXN], dk[MAXN];
                        int pass[MAXN];
                        const int o0 = ax == 0 ? l * N : l, str = ax == 0 ? 1 : N;
                        line_coeff((ax == 0 ? pab : pbb) + o0, (ax == 0 ? pas : pbs) + o0, str, N, &sw[s], smooth3,
                                   has_max, (REAL)cmax, (REAL)eps, kap, pass);
                        thomas(kap, N, (REAL)eps, st + (size_t)s * P + o0, str, tmp, 1, cs, den);   /* refactor */
                        thomas_adj(kap, N, cs, den, r + o0, str, g + o0, str);
                        for (int i = 0; i < N; ++i) {
                            const REAL xi = x[o0 + i * str];
                            REAL lx = (i == 0 || i == N - 1) ? xi : (REAL)2 * xi;
                            if (i > 0) lx -= x[o0 + (i - 1) * str];
                            if (i < N - 1) lx -= x[o0 + (i + 1) * str];
                            dk[i] = -g[o0 + i * str] * lx * w;                /* dL/dtheta~_i */
                        }
                        for (int i = 0; i < N; ++i) {
                            REAL dth = dk[i];
                            if (smooth3) {
                                dth = dk[i] * ((i == 0 || i == N - 1) ? (REAL)2 : (REAL)1);
                                if (i > 0) dth += dk[i - 1];
                                if (i < N - 1) dth += dk[i + 1];
                                dth *= (REAL)1 / (REAL)3;
                            }
                            if (pass[i]) {
                                gb[o0 + i * str] += (double)dth;
                                gs[o0 + i * str] += (double)(dth * t);
                            }
                        }
                    }
                    memcpy(r, g, P * sizeof(REAL));
                }
                memcpy(gu + pc * P, r, P * sizeof(REAL));
            }
#pragma omp critical
            for (size_t e = 0; e < PC; ++e) {
                g_ab[e] += (REAL)acc[e];
                g_bb[e] += (REAL)acc[PC + e];
                g_as[e] += (REAL)acc[2 * PC + e];
                g_bs[e] += (REAL)acc[3 * PC + e];
            }
        }
        free(st); free(acc);
    }
    return fail ? -2 : 0;
}
