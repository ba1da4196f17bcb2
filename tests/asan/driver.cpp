// Host-side walk through the C ABI of libpnmol_hip under AddressSanitizer + UBSan, against tests/asan/hip_mock.c (no GPU:
// kernels do not run, results are zeros).  Every call's return code is printed; the run passes when the sanitizers stay
// silent and the process exits 0.  Exercises: descriptor validation, filter/state creation for all sizes of n, layout
// conversions of set/get (the index arithmetic), ELL rebuilds (set_operator, dense and wide rows), error model, step and
// steps bookkeeping (graph capture path included), the lifetime rule in every wrong order, the square-root entry points.
#include <cmath>
#include <cstdio>
#include <vector>

#include "pnmol_hip.h"
#include "pnmol_sqrt.h"

static int fails = 0;
#define EXPECT(call, want)                                                              \
    do {                                                                                \
        const int rc__ = (call);                                                        \
        std::printf("%-70s -> %d%s\n", #call, rc__, rc__ == (want) ? "" : "   UNEXPECTED"); \
        if (rc__ != (want)) ++fails;                                                    \
    } while (0)
#define ANY(call)                                        \
    do {                                                 \
        const int rc__ = (call);                         \
        std::printf("%-70s -> %d\n", #call, rc__);       \
    } while (0)

int main() {
    pnmol_ctx* ctx = nullptr;
    EXPECT(pnmol_ctx_create(0, &ctx), 0);
    EXPECT(pnmol_ctx_create(5, nullptr), -1);
    for (int nu = 1; nu <= 3; ++nu)
        for (int d : {5, 33, 70}) {
            const int nB = 2, n = nu + 1, D = n * d;
            std::vector<double> L(d * d, 0.0), B(nB * d, 0.0), E(d * d, 0.0), R(nB * nB, 0.0), Gm(d * d, 0.0);
            for (int i = 0; i < d; ++i) {
                L[i * d + i] = -2.0;
                if (i) L[i * d + i - 1] = 1.0;
                if (i + 1 < d) L[i * d + i + 1] = 1.0;
                E[i * d + i] = 1e-3;
                for (int k = 0; k <= i; ++k) Gm[i * d + k] = (i == k) ? 1.0 : 0.1 / (1 + i - k);
            }
            B[0] = 1.0, B[nB * d - 1] = 1.0;
            pnmol_filter_desc desc{};
            desc.d = d, desc.num_derivatives = nu, desc.nB = nB, desc.L = L.data(), desc.B = B.data();
            desc.E_sqrtm = E.data(), desc.R_sqrtm = R.data(), desc.Gamma = Gm.data();
            pnmol_filter* f = nullptr;
            desc.dtype = 3;
            EXPECT(pnmol_filter_create(ctx, &desc, &f), -1);
            desc.dtype = 0;
            desc.d = 0;
            EXPECT(pnmol_filter_create(ctx, &desc, &f), -1);
            desc.d = d;
            EXPECT(pnmol_filter_create(ctx, &desc, &f), 0);
            int dd, nn, mm, dp, mp, kern, home;
            EXPECT(pnmol_filter_dims(f, &dd, &nn, &mm, &dp, &mp), 0);
            EXPECT(pnmol_filter_sweep_layout(f, &kern, &home), 0);
            pnmol_state *s0 = nullptr, *s1 = nullptr, *s2 = nullptr;
            EXPECT(pnmol_state_create(f, &s0), 0);
            EXPECT(pnmol_state_create(f, &s1), 0);
            std::vector<double> mean(n * d), cov((size_t)D * D, 0.0), out((size_t)D * D), v(n * d);
            for (int i = 0; i < n * d; ++i) mean[i] = std::sin(0.1 * i);
            for (int i = 0; i < D; ++i) cov[(size_t)i * D + i] = 1.0 + i;
            EXPECT(pnmol_state_set(s0, 0.25, mean.data(), cov.data()), 0);
            EXPECT(pnmol_state_set_sqrtm(s0, 0.25, mean.data(), cov.data()), 0);
            EXPECT(pnmol_state_clone(s0, &s2), 0);
            EXPECT(pnmol_state_get_mean(s0, v.data()), 0);
            EXPECT(pnmol_state_get_cov(s0, out.data()), 0);
            EXPECT(pnmol_state_get_marginal_var(s0, v.data()), 0);
            ANY(pnmol_state_get_cov_sqrtm(s0, out.data()));     // (mock: the factorisation kernel does not run)
            pnmol_step_out so{};
            std::vector<double> err(d);
            EXPECT(pnmol_filter_step(f, s0, 0.1, s0, &so, err.data()), -1);   // aliasing
            EXPECT(pnmol_filter_step(f, s0, -1.0, s1, &so, err.data()), -1);  // dt <= 0
            ANY(pnmol_filter_step(f, s0, 0.1, s1, &so, err.data()));          // (mock: info word 0 -> "not positive definite")
            ANY(pnmol_filter_prepare_error_model(f, 0.1));
            std::vector<double> sq((size_t)(d + nB) * (d + nB), 0.0), sqd(d + nB, 1.0);
            EXPECT(pnmol_filter_set_error_model(f, 0.1, sq.data(), sqd.data()), 0);
            std::vector<double> m_at(d), Mdense((size_t)d * d, 0.5), shift(d, 0.1);
            EXPECT(pnmol_filter_predict_mean(f, s0, 0.1, m_at.data()), 0);
            EXPECT(pnmol_filter_set_operator(f, Mdense.data(), shift.data()), 0);   // dense Jacobian: wide ELL, reallocation
            EXPECT(pnmol_filter_set_operator(f, L.data(), nullptr), 0);             // back to the stencil
            for (int k : {1, 2, 3, 13, 24}) {
                std::vector<double> means((size_t)k * d), stds((size_t)k * d);
                std::vector<pnmol_step_out> infos(k);
                ANY(pnmol_filter_prepare_steps(f, s0, k, 0.05));
                ANY(pnmol_filter_steps(f, s0, k, 0.05, means.data(), stds.data(), infos.data()));
            }
            EXPECT(pnmol_filter_steps(f, s0, 0, 0.05, nullptr, nullptr, nullptr), -1);
            EXPECT(pnmol_filter_steps_begin(f, s0, 4, 0.05), 0);
            EXPECT(pnmol_state_destroy(s0), -1);                 // target of an unfinished steps_begin
            ANY(pnmol_filter_steps_end(f, s0, nullptr, nullptr, nullptr));
            EXPECT(pnmol_filter_steps_end(f, s0, nullptr, nullptr, nullptr), -1);   // nothing pending
            // lifetime rule, every wrong order
            EXPECT(pnmol_ctx_destroy(ctx), -1);
            EXPECT(pnmol_filter_destroy(f), -1);
            EXPECT(pnmol_state_destroy(s0), 0);
            EXPECT(pnmol_state_destroy(s1), 0);
            EXPECT(pnmol_filter_destroy(f), -1);
            EXPECT(pnmol_state_destroy(s2), 0);
            EXPECT(pnmol_filter_destroy(f), 0);
            // the square-root side
            pnmol_sqrt_filter* q = nullptr;
            desc.dtype = 1;
            EXPECT(pnmol_sqrt_filter_create(ctx, &desc, &q), -1);
            desc.dtype = 0;
            EXPECT(pnmol_sqrt_filter_create(ctx, &desc, &q), 0);
            EXPECT(pnmol_ctx_destroy(ctx), -1);
            ANY(pnmol_sqrt_filter_set_state(q, 0.0, mean.data(), cov.data()));
            pnmol_step_out qo{};
            ANY(pnmol_sqrt_filter_step(q, 0.1, &qo, err.data()));
            { double tq = 0.0; ANY(pnmol_sqrt_filter_get_state(q, &tq, mean.data(), out.data())); }
            EXPECT(pnmol_sqrt_filter_destroy(q), 0);
            std::vector<double> A((size_t)(2 * D) * D, 0.3), Rr((size_t)D * D);
            ANY(pnmol_qr_r(ctx, A.data(), 2 * D, D, Rr.data()));
            ANY(pnmol_sqrt_propagate_cholesky_factor(ctx, cov.data(), D, D, cov.data(), D, Rr.data()));
        }
    EXPECT(pnmol_ctx_destroy(ctx), 0);
    EXPECT(pnmol_filter_destroy(nullptr), -1);
    EXPECT(pnmol_state_destroy(nullptr), -1);
    std::printf("%s (%d unexpected return codes)\n", fails ? "FAILED" : "ok", fails);
    return fails ? 1 : 0;
}
