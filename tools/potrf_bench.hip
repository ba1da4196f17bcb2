// Standalone timing harness for the 32x32 diagonal-block factorisations (not part of the product):
// the two-wave column-at-a-time scheme (diag2w_from_lds) and the four-wave 4-column-blocked scheme (diag4_factor).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/potrf_bench.hip -o tools/potrf_bench
#include "../pnmol-experiments_amd/csrc/pnmol_hip.hip"

#include <random>

__global__ __launch_bounds__(128) void k_time_diag2w(const double* __restrict__ G, double* __restrict__ F,
                                                     double* __restrict__ Linv, int* info, double* sdiag,
                                                     unsigned long long* out) {
    __shared__ double sT[NB * TLD];
    __shared__ __attribute__((aligned(16))) Diag2wLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < NB * NB; e += 128) sT[(e >> 5) * TLD + (e & 31)] = G[(e >> 5) * NB + (e & 31)];
    if (tid < NB) L.flag[tid] = 0;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    diag2w_from_lds(sT, F, NB, Linv, wave, lane, info, 0, sdiag, sdiag[NB], &L);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        out[2 * wave] = t1 - t0;
        out[2 * wave + 1] = r1 - r0;
    }
}

__global__ __launch_bounds__(256) void k_time_diag4(const double* __restrict__ G, double* __restrict__ F,
                                                    double* __restrict__ Linv, int* info, unsigned long long* out) {
    __shared__ double sT[NB * TLD];
    __shared__ double sd[NB];
    __shared__ __attribute__((aligned(16))) Diag4Lds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < NB * NB; e += 256) sT[(e >> 5) * TLD + (e & 31)] = G[(e >> 5) * NB + (e & 31)];
    if (tid < NB) sd[tid] = G[tid * NB + tid];
    if (tid < 16) L.flagA[tid] = 0;
    __syncthreads();
    const d4 nq = diag4_quadrant_from_lds(sT, wave, lane);
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    diag4_factor<true>(nq, &L, wave, lane, F, NB, Linv, info, 0, sd, 1e300);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();   // role finished, stores issued
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();   // ... and drained
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        out[3 * wave] = t1 - t0;
        out[3 * wave + 1] = t2 - t0;
        out[3 * wave + 2] = r1 - r0;
    }
}

static void check(const char* name, const std::vector<double>& S, double* dF, double* dL) {
    std::vector<double> L(NB * NB), X(NB * NB);
    hipMemcpy(L.data(), dF, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
    hipMemcpy(X.data(), dL, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, up = 0;
    for (int i = 0; i < NB; ++i)
        for (int j = 0; j < NB; ++j) {
            double s = 0, t = 0;
            for (int k = 0; k < NB; ++k) s += L[i * NB + k] * L[j * NB + k], t += X[i * NB + k] * L[k * NB + j];
            e1 = std::max(e1, std::fabs(s - S[i * NB + j]));
            e2 = std::max(e2, std::fabs(t - (i == j)));
            if (j > i) up = std::max(up, std::max(std::fabs(L[i * NB + j]), std::fabs(X[i * NB + j])));
        }
    std::printf("%s: max |LL^T - S| = %.3e   max |Linv L - I| = %.3e   max upper = %.1e\n", name, e1, e2, up);
}

int main() {
    std::mt19937 rng(1);
    std::normal_distribution<double> nd;
    std::vector<double> A(NB * NB), S(NB * NB, 0.0), sd(NB + 1);
    for (auto& x : A) x = nd(rng);
    for (int i = 0; i < NB; ++i)
        for (int j = 0; j < NB; ++j) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int k = 0; k < NB; ++k) s += A[i * NB + k] * A[j * NB + k];
            S[i * NB + j] = s;
        }
    double smax = 0;
    for (int i = 0; i < NB; ++i) sd[i] = S[i * NB + i], smax = std::max(smax, sd[i]);
    sd[NB] = smax;
    double *dG, *dF, *dL, *dsd;
    int* dinfo;
    unsigned long long* dout;
    hipMalloc(&dG, sizeof(double) * NB * NB), hipMalloc(&dF, sizeof(double) * NB * NB), hipMalloc(&dL, sizeof(double) * NB * NB);
    hipMalloc(&dsd, sizeof(double) * (NB + 1)), hipMalloc(&dinfo, 4), hipMalloc(&dout, 128);
    hipMemcpy(dG, S.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
    hipMemcpy(dsd, sd.data(), sizeof(double) * (NB + 1), hipMemcpyHostToDevice);
    hipMemset(dinfo, 0x7f, 4);
    unsigned long long out[16];
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(dF, 0, sizeof(double) * NB * NB), hipMemset(dL, 0, sizeof(double) * NB * NB);
        k_time_diag2w<<<1, 128>>>(dG, dF, dL, dinfo, dsd, dout);
        hipDeviceSynchronize();
        hipMemcpy(out, dout, 32, hipMemcpyDeviceToHost);
        std::printf("2-wave rep %d: factor wave %llu cycles (%.2f us), inverse wave %llu cycles (%.2f us)\n", rep, out[0],
                    out[1] / 100.0, out[2], out[3] / 100.0);
    }
    check("2-wave", S, dF, dL);
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(dF, 0xff, sizeof(double) * NB * NB), hipMemset(dL, 0, sizeof(double) * NB * NB);
        k_time_diag4<<<1, 256>>>(dG, dF, dL, dinfo, dout);
        hipDeviceSynchronize();
        hipMemcpy(out, dout, 96, hipMemcpyDeviceToHost);
        std::printf("4-wave rep %d (cycles: role done / stores drained; us):", rep);
        for (int w = 0; w < 4; ++w) std::printf("  w%d %llu / %llu (%.2f us)", w, out[3 * w], out[3 * w + 1], out[3 * w + 2] / 100.0);
        std::printf("  %s\n", hipGetErrorString(hipGetLastError()));
    }
    check("4-wave", S, dF, dL);
    // a semi-definite block: rows/columns 5 and 20 zero -> zero columns of L, zero rows/columns of L^-1, no NaN
    std::vector<double> S2 = S;
    for (int z : {5, 20})
        for (int k = 0; k < NB; ++k) S2[z * NB + k] = S2[k * NB + z] = 0.0;
    hipMemcpy(dG, S2.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
    hipMemset(dF, 0xff, sizeof(double) * NB * NB), hipMemset(dL, 0, sizeof(double) * NB * NB);
    hipMemset(dinfo, 0x7f, 4);
    k_time_diag4<<<1, 256>>>(dG, dF, dL, dinfo, dout);
    hipDeviceSynchronize();
    {
        std::vector<double> L(NB * NB), X(NB * NB);
        int inf;
        hipMemcpy(L.data(), dF, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
        hipMemcpy(X.data(), dL, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
        hipMemcpy(&inf, dinfo, 4, hipMemcpyDeviceToHost);
        double e1 = 0, zc = 0;
        bool finite = true;
        for (int i = 0; i < NB; ++i)
            for (int j = 0; j < NB; ++j) {
                double s = 0;
                for (int k = 0; k < NB; ++k) s += L[i * NB + k] * L[j * NB + k];
                e1 = std::max(e1, std::fabs(s - S2[i * NB + j]));
                finite = finite && std::isfinite(L[i * NB + j]) && std::isfinite(X[i * NB + j]);
                if (j == 5 || j == 20 || i == 5 || i == 20) zc = std::max(zc, std::fabs(X[i * NB + j]));
                if (j == 5 || j == 20) zc = std::max(zc, std::fabs(L[i * NB + j]));
            }
        std::printf("4-wave, semi-definite: max |LL^T - S| = %.3e  dropped rows/cols max = %.1e  finite %d  info %s\n", e1, zc,
                    (int)finite, inf == 0x7f7f7f7f ? "ok" : "FLAGGED");
    }
    return 0;
}
