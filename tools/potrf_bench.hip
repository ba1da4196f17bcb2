// Standalone timing harness for the 32x32 diagonal-block factorisation (not part of the product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/potrf_bench.hip -o /tmp/potrf_bench
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
    hipMalloc(&dsd, sizeof(double) * (NB + 1)), hipMalloc(&dinfo, 4), hipMalloc(&dout, 16);
    hipMemcpy(dG, S.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
    hipMemcpy(dsd, sd.data(), sizeof(double) * (NB + 1), hipMemcpyHostToDevice);
    hipMemset(dinfo, 0x7f, 4);
    unsigned long long out[4];
    hipFree(dout);
    hipMalloc(&dout, 32);
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(dF, 0, sizeof(double) * NB * NB), hipMemset(dL, 0, sizeof(double) * NB * NB);
        k_time_diag2w<<<1, 128>>>(dG, dF, dL, dinfo, dsd, dout);
        hipDeviceSynchronize();
        hipMemcpy(out, dout, 32, hipMemcpyDeviceToHost);
#ifdef PNMOL_STAMP
        {
            unsigned long long st[40];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(pnmol_stamp_out), sizeof(st));
            std::printf("per-column cycles:");
            for (int j = 0; j < 32; ++j) std::printf(" %llu", st[j + 1] - st[j]);
            std::printf("\n");
        }
#endif
        std::printf("2-wave rep %d: factor wave %llu cycles (%.2f us), inverse wave %llu cycles (%.2f us)\n", rep, out[0],
                    out[1] / 100.0, out[2], out[3] / 100.0);
    }
    {
        std::vector<double> L(NB * NB), X(NB * NB);
        hipMemcpy(L.data(), dF, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
        hipMemcpy(X.data(), dL, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0;
        for (int i = 0; i < NB; ++i)
            for (int j = 0; j < NB; ++j) {
                double s = 0, t = 0;
                for (int k = 0; k < NB; ++k) s += L[i * NB + k] * L[j * NB + k], t += X[i * NB + k] * L[k * NB + j];
                e1 = std::max(e1, std::fabs(s - S[i * NB + j]));
                e2 = std::max(e2, std::fabs(t - (i == j)));
            }
        std::printf("2-wave: max |LL^T - S| = %.3e   max |Linv L - I| = %.3e\n", e1, e2);
    }
    std::vector<double> L(NB * NB), X(NB * NB);
    hipMemcpy(L.data(), dF, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
    hipMemcpy(X.data(), dL, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < NB; ++i)
        for (int j = 0; j < NB; ++j) {
            double s = 0, t = 0;
            for (int k = 0; k < NB; ++k) s += L[i * NB + k] * L[j * NB + k], t += X[i * NB + k] * L[k * NB + j];
            e1 = std::max(e1, std::fabs(s - S[i * NB + j]));
            e2 = std::max(e2, std::fabs(t - (i == j)));
        }
    std::printf("max |LL^T - S| = %.3e   max |Linv L - I| = %.3e\n", e1, e2);
    return 0;
}
