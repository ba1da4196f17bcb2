// Standalone timing of k_downdate<3> (N=512 sizes) for several chunk ranges.
#include "../pnmol-experiments_amd/csrc/pnmol_hip.hip"
#include <random>
int main() {
    const int dp = 512, mp = 544, n = 3; const long Dp = (long)n * dp;
    std::vector<double> hW((size_t)Dp * mp), hP((size_t)Dp * Dp);
    std::mt19937 rng(3); std::uniform_real_distribution<double> u(-1, 1);
    for (auto& x : hW) x = u(rng) * 0.01;
    for (auto& x : hP) x = u(rng);
    double *W, *P, *Po, *var;
    hipMalloc(&W, sizeof(double) * hW.size()); hipMalloc(&P, sizeof(double) * hP.size());
    hipMalloc(&Po, sizeof(double) * hP.size()); hipMalloc(&var, sizeof(double) * Dp);
    hipMemcpy(W, hW.data(), sizeof(double) * hW.size(), hipMemcpyHostToDevice);
    hipMemcpy(P, hP.data(), sizeof(double) * hP.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ranges[][2] = {{0, 0}, {0, 8}, {0, 24}, {0, 68}, {24, 68}};
    for (auto& rg : ranges) {
        for (int rep = 0; rep < 3; ++rep) k_downdate<3><<<dim3(dp / 16, dp / 16), 256>>>(P, W, Po, var, dp, mp, rg[0], rg[1], VecArgs{});
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 20; ++rep) k_downdate<3><<<dim3(dp / 16, dp / 16), 256>>>(P, W, Po, var, dp, mp, rg[0], rg[1], VecArgs{});
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::printf("chunks [%2d,%2d): %.1f us per launch\n", rg[0], rg[1], ms * 1e3 / 20);
    }
    // correctness spot check of the full range against the host
    k_downdate<3><<<dim3(dp / 16, dp / 16), 256>>>(P, W, Po, var, dp, mp, 0, 68, VecArgs{});
    std::vector<double> out(hP.size());
    hipMemcpy(out.data(), Po, sizeof(double) * out.size(), hipMemcpyDeviceToHost);
    double emax = 0;
    for (int t = 0; t < 2000; ++t) {
        long i = rng() % Dp, j = rng() % Dp;
        long pi = ((i % dp) / 16 >= (j % dp) / 16) ? i : j, pj = (pi == i) ? j : i;   // the lower point tile is the source
        double s = 0;
        for (int k = 0; k < mp; ++k) s += hW[i * mp + k] * hW[j * mp + k];
        emax = std::max(emax, std::fabs(out[i * Dp + j] - (hP[pi * Dp + pj] - s)));
    }
    std::printf("max err vs host (2000 samples): %.3e\n", emax);
    return 0;
}
