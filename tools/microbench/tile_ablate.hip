// Diagnostic: time solve_tile_kernel<float,7,25,1> at the C2 shape with loop phases stubbed out
// (SSN_ABLATE bit mask: 1 nonlinearity, 2 DPP reduction, 4 stop flags, 8 LDS r reads, 16 barrier, 32 LDS W reads of
// the split shape).  argv[2] = kernel shape (0 library default, 1 split, 2 all-register).
// argv[1] = number of weight draws (256 = one workgroup per CU, 512 = two, 768 = three per CU).
// Results are WRONG by construction; only the timing matters.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSSN_ABLATE=<mask> -o tile_ablate_<mask> tile_ablate.hip
#include "../../tc_gan_amd/csrc/ssn_tile.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, M = 200, NB = 1, T = 2000;
    const int shape = argc > 2 ? atoi(argv[2]) : 0;
    std::vector<float> hW((size_t)B * M * M), hext(M, 1.0f);
    for (size_t i = 0; i < hW.size(); ++i) hW[i] = (((i * 2654435761u) % 1000) / 1000.f - 0.6f) * 0.01f;
    float *W, *ext, *r; int *codes, *steps;
    hipMalloc(&W, hW.size() * 4); hipMalloc(&ext, M * 4); hipMalloc(&r, (size_t)B * M * 4);
    hipMalloc(&codes, B * 4); hipMalloc(&steps, B * 4);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ext, hext.data(), M * 4, hipMemcpyHostToDevice);
    hipMemset(r, 0, (size_t)B * M * 4);
    ssn_solver_params p{SSN_IO_TANH, T, 0.01, 2.2, 0.01589, 0.002, 8e-4, 0.0, 200.0, 1000.0};
    ssn::SolveArgs<float> a;
    a.W = W; a.ext = ext; a.r = r; a.r_prev = nullptr; a.codes = codes; a.steps = steps; a.ext_per_draw = 0;
    a.B = B; a.NB = NB; a.M = M; a.N = M / 2; a.io = ssn::make_io_consts<float>(p); a.st = ssn::make_step_consts<float>(p);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    ssn::launch_tile<float>(a, nullptr, shape); hipDeviceSynchronize();
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
        hipMemset(r, 0, (size_t)B * M * 4);
        hipEventRecord(e0); ssn::launch_tile<float>(a, nullptr, shape); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("SSN_ABLATE=%2d shape=%d B=%d  %.3f ms  = %.0f cycles@2.4GHz per Euler step of one launch wave (B<=768)\n",
           SSN_ABLATE, shape, B, best, best * 1e-3 * 2.4e9 / T);
    return 0;
}
