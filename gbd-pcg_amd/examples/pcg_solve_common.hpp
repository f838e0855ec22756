// Shared body of the two example drivers (rewrite of the reference's examples/pcg_solve.cu and
// examples/pcg_solve_dp.cu host drivers against this library).  The 2-state, 3-knot system is the
// input data those examples hold (examples/pcg_solve.cu:14-25).
#pragma once
#include <cstdio>
#include <iostream>

#include "gpu_pcg.cuh"  // resolves to include/gbdpcg.hpp

template <typename T> int run_example()
{
    const uint32_t state_size = 2;
    const uint32_t knot_points = 3;

    // [L_k | D_k | R_k] per knot, column-major 2x2 blocks; L_0 and R_2 are unused
    T h_S[36] = {0,     0,     0,     0,      -.999,  0,     0,     -.999,   .999, .0999, -.98, .999,
                 .999,  -.98,  .0999, .999,   -2.008, .8801, .8801, -3.0584, .999, .0999, -.98, .999,
                 .999,  -.98,  .0999, .999,   -1.019, .8801, .8801, -2.0694, 0,    0,     0,    0};
    T h_gamma[6] = {3.1385, 0, 0, 3.0788, .0031, 3.0788};
    T h_lambda[6] = {0, 0, 0, 0, 0, 0};

    pcg_config<T> config;  // tol 1e-6, 25 iterations, identity preconditioner (empty_pinv = 1)
    uint32_t res = solvePCG<T>(h_S, h_gamma, h_lambda, state_size, knot_points, &config);

    std::cout << "GBD-PCG returned in " << res << " iters." << std::endl;
    std::cout << "Lambda: " << std::endl;
    for (int i = 0; i < 6; i++) std::cout << h_lambda[i] << " ";
    std::cout << std::endl;

    // same system through the stair preconditioner built on the device (empty_pinv = 0)
    T h_lambda2[6] = {0, 0, 0, 0, 0, 0};
    pcg_config<T> config2;
    config2.empty_pinv = 0;
    uint32_t res2 = solvePCG<T>(h_S, h_gamma, h_lambda2, state_size, knot_points, &config2);
    std::cout << "With the symmetric-stair preconditioner: " << res2 << " iters." << std::endl;
    std::cout << "Lambda: " << std::endl;
    for (int i = 0; i < 6; i++) std::cout << h_lambda2[i] << " ";
    std::cout << std::endl;
    return 0;
}
