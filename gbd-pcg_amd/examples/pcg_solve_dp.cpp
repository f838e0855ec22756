// fp64 example driver (counterpart of the reference's examples/pcg_solve_dp.cu).
#include "pcg_solve_common.hpp"
int main() { return run_example<double>(); }
