// fp32 example driver (counterpart of the reference's examples/pcg_solve.cu).
#include "pcg_solve_common.hpp"
int main() { return run_example<float>(); }
