// dpp_probe.hip -- unit check of the cross-lane reductions used by bt_sym.hpp (DPP quad_perm / row_half_mirror /
// row_ror) and of v_permlane16/32_swap, whose second result came back wrong on this toolchain (why bt_sym.hpp
// uses xor shuffles for the cross-row part).  Prints the lane values; expected patterns are in the labels.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dpp_add_f(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, moved);
}
__global__ void k(float* o)
{
    const int l = threadIdx.x;
    float v = (float)(1 << (l & 7)) + 1000.f * (l >> 3);   // per-lane tag: bit rp, thousands = group
    float a = dpp_add_f<0xB1>(v); float b = dpp_add_f<0x4E>(a); float c = dpp_add_f<0x141>(b);
    o[l] = c;                                   // expect 255 + 8000*g
    float w = (float)(l >> 3) * 1.0f + 100.f * (l & 7);      // group id + 100*rp ; sum over groups = 28 + 800*rp
    float r = dpp_add_f<0x128>(w);
    o[64 + l] = r;
    unsigned u = __builtin_bit_cast(unsigned, r);
    unsigned ucopy = u;
    asm volatile("" : "+v"(ucopy));   // distinct SSA value for the second operand
    auto s16 = __builtin_amdgcn_permlane16_swap(u, ucopy, false, false);
    float r2 = __builtin_bit_cast(float, s16[0]) + __builtin_bit_cast(float, s16[1]);
    o[128 + l] = r2;
    u = __builtin_bit_cast(unsigned, r2);
    unsigned ucopy2 = u;
    asm volatile("" : "+v"(ucopy2));
    auto s32 = __builtin_amdgcn_permlane32_swap(u, ucopy2, false, false);
    o[192 + l] = __builtin_bit_cast(float, s32[0]) + __builtin_bit_cast(float, s32[1]);
    o[256 + l] = __builtin_bit_cast(float, s16[0]);
    o[320 + l] = __builtin_bit_cast(float, s16[1]);
}
int main()
{
    float* d; hipMalloc(&d, 384 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[384]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[6] = {"sum8 (expect 255+8000g)", "row_ror8 add", "after permlane16 (sum)", "after permlane32 (expect 28+800rp)", "p16[0]", "p16[1]"};
    for (int t = 0; t < 6; ++t) { printf("%s\n", nm[t]); for (int l = 0; l < 64; ++l) printf("%7.0f%s", h[t * 64 + l], (l & 15) == 15 ? "\n" : ""); }
    return 0;
}
