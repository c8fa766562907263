#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void k(const unsigned* a, const unsigned* b, float* o) {
    unsigned x = a[threadIdx.x], y = b[threadIdx.x];
    float c1 = 10.f, c2 = 10.f, c3 = 10.f;
    c1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x), __builtin_bit_cast(bf16x2, y), c1, false);
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(c2) : "v"(x), "v"(y));
#ifdef V3
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(c3) : "v"(x), "v"(y), "v"(c3));
#endif
    o[threadIdx.x * 3] = c1; o[threadIdx.x * 3 + 1] = c2; o[threadIdx.x * 3 + 2] = c3;
}
int main() {
    unsigned ha[64], hb[64]; float ho[192];
    // a = (1.0, 2.0) bf16 pairs: 1.0 = 0x3F80, 2.0 = 0x4000; b = (3.0, 0.5): 3.0 = 0x4040, 0.5 = 0x3F00
    for (int i = 0; i < 64; ++i) { ha[i] = 0x3F80u | (0x4000u << 16); hb[i] = 0x4040u | (0x3F00u << 16); }
    unsigned *da, *db; float* d;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&d, 768);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, d);
    hipMemcpy(ho, d, 768, hipMemcpyDeviceToHost);
    printf("expect 10 + 1*3 + 2*0.5 = 14: builtin %f  asm dot2c %f  asm dot2 %f\n", ho[0], ho[1], ho[2]);
    return 0;
}
