// Probe: is the v_dot2c_f32_bf16 residual (x - bf16(x)) exact?  Compares the 3-way split built with
// dot2c against the shift/and/sub split, bit for bit.  Build: hipcc --offload-arch=gfx950 -shared -fPIC.
#include <hip/hip_runtime.h>
__device__ __forceinline__ unsigned pk(float lo, float hi) {
    unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r;
}
__global__ void probe(const float2* x, uint2* ref, uint2* alt, unsigned* p3ref, unsigned* p3alt, long n2) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    float2 v = x[i];
    unsigned p1 = pk(v.x, v.y);
    float rx = v.x - __uint_as_float(p1 << 16), ry = v.y - __uint_as_float(p1 & 0xffff0000u);
    unsigned p2 = pk(rx, ry);
    float sx = rx - __uint_as_float(p2 << 16), sy = ry - __uint_as_float(p2 & 0xffff0000u);
    ref[i] = make_uint2(p1, p2); p3ref[i] = pk(sx, sy);
    float ax = v.x, ay = v.y;
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(ax) : "v"(p1), "v"(0x0000BF80u));
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(ay) : "v"(p1), "v"(0xBF800000u));
    unsigned q2 = pk(ax, ay);
    float bx = ax, by = ay;
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(bx) : "v"(q2), "v"(0x0000BF80u));
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(by) : "v"(q2), "v"(0xBF800000u));
    alt[i] = make_uint2(p1, q2); p3alt[i] = pk(bx, by);
}
extern "C" int probe_split(const float* x, void* ref, void* alt, void* p3ref, void* p3alt, long n, void* stream) {
    long n2 = n / 2;
    hipLaunchKernelGGL(probe, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)x, (uint2*)ref, (uint2*)alt, (unsigned*)p3ref, (unsigned*)p3alt, n2);
    return (int)hipGetLastError();
}
