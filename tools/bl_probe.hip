// Probe: raw-buffer loads (SRD + voffset + SGPR soffset) from inline asm -- do out-of-range loads return 0 (no fault), whole
// or per dword, and do 4-byte-aligned 16-byte loads work?   hipcc --offload-arch=gfx950 -O3 tools/bl_probe.hip -o tools/bin/bl_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_srd(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    i32x4 r;
    r.x = (int)(uint32_t)a;
    r.y = (int)((uint32_t)(a >> 32) & 0xffffu);
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ f32x4 bload(i32x4 srd, uint32_t voff, uint32_t soff) {
    f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(srd), "s"(soff) : "memory");
    return v;
}
__global__ void k(const float* p, float* out, uint32_t bytes, int n) {
    const i32x4 srd = make_srd(p, bytes);
    uint32_t soff = 0;
    f32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        f32x4 v = bload(srd, threadIdx.x * 16u, soff);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(v));
        acc += v;
        soff += 4096;
    }
    *(f32x4*)(out + (blockIdx.x * 256 + threadIdx.x) * 4) = acc;
}
int main() {
    float *p, *o; (void)hipMalloc(&p, 1 << 20); (void)hipMalloc(&o, 256 * 16); (void)hipMemset(p, 0, 1 << 20);
    float one[4] = {1, 2, 3, 4}; for (int i = 0; i < 1024; ++i) (void)hipMemcpy(p + i * 4, one, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, p, o, 8192u, 4);  // only the first 8 KB are in range: steps 0,1 read data, 2,3 read zeros
    float h[8]; (void)hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
    printf("%g %g %g %g (expect 2 4 6 8 if out-of-range loads return 0)\n", h[0], h[1], h[2], h[3]);
    // num_records ends 8 bytes into lane 0's second vector (offset 4096 + 8): per-dword range check -> x,y of step 1 read, z,w zero
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, p, o, 4096u + 8u, 2);
    (void)hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
    printf("%g %g %g %g (2 4 3 4: per-dword check; 1 2 3 4: whole vector dropped; 2 4 6 8: no check inside a vector)\n", h[0], h[1], h[2], h[3]);
    // 4-byte-aligned base (p + 1 float): lane 0 reads 2 3 4 1
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, p + 1, o, 8192u, 1);
    (void)hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
    printf("%g %g %g %g (expect 2 3 4 1: 16-byte load from a 4-byte-aligned address)\n", h[0], h[1], h[2], h[3]);
    return 0;
}
