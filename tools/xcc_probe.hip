// Which XCD does block b land on?  (the SpMM schedule assumes blocks b and b+8 share an XCD)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    for (int nb : {64, 1024, 2048}) {
        int* d; hipMalloc(&d, nb * 4);
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d);
        int* h = (int*)malloc(nb * 4); hipMemcpy(h, d, nb * 4, hipMemcpyDeviceToHost);
        printf("grid %d: first 24 blocks:", nb);
        for (int i = 0; i < 24; ++i) printf(" %d", h[i] & 0xf);
        int bad = 0;
        for (int i = 8; i < nb; ++i) bad += ((h[i] & 0xf) != (h[i - 8] & 0xf));
        printf("   blocks whose XCC differs from block b-8: %d of %d (raw %08x)\n", bad, nb - 8, h[0]);
    }
    return 0;
}
