// Micro-benchmark: how fast does the GPU start workgroups, as a function of block size, LDS and VGPR use?
// (scripts/dev, not part of the product)  hipcc -O3 --offload-arch=gfx950 -o launch_rate launch_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDSW, int SPIN>
__global__ void k(unsigned *out) {
    __shared__ unsigned s[LDSW > 0 ? LDSW : 1];
    if (LDSW > 0) s[threadIdx.x % LDSW] = threadIdx.x;
    unsigned v = threadIdx.x;
    for (int i = 0; i < SPIN; ++i) { __builtin_amdgcn_s_sleep(16); v = v * 1664525u + 1013904223u; }
    if (LDSW > 0) { __syncthreads(); v += s[(threadIdx.x + 1) % LDSW]; }
    if (v == 0x12345678u) out[0] = v;
}
template <int LDSW, int SPIN>
void run(const char *name, int blocks, int threads, unsigned *d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) k<LDSW, SPIN><<<blocks, threads>>>(d);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) k<LDSW, SPIN><<<blocks, threads>>>(d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s blocks %6d x %4d thr  %8.1f us/launch  %7.1f ns/block  %6.1f ns/wave\n", name, blocks, threads, ms * 200, ms * 2e5 / blocks, ms * 2e5 / blocks / (threads / 64));
}
int main() {
    unsigned *d; hipMalloc(&d, 64);
    for (int blocks : {2048, 8192}) {
        run<0, 0>("lds 0 spin 0", blocks, 64, d);
        run<0, 0>("lds 0 spin 0", blocks, 256, d);
        run<0, 0>("lds 0 spin 0", blocks, 1024, d);
        run<2048, 0>("lds 8K spin 0", blocks, 256, d);
        run<4096, 0>("lds 16K spin 0", blocks, 256, d);
        run<8192, 0>("lds 32K spin 0", blocks, 256, d);
        run<8192, 0>("lds 32K spin 0", blocks, 512, d);
        run<8192, 0>("lds 32K spin 0", blocks, 1024, d);
        run<0, 50>("lds 0 spin 50 (~10us)", blocks, 256, d);
        run<4096, 50>("lds 16K spin 50", blocks, 256, d);
        run<8192, 50>("lds 32K spin 50", blocks, 256, d);
        run<8192, 50>("lds 32K spin 50", blocks, 1024, d);
    }
    return 0;
}
