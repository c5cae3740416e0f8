// Development probe: how fast can one frame batch be read once (the mask kernel's job)?  dwordx4 loads, U of them in flight per
// lane, one wave per row of 1216 pixels or flat.  (hipcc --offload-arch=gfx950 -O3 -o read_rate read_rate.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void k_read(const f4 *x, size_t n4, unsigned *out) {
    // a block reads 256 * U consecutive float4
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = base + (size_t)u * 256 < n4 ? x[base + (size_t)u * 256] : f4{0, 0, 0, 0};
    unsigned m = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) m |= (v[u].x > 0.1f) | (v[u].y > 0.1f) << 1 | (v[u].z > 0.1f) << 2 | (v[u].w > 0.1f) << 3;
    const unsigned long long b = __ballot(m != 0);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = (unsigned)__popcll(b);
}

template <int U>
void run(const f4 *x, size_t n4, unsigned *out) {
    const int blocks = (int)((n4 + 256 * U - 1) / (256 * U));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) k_read<U><<<blocks, 256>>>(x, n4, out);
    (void)hipEventRecord(e0);
    const int K = 50;
    for (int i = 0; i < K; ++i) k_read<U><<<blocks, 256>>>(x, n4, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("dwordx4 x %d in flight  blocks %6d  %7.1f us  %6.0f GB/s\n", U, blocks, ms / K * 1e3, 16.0 * n4 / (ms / K) / 1e6);
}

int main() {
    const size_t n = (size_t)32 * 352 * 1216, n4 = n / 4;
    f4 *x; unsigned *out;
    (void)hipMalloc(&x, n * 4); (void)hipMalloc(&out, 1 << 22);
    (void)hipMemset(x, 0, n * 4);
    run<1>(x, n4, out); run<2>(x, n4, out); run<4>(x, n4, out); run<5>(x, n4, out); run<8>(x, n4, out); run<16>(x, n4, out);
    return 0;
}
