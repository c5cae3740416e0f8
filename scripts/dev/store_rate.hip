// Development probe: how fast do three output maps go out, by store width and kind?
// (hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip; run on the GPU box)
// Pattern of the fill kernels: N pixels, three float arrays, a wave stores runs of consecutive pixels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int V, bool NT>
__global__ __launch_bounds__(256) void k_store(float *a, float *b, float *c, size_t n, int per_thread) {
    // a block handles 256 * V * per_thread consecutive pixels, a wave's store = 64 * V consecutive pixels
    size_t base = (size_t)blockIdx.x * 256 * V * per_thread;
    for (int it = 0; it < per_thread; ++it) {
        const size_t p = base + ((size_t)it * 256 + threadIdx.x) * V;
        if (p + V > n) return;
        const float v = (float)(p & 1023);
        if constexpr (V == 1) {
            if (NT) { __builtin_nontemporal_store(v, a + p); __builtin_nontemporal_store(v, b + p); __builtin_nontemporal_store(v, c + p); }
            else { a[p] = v; b[p] = v; c[p] = v; }
        } else if constexpr (V == 2) {
            const f2 w = {v, v};
            if (NT) { __builtin_nontemporal_store(w, (f2 *)(a + p)); __builtin_nontemporal_store(w, (f2 *)(b + p)); __builtin_nontemporal_store(w, (f2 *)(c + p)); }
            else { *(f2 *)(a + p) = w; *(f2 *)(b + p) = w; *(f2 *)(c + p) = w; }
        } else {
            const f4 w = {v, v, v, v};
            if (NT) { __builtin_nontemporal_store(w, (f4 *)(a + p)); __builtin_nontemporal_store(w, (f4 *)(b + p)); __builtin_nontemporal_store(w, (f4 *)(c + p)); }
            else { *(f4 *)(a + p) = w; *(f4 *)(b + p) = w; *(f4 *)(c + p) = w; }
        }
    }
}

template <int V, bool NT>
void run(float *a, float *b, float *c, size_t n, int per_thread, const char *name) {
    const int blocks = (int)((n + (size_t)256 * V * per_thread - 1) / ((size_t)256 * V * per_thread));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) k_store<V, NT><<<blocks, 256>>>(a, b, c, n, per_thread);
    hipEventRecord(e0);
    const int K = 50;
    for (int i = 0; i < K; ++i) k_store<V, NT><<<blocks, 256>>>(a, b, c, n, per_thread);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s per_thread %2d blocks %6d  %7.1f us  %6.0f GB/s\n", name, per_thread, blocks, ms / K * 1e3, 12.0 * n / (ms / K) / 1e6);
}

int main() {
    const size_t n = (size_t)32 * 352 * 1216;  // the headline batch
    float *a, *b, *c;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4);
    for (int pt : {1, 4, 15, 60}) {
        run<1, false>(a, b, c, n, pt, "dword");
        run<1, true>(a, b, c, n, pt, "dword nontemporal");
        run<2, false>(a, b, c, n, pt, "dwordx2");
        run<2, true>(a, b, c, n, pt, "dwordx2 nontemporal");
        run<4, false>(a, b, c, n, pt, "dwordx4");
        run<4, true>(a, b, c, n, pt, "dwordx4 nontemporal");
    }
    return 0;
}
