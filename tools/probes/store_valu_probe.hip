// Do 16-byte streaming stores overlap with VALU work of the same SIMD?  The warp+variance kernel writes
// 503 MB with one 1-KB wave-store per ~150 VALU instructions and runs at 3.2 TB/s; removing the stores
// saves a third of its time, removing VALU work saves nothing.  This probe runs the same store pattern
// (volume [4][D][hw][8] fp32, a wave = 8 pixels x 32 channels, 24 depth steps per wave) with NV packed
// FMAs per step and reports time for: stores only, VALU only, both.
//   hipcc --offload-arch=gfx950 -O3 store_valu_probe.hip -o /tmp/sv_probe && /tmp/sv_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, bool STORE>
__global__ __launch_bounds__(256) void probe(float* __restrict__ vol, int D, int hw, int slab, float seed) {
    const int sub = threadIdx.x & 7;
    const int pl = sub >> 1, cin = (sub & 1) * 4;
    const int p = blockIdx.x * 32 + (threadIdx.x >> 3);
    const int d0 = blockIdx.y * slab;
    f32x2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = (f32x2){seed + i + threadIdx.x, seed - i};
    const f32x2 m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
    float* out = vol + (((size_t)pl * D + d0) * hw + p) * 8 + cin;
    for (int d = 0; d < slab; ++d) {
#pragma unroll
        for (int k = 0; k < NV / 8; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        if (STORE) {
            *reinterpret_cast<f32x4*>(out) = (f32x4){a[0].x, a[1].y, a[2].x, a[3].y};
            out += (size_t)hw * 8;
        }
    }
    if (!STORE || a[7].x == 12345.678f) vol[(size_t)blockIdx.x * 256 + threadIdx.x] = a[0].x + a[4].y;
}

template <int NV, bool STORE>
float run(float* vol, int D, int hw) {
    const int slab = 24;
    dim3 grid(hw / 32, D / slab);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) probe<NV, STORE><<<grid, 256>>>(vol, D, hw, slab, 1.0f);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) probe<NV, STORE><<<grid, 256>>>(vol, D, hw, slab, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const int D = 192, hw = 128 * 160;
    float* vol;
    const size_t bytes = (size_t)4 * D * hw * 8 * 4;
    hipMalloc(&vol, bytes);
    hipMemset(vol, 0, bytes);
    printf("volume %.1f MB, %d wave-stores of 1 KB\n", bytes / 1e6, (int)(bytes / 1024));
    printf("stores only            : %.4f ms (%.2f TB/s)\n", run<0, true>(vol, D, hw), bytes / run<0, true>(vol, D, hw) / 1e9);
    printf("VALU  64/step, no store: %.4f ms\n", run<64, false>(vol, D, hw));
    printf("VALU  64/step + stores : %.4f ms\n", run<64, true>(vol, D, hw));
    printf("VALU 160/step, no store: %.4f ms\n", run<160, false>(vol, D, hw));
    printf("VALU 160/step + stores : %.4f ms\n", run<160, true>(vol, D, hw));
    printf("VALU 320/step, no store: %.4f ms\n", run<320, false>(vol, D, hw));
    printf("VALU 320/step + stores : %.4f ms\n", run<320, true>(vol, D, hw));
    return 0;
}
