// Probe the operand/result lane maps of v_mfma_f32_4x4x1_16b_f32 on gfx950.
// hipcc --offload-arch=gfx950 -O2 mfma4x4_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {  // grid: 64*64 blocks (p,q), 64 threads
    const int p = blockIdx.x >> 6, q = blockIdx.x & 63, lane = threadIdx.x;
    const float a = lane == p ? 1.0f : 0.0f, b = lane == q ? 1.0f : 0.0f;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[((size_t)blockIdx.x * 64 + lane) * 4 + r] = c[r];
}
int main() {
    float* d;
    const size_t n = 64 * 64 * 64 * 4;
    hipMalloc(&d, n * sizeof(float));
    probe<<<64 * 64, 64>>>(d);
    std::vector<float> h(n);
    hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost);
    // for every (p,q) list where the 1 landed
    int shown = 0;
    for (int p = 0; p < 64; ++p)
        for (int q = 0; q < 64; ++q) {
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r)
                    if (h[(((size_t)p * 64 + q) * 64 + lane) * 4 + r] != 0.0f) {
                        if ((p < 8 && q < 8) || (p % 13 == 5 && q % 7 == 3) || shown < 0)
                            printf("A lane %2d x B lane %2d -> D lane %2d reg %d\n", p, q, lane, r);
                        // check hypothesis: block = p/4 == q/4, i = p%4, j = q%4 -> lane 4*block + j, reg i
                        const bool ok = (p / 4 == q / 4) && lane == (p / 4) * 4 + (q % 4) && r == p % 4;
                        if (!ok) { printf("HYPOTHESIS VIOLATED at p=%d q=%d lane=%d r=%d\n", p, q, lane, r); shown++; }
                    }
        }
    int nonzero_pairs = 0;
    for (int p = 0; p < 64; ++p)
        for (int q = 0; q < 64; ++q) {
            bool any = false;
            for (int k = 0; k < 256; ++k) any |= h[((size_t)p * 64 + q) * 256 + k] != 0.0f;
            nonzero_pairs += any;
        }
    printf("pairs with output: %d (expect 16 blocks x 16 = 256); violations %d\n", nonzero_pairs, shown);
    return 0;
}
