// What does an exec-masked 16-byte gather cost?  The tap-cache warp kernel re-gathers with ~20 % of a
// wave's lanes active; if the texture path charges a masked buffer_load_dwordx4 like a full one, the
// number of load INSTRUCTIONS (not bytes) bounds that kernel.  Cycles per load instruction per wave for
// 100 % / 25 % (whole quads) / 25 % (one lane per quad) / 6 % active lanes, data resident in L1/L2.
//   hipcc --offload-arch=gfx950 -O3 vmem_mask_probe.hip -o /tmp/vmem_probe && /tmp/vmem_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kIters = 400, kUnroll = 8;

template <int MODE>
__global__ void probe(const float* __restrict__ src, float* out, unsigned long long* cyc, int span_f4) {
    const int lane = threadIdx.x & 63;
    bool active = true;
    if (MODE == 1) active = ((lane >> 2) & 3) == 0;        // 25 %: one quad in four
    if (MODE == 2) active = (lane & 3) == 0;               // 25 %: one lane per quad
    if (MODE == 3) active = (lane & 15) == 0;              // 6 %
    const f32x4* p = reinterpret_cast<const f32x4*>(src);
    unsigned idx = (threadIdx.x * 7 + blockIdx.x * 131) % span_f4;
    f32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
        if (active) {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const f32x4 v = p[(idx + u * 37) % span_f4];
                acc += v;
            }
        }
        idx = (idx + 97) % span_f4;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const float* src, int span_f4) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipMalloc(&cyc, 1024 * sizeof(unsigned long long));
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 256 * waves_per_simd;
        probe<MODE><<<256, threads>>>(src, out, cyc, span_f4);
        hipDeviceSynchronize();
        probe<MODE><<<256, threads>>>(src, out, cyc, span_f4);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256);
        hipMemcpy(h.data(), cyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : h) sum += (double)v;
        const double per_load = sum / 256 / ((double)kIters * kUnroll);
        printf("%-28s span %7d KB, %d waves/SIMD: %.1f cycles per load instruction per wave, %.2f per CU slot\n", name,
               span_f4 * 16 / 1024, waves_per_simd, per_load, per_load / (4 * waves_per_simd));
    }
    hipFree(out);
    hipFree(cyc);
}

int main() {
    float* src;
    const size_t n = 64u << 20;   // 64 MB of floats
    hipMalloc(&src, n);
    hipMemset(src, 0, n);
    for (int span_kb : {16, 2048}) {   // L1-resident / L2-resident working set
        const int span_f4 = span_kb * 1024 / 16;
        run<0>("all 64 lanes", src, span_f4);
        run<1>("16 lanes (whole quads)", src, span_f4);
        run<2>("16 lanes (1 per quad)", src, span_f4);
        run<3>("4 lanes", src, span_f4);
    }
    return 0;
}
