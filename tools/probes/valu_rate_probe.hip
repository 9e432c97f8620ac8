// VALU issue-rate probe for gfx950: cycles per wave-instruction of v_fma_f32, v_pk_fma_f32,
// v_pk_mul_f32, v_mov_b32 dpp and v_cndmask at 1, 2 and 4 waves per SIMD (s_memtime around an
// unrolled loop of independent instructions).  Answers: is a packed f32 op one issue slot or two?
//   hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 2000;
constexpr int kUnroll = 16;

template <int KIND>
__global__ void probe(float* out, unsigned long long* cyc, float seed) {
    f32x2 a[kUnroll];
    for (int i = 0; i < kUnroll; ++i) a[i] = (f32x2){seed + i + threadIdx.x, seed - i};
    const f32x2 m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) {
            if (KIND == 0) {          // v_fma_f32 (one per element: two per slot of a[])
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(m.y), "v"(c.y));
            } else if (KIND == 1) {   // v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            } else if (KIND == 2) {   // v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            } else if (KIND == 3) {   // v_mov_b32 dpp quad_perm
                a[i].x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[i].x), 0x55, 0xF, 0xF, true));
                a[i].y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a[i].y), 0xAA, 0xF, 0xF, true));
            } else if (KIND == 4) {   // v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            } else {                  // v_mov_b64
                asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) % kUnroll]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < kUnroll; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_elem) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipMalloc(&cyc, 1024 * sizeof(unsigned long long));
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 256 * waves_per_simd;   // one block per CU, 4 SIMDs
        probe<KIND><<<256, threads>>>(out, cyc, 1.0f);
        hipDeviceSynchronize();
        probe<KIND><<<256, threads>>>(out, cyc, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256);
        hipMemcpy(h.data(), cyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : h) sum += (double)v;
        const double per_wave_instr = sum / 256 / ((double)kIters * kUnroll * per_elem);
        printf("%-14s %d waves/SIMD: %.2f cycles per instruction per wave, %.2f per SIMD issue slot\n", name,
               waves_per_simd, per_wave_instr, per_wave_instr / waves_per_simd);
    }
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32", 2);
    run<1>("v_pk_fma_f32", 1);
    run<2>("v_pk_mul_f32", 1);
    run<4>("v_pk_add_f32", 1);
    run<3>("v_mov_dpp", 2);
    run<5>("v_mov_b64", 1);
    return 0;
}
