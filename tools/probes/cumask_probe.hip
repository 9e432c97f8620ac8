// Which XCDs / CUs does a HIP stream created with hipExtStreamCreateWithCUMask run on?  Several maps in
// flight could each get their own group of XCDs (own L2s) instead of interleaving on all 256 CUs.
//   hipcc --offload-arch=gfx950 -O3 cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void where(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[2 * blockIdx.x] = xcc & 0xF;
        out[2 * blockIdx.x + 1] = hwid;
    }
    // stay resident for a while so that the blocks spread over every CU the stream may use
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000ull) {}
}

static void run(const char* name, const std::vector<unsigned>& mask) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    unsigned* d;
    const int nb = 2048;
    hipMalloc(&d, nb * 2 * sizeof(unsigned));
    where<<<nb, 64, 0, s>>>(d);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(nb * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    int xcc[16] = {0};
    std::vector<int> cus(16 * 64, 0);
    for (int b = 0; b < nb; ++b) {
        const unsigned x = h[2 * b], id = h[2 * b + 1];
        xcc[x]++;
        const unsigned cu = (id >> 8) & 0xF, sh = (id >> 12) & 1, se = (id >> 13) & 7;   // gfx9 HW_ID layout
        cus[x * 64 + (se * 2 + sh) * 16 + cu % 16]++;
    }
    int distinct = 0;
    for (int v : cus) distinct += v > 0;
    printf("%-34s blocks per XCC:", name);
    for (int i = 0; i < 8; ++i) printf(" %4d", xcc[i]);
    printf("   distinct (xcc,se,sh,cu): %d\n", distinct);
    hipFree(d);
    hipStreamDestroy(s);
}

int main() {
    std::vector<unsigned> m(8);
    m.assign(8, 0xFFFFFFFFu); run("all 256 bits", m);
    m.assign(8, 0); for (int i = 0; i < 4; ++i) m[i] = 0xFFFFFFFFu; run("bits 0..127", m);
    m.assign(8, 0); for (int i = 4; i < 8; ++i) m[i] = 0xFFFFFFFFu; run("bits 128..255", m);
    m.assign(8, 0x55555555u); run("even bits", m);
    m.assign(8, 0); m[0] = 0xFFFFFFFFu; run("bits 0..31", m);
    m.assign(8, 0x01010101u); run("every 8th bit (bit % 8 == 0)", m);
    m.assign(8, 0x03030303u); run("bit % 8 in {0,1}", m);
    m.assign(8, 0x0F0F0F0Fu); run("bit % 8 in {0..3}", m);
    return 0;
}
