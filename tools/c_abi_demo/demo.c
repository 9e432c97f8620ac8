/* demo.c -- the C ABI of libmvs_hip.so used from plain C, no Python / torch in the process.
 *
 * What a C / C++ / Go(cgo) / Rust(FFI) host does to run the MVSNet depth path (reference
 * models/mvsnet.py:145-218) on one batch item: allocate device memory with the HIP runtime, pack the
 * weights once on the host, call mvs_depth_infer, copy depth + confidence back.
 *
 *   demo <in.bin> <out.bin>
 * in.bin  : int32 N, D, h, w; then fp32 feats[N][32][h][w], proj[N][4][4], depth_values[D];
 *           then the 11 conv weights, 10 x 4 BN vectors and the prob bias in the order
 *           mvs_pack_weights takes them (written by tests/test_gpu_c_abi.py)
 * out.bin : fp32 depth[h][w], conf[h][w]
 * Build   : gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude demo.c -o demo \
 *               -L/opt/rocm/lib -lamdhip64 -ldl          (tools/c_abi_demo/Makefile)
 */
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mvs_abi.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static const int kCh[11][2] = {{32, 8}, {8, 16}, {16, 16}, {16, 32}, {32, 32}, {32, 64}, {64, 64},
                               {64, 32}, {32, 16}, {16, 8}, {8, 1}};

static float* read_floats(FILE* f, size_t n) {
    float* p = (float*)malloc(n * sizeof(float));
    if (!p || fread(p, sizeof(float), n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: demo libmvs_hip.so in.bin out.bin\n"); return 1; }
    void* so = dlopen(argv[1], RTLD_NOW);
    if (!so) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    int (*query_ws)(int, int, int, int, int, int, size_t*) = dlsym(so, "mvs_query_workspace");
    int (*query_blob)(size_t*) = dlsym(so, "mvs_query_weights_blob");
    int (*pack)(const float* const*, const float* const*, const float*, float, void*, size_t) = dlsym(so, "mvs_pack_weights");
    int (*infer)(const float*, const float*, const float*, const void*, float*, float*, void*, size_t,
                 int, int, int, int, int, int, void*) = dlsym(so, "mvs_depth_infer");
    const char* (*errstr)(void) = dlsym(so, "mvs_last_error_string");
    if (!query_ws || !query_blob || !pack || !infer || !errstr) { fprintf(stderr, "missing symbol\n"); return 1; }

    FILE* f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 1; }
    int32_t dims[4];
    if (fread(dims, sizeof(int32_t), 4, f) != 4) return 3;
    const int N = dims[0], D = dims[1], h = dims[2], w = dims[3];
    const size_t nf = (size_t)N * 32 * h * w;
    float* feats = read_floats(f, nf);
    float* proj = read_floats(f, (size_t)N * 16);
    float* dv = read_floats(f, (size_t)D);
    const float* convs[11];
    const float* bns[40];
    for (int l = 0; l < 11; ++l) convs[l] = read_floats(f, (size_t)kCh[l][0] * kCh[l][1] * 27);
    for (int l = 0; l < 10; ++l)
        for (int j = 0; j < 4; ++j) bns[4 * l + j] = read_floats(f, (size_t)kCh[l][1]);
    float* prob_bias = read_floats(f, 1);
    fclose(f);

    size_t blob_bytes = 0, ws_bytes = 0;
    if (query_blob(&blob_bytes) || query_ws(N, 32, D, h, w, MVS_F32, &ws_bytes)) { fprintf(stderr, "%s\n", errstr()); return 4; }
    void* blob_h = malloc(blob_bytes);
    if (pack(convs, bns, prob_bias, 1e-5f, blob_h, blob_bytes)) { fprintf(stderr, "%s\n", errstr()); return 4; }

    float *d_feats, *d_proj, *d_dv, *d_depth, *d_conf;
    void *d_blob, *d_ws;
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_HIP(hipMalloc((void**)&d_feats, nf * 4));
    CHECK_HIP(hipMalloc((void**)&d_proj, (size_t)N * 64));
    CHECK_HIP(hipMalloc((void**)&d_dv, (size_t)D * 4));
    CHECK_HIP(hipMalloc((void**)&d_depth, (size_t)h * w * 4));
    CHECK_HIP(hipMalloc((void**)&d_conf, (size_t)h * w * 4));
    CHECK_HIP(hipMalloc(&d_blob, blob_bytes));
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMemcpy(d_feats, feats, nf * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_proj, proj, (size_t)N * 64, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_dv, dv, (size_t)D * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_blob, blob_h, blob_bytes, hipMemcpyHostToDevice));

    if (infer(d_feats, d_proj, d_dv, d_blob, d_depth, d_conf, d_ws, ws_bytes, N, 32, D, h, w, MVS_F32, stream)) {
        fprintf(stderr, "mvs_depth_infer: %s\n", errstr());
        return 4;
    }
    CHECK_HIP(hipStreamSynchronize(stream));
    float* out = (float*)malloc((size_t)2 * h * w * 4);
    CHECK_HIP(hipMemcpy(out, d_depth, (size_t)h * w * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + (size_t)h * w, d_conf, (size_t)h * w * 4, hipMemcpyDeviceToHost));
    f = fopen(argv[3], "wb");
    if (!f || fwrite(out, 4, (size_t)2 * h * w, f) != (size_t)2 * h * w) { perror(argv[3]); return 1; }
    fclose(f);
    /* a bad shape comes back as a status + message, never an abort */
    if (infer(d_feats, d_proj, d_dv, d_blob, d_depth, d_conf, d_ws, ws_bytes, N, 32, D + 1, h, w, MVS_F32, stream) != MVS_ERR_BAD_SHAPE) return 5;
    printf("ok N=%d D=%d h=%d w=%d depth[0]=%.4f (bad-shape message: %s)\n", N, D, h, w, out[0], errstr());
    return 0;
}
