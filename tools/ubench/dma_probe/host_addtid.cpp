// Diagnostic host for gen_addtid.py: checks the transposition row layout -> column layout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
int main() {
    hipModule_t m; hipFunction_t f;
    if (hipModuleLoad(&m, "addtid.hsaco") != hipSuccess || hipModuleGetFunction(&f, m, "addtid") != hipSuccess) { printf("load failed\n"); return 2; }
    std::vector<float> in(64 * 16), out(64 * 16);
    // row layout: lane (hf, l) holds row h = l, element k <-> column w = hf ? 31 - k : k ; value = 100*h + w
    for (int lane = 0; lane < 64; ++lane) for (int k = 0; k < 16; ++k) { int hf = lane >> 5, l = lane & 31, w = hf ? 31 - k : k; in[lane * 16 + k] = 100.f * l + w; }
    float *src, *dst; hipMalloc(&src, 4096); hipMalloc(&dst, 4096);
    hipMemcpy(src, in.data(), 4096, hipMemcpyHostToDevice); hipMemset(dst, 0, 4096);
    struct { void* s; void* d; } args{src, dst}; size_t size = sizeof(args);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    if (hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, 0, nullptr, extra) != hipSuccess) { printf("launch failed\n"); return 3; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\n"); return 4; }
    hipMemcpy(out.data(), dst, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    // column layout: lane (hf', l') holds column w = l', element k' <-> row h = hf' ? 31 - k' : k'
    for (int lane = 0; lane < 64; ++lane) for (int k = 0; k < 16; ++k) {
        int hf = lane >> 5, l = lane & 31, h = hf ? 31 - k : k; float want = 100.f * h + l;
        if (out[lane * 16 + k] != want) { if (bad < 8) printf("lane %d k %d: got %g want %g\n", lane, k, out[lane * 16 + k], want); ++bad; }
    }
    printf("addtid transpose: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return bad ? 1 : 0;
}
