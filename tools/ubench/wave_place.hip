// Where do the waves of a workgroup land?  768 workgroups x 4 waves with 50 KB of LDS each (the shape of the fused
// mixing backward): prints, from HW_ID, how many distinct SIMDs the 4 waves of a workgroup use and how many workgroups
// share a CU AT THE SAME TIME (from begin/end stamps of the 100 MHz s_memrealtime counter; LDS bytes from argv[1]).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, int spin, unsigned long long* tt) {
    extern __shared__ float sm[];
    const unsigned long long t0 = wall_clock64();
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;           // stay resident long enough to overlap
    sm[threadIdx.x] = x;
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { tt[blockIdx.x * 2] = t0; tt[blockIdx.x * 2 + 1] = t1; }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = id; out[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = xcc; }
}
int main(int argc, char** argv) {
    const int G = 768;
    const int lds = argc > 1 ? atoi(argv[1]) : 52000;
    unsigned* d; (void)hipMalloc(&d, G * 4 * 2 * 4);
    unsigned long long* tt; (void)hipMalloc(&tt, G * 2 * 8);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), lds, 0, d, 200000, tt);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> ht(G * 2);
    (void)hipMemcpy(ht.data(), tt, ht.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned> h(G * 8);
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<int, int> simd_hist;                 // distinct SIMDs per workgroup
    std::map<unsigned, std::set<int>> cu_wgs;     // (xcc, se, cu) -> workgroups
    for (int g = 0; g < G; ++g) {
        std::set<unsigned> simds;
        for (int w = 0; w < 4; ++w) {
            const unsigned id = h[(g * 4 + w) * 2], xcc = h[(g * 4 + w) * 2 + 1] & 0xf;
            const unsigned simd = (id >> 4) & 3, cu = (id >> 8) & 0xf, sh = (id >> 12) & 1, se = (id >> 13) & 7;
            simds.insert(simd);
            cu_wgs[(xcc << 16) | (se << 8) | (sh << 4) | cu].insert(g);
        }
        simd_hist[(int)simds.size()]++;
    }
    for (auto& p : simd_hist) printf("workgroups whose 4 waves sit on %d distinct SIMDs: %d\n", p.first, p.second);
    std::map<int, int> per_cu;
    for (auto& p : cu_wgs) per_cu[(int)p.second.size()]++;
    printf("distinct CUs seen: %zu\n", cu_wgs.size());
    for (auto& p : per_cu) printf("CUs that ran %d workgroups: %d\n", p.first, p.second);
    // how many of a CU's workgroups were resident together: the largest number of overlapping [begin, end] intervals
    std::map<int, int> conc;
    for (auto& p : cu_wgs) {
        int best = 0;
        for (int g : p.second) {
            int n = 0;
            for (int h2 : p.second) n += (ht[h2 * 2] <= ht[g * 2] && ht[g * 2] < ht[h2 * 2 + 1]) ? 1 : 0;
            best = n > best ? n : best;
        }
        conc[best]++;
    }
    printf("dynamic LDS %d bytes per workgroup\n", lds);
    for (auto& p : conc) printf("CUs with at most %d workgroups resident together: %d\n", p.first, p.second);
    return 0;
}
