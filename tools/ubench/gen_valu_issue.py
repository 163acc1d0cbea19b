#!/usr/bin/env python3
"""Generates tools/ubench/valu_issue.hip: VALU issue rate on gfx950 with EXPLICIT registers (inline asm), so that operand
banks (register number mod 4), dependency distance and encoding (VOP2 / VOP3 / DPP) are what the test says they are."""
import os

def block(kind):
    """64 instructions on v[0..95]; v0..v31 accumulators, v32..v63 / v64..v95 operands."""
    L = []
    if kind == "fmac_nc":            # 16 independent chains, sources in three different banks
        for r in range(4):
            for i in range(16):
                d = i; a = 32 + ((i + 1) % 16) + 0; b = 64 + ((i + 2) % 16)
                # make banks distinct: d%4, a%4, b%4
                a = 32 + (i // 4) * 4 + (i + 1) % 4; b = 64 + (i // 4) * 4 + (i + 2) % 4
                L.append(f"v_fmac_f32 v{d}, v{a}, v{b}")
    elif kind == "fmac_ab":          # sources a, b in the same bank (dst in another)
        for r in range(4):
            for i in range(16):
                d = i; a = 32 + (i // 4) * 4 + (i + 1) % 4; b = 64 + (i // 4) * 4 + (i + 1) % 4
                L.append(f"v_fmac_f32 v{d}, v{a}, v{b}")
    elif kind == "fmac_all":         # all three in the same bank
        for r in range(4):
            for i in range(16):
                d = i; a = 32 + i; b = 64 + i
                L.append(f"v_fmac_f32 v{d}, v{a}, v{b}")
    elif kind == "chain2":           # the backward's pattern: two chains, each instruction depends on the one two before
        # v_fmac v[k-1], e[k], v[k]  for planes p = 0, 1 interleaved; 32 instructions per pass, two passes
        for r in range(2):
            for k in range(15, -1, -1):
                for p in range(2):
                    src = (16 * p + k + 1) if k < 15 else 31 - p * 0
                    if k == 15:
                        L.append(f"v_mul_f32 v{16 * p + 15}, v{32 + 15}, v{16 * p + 15}")
                    else:
                        L.append(f"v_fmac_f32 v{16 * p + k}, v{32 + k}, v{16 * p + k + 1}")
    elif kind == "chain4":           # four chains (J = 4)
        for k in range(15, -1, -1):
            for p in range(4):
                base = 8 * p
                kk = k % 8
                if kk == 7:
                    L.append(f"v_mul_f32 v{base + 7}, v{32 + k}, v{base + 7}")
                else:
                    L.append(f"v_fmac_f32 v{base + kk}, v{32 + k}, v{base + kk + 1}")
    elif kind == "fma3_neg":         # VOP3: 2*x - y (inline constant + neg modifier), 16 independent
        for r in range(4):
            for i in range(16):
                L.append(f"v_fma_f32 v{i}, v{32 + (i // 4) * 4 + (i + 1) % 4}, 2.0, -v{64 + (i // 4) * 4 + (i + 2) % 4}")
    elif kind == "fma3_vvv":         # VOP3, three VGPR sources, 16 independent
        for r in range(4):
            for i in range(16):
                L.append(f"v_fma_f32 v{i}, v{32 + (i // 4) * 4 + (i + 1) % 4}, v{64 + (i // 4) * 4 + (i + 2) % 4}, v{i}")
    elif kind == "sub":              # VOP2 v_sub, independent
        for r in range(4):
            for i in range(16):
                L.append(f"v_sub_f32 v{i}, v{32 + (i // 4) * 4 + (i + 1) % 4}, v{64 + (i // 4) * 4 + (i + 2) % 4}")
    elif kind == "fmac_dpp":         # the y sweep's Laplacian: fmac with a wave shift
        for r in range(4):
            for i in range(16):
                L.append(f"v_fmac_f32_dpp v{i}, v{32 + (i // 4) * 4 + (i + 1) % 4}, v{64 + (i // 4) * 4 + (i + 2) % 4} wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
    elif kind == "fmac_s":           # VOP2 with an SGPR as src0
        for r in range(4):
            for i in range(16):
                L.append(f"v_fmac_f32 v{i}, s{20 + i % 8}, v{64 + (i // 4) * 4 + (i + 2) % 4}")
    elif kind == "swap_ind":         # v_permlane32_swap, independent pairs
        for r in range(4):
            for i in range(16):
                L.append(f"v_permlane32_swap_b32 v{2 * i}, v{2 * i + 1}")
    elif kind == "swap_mix":         # one swap per 15 fmacs that depend on it (the sweeps' junction pattern)
        for r in range(4):
            L.append("s_nop 1")
            L.append("v_permlane32_swap_b32 v0, v1")
            for i in range(14):
                L.append(f"v_fmac_f32 v{(i + 1) % 2}, v{32 + i}, v{i % 2}")
    elif kind == "mov_dpp":          # v_mov_b32 with a wave shift
        for r in range(4):
            for i in range(16):
                L.append(f"v_mov_b32_dpp v{i}, v{32 + i} wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
    elif kind == "fmac_rowshr":      # fmac with a row shift (stays inside 16 lanes)
        for r in range(4):
            for i in range(16):
                L.append(f"v_fmac_f32_dpp v{i}, v{32 + (i // 4) * 4 + (i + 1) % 4}, v{64 + (i // 4) * 4 + (i + 2) % 4} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
    elif kind == "cndmask":          # v_cndmask with an SGPR-pair condition (VOP3)
        for r in range(4):
            for i in range(16):
                L.append(f"v_cndmask_b32_e64 v{i}, v{32 + i}, v{64 + i}, s[20:21]")
    elif kind == "pk_fma":           # packed: 32 instructions = 64 lanes-ops
        for r in range(4):
            for i in range(8):
                L.append(f"v_pk_fma_f32 v[{2 * i}:{2 * i + 1}], v[{32 + 2 * i}:{33 + 2 * i}], v[{64 + 2 * i}:{65 + 2 * i}], v[{2 * i}:{2 * i + 1}]")
    return L

KINDS = ["fmac_nc", "chain2", "fma3_neg", "fmac_dpp", "fmac_rowshr", "mov_dpp", "fmac_s", "cndmask", "swap_ind", "swap_mix", "pk_fma"]
out = ['// GENERATED by gen_valu_issue.py - do not edit.  See that file.',
       '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>', '#include <algorithm>', '']
clob = ", ".join(f'"v{i}"' for i in range(96)) + ', "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"'
for kd in KINDS:
    body = block(kd)
    init = [f"v_mov_b32 v{i}, 0x3f800000" for i in range(32)] + [f"v_mov_b32 v{i}, 0x3f000000" for i in range(32, 96)] + \
           [f"s_mov_b32 s{20 + i}, 0x3f000000" for i in range(8)]
    out.append(f"__global__ void k_{kd}(unsigned long long* cyc, float* outp, int reps) {{")
    out.append("    extern __shared__ float pad[];")
    out.append('    asm volatile("' + "\\n\\t".join(init) + f'" ::: {clob});')
    out.append("    __syncthreads();")
    out.append("    const unsigned long long t0 = __builtin_amdgcn_s_memtime();")
    out.append("    for (int it = 0; it < reps; ++it) {")
    out.append('        asm volatile("' + "\\n\\t".join(body) + f'" ::: {clob});')
    out.append("    }")
    out.append("    const unsigned long long t1 = __builtin_amdgcn_s_memtime();")
    out.append("    float r; asm volatile(\"v_mov_b32 %0, v0\" : \"=v\"(r));")
    out.append("    outp[blockIdx.x * blockDim.x + threadIdx.x] = r;")
    out.append("    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;")
    out.append("    if (threadIdx.x == 99999) pad[0] = r;")
    out.append("}")
    out.append(f"static const int n_{kd} = {len(body)};")
out.append('''
typedef void (*kern_t)(unsigned long long*, float*, int);
static void run(const char* name, kern_t k, int ninst, int W, unsigned long long* cyc, float* outp) {
    const int reps = 4096;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(256 * W), 100 * 1024, 0, cyc, outp, reps);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(256), dim3(256 * W), 100 * 1024, 0, cyc, outp, reps);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 4 * W);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double n = (double)reps * ninst;
    printf("%-9s W=%d: %6.2f cyc/instr/wave, %5.2f SIMD-cyc per wave-instr; wall %.3f ms -> clock %.2f GHz, %.2f ns of SIMD per wave-instr\\n",
           name, W, med / n, med / (n * W), ms, med / (ms * 1e6), ms * 1e6 / (n * W));
}
int main() {
    unsigned long long* cyc; float* outp;
    hipMalloc(&cyc, 256 * 16 * 8); hipMalloc(&outp, 256 * 1024 * 4);
    for (int W = 1; W <= 4; ++W) {''')
for kd in KINDS:
    out.append(f'        run("{kd}", k_{kd}, n_{kd}, W, cyc, outp);')
out.append("    }\n    return 0;\n}")
open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_issue.hip"), "w").write("\n".join(out) + "\n")
