#!/usr/bin/env python3
"""Generates tools/ubench/issue_model.hip: how fast can ONE wave per SIMD issue the instruction mix of a Fisher-vector-product tile
(78 v_mfma_f32_16x16x4_f32 in the real accumulator-chain structure) with NV independent VALU instructions, ND LDS and NG buffer-load
instructions dealt evenly into the MFMA gaps?  The kernel computes nothing meaningful; only the clock is read.
Variants are compiled into one binary: issue_model<NV_PER_GAP x 10, with_lds, with_vmem>."""
import sys

def body(nv10, lds, vmem):
    """one tile: 78 MFMAs; per gap nv10/10 VALU on average (groups of 4 independent chains), 20 DS, 10 VMEM spread"""
    L = []
    # accumulators: t0 v[0:3],v[4:7]; t1 v[8:11],v[12:15]; d0n v[16:19],v[20:23]; gW1 a[0:15]; gW0 a[16:23]
    # operands: A from a[32..95] (weights), B from v[24..55] (activations); VALU scratch v[64..127]
    chains = []
    chains += [('v[%d:%d]' % (4 * (g % 2), 4 * (g % 2) + 3), False) for g in range(6)]           # S1
    chains += [('v[%d:%d]' % (8 + 4 * (g % 2), 8 + 4 * (g % 2) + 3), False) for g in range(32)]  # S2 + S3
    chains += [('v[%d:%d]' % (16 + 4 * (g % 2), 16 + 4 * (g % 2) + 3), False) for g in range(16)]  # S6
    chains += [('a[%d:%d]' % (4 * (g % 4), 4 * (g % 4) + 3), True) for g in range(16)]           # gW1
    chains += [('a[%d:%d]' % (16 + 4 * (g % 2), 16 + 4 * (g % 2) + 3), True) for g in range(8)]  # gW0
    assert len(chains) == 78
    nv_total = nv10 * 78 // 10
    vdone = ddone = gdone = 0
    for g, (acc, _) in enumerate(chains):
        L.append('v_mfma_f32_16x16x4_f32 %s, a%d, v%d, %s' % (acc, 32 + g % 54, 24 + g % 32, acc))
        want_v = nv_total * (g + 1) // 78
        while vdone < want_v:
            r = 64 + (vdone % 4) + 4 * ((vdone // 4) % 8)       # four independent chains at a time, 12 groups
            if (vdone // 4) % 3 == 0: L.append('v_fma_f32 v%d, -v%d, v%d, 1.0' % (r, 24 + vdone % 32, 24 + vdone % 32))
            elif (vdone // 4) % 3 == 1: L.append('v_mul_f32 v%d, v%d, v%d' % (r, r, 96 + vdone % 4))
            else: L.append('v_fmac_f32 v%d, v%d, v%d' % (104 + vdone % 4, r - 4 if r >= 68 else r, 100 + vdone % 4))
            vdone += 1
        if lds:
            want_d = 20 * (g + 1) // 78
            while ddone < want_d:
                if ddone < 16: L.append('ds_write_b32 v56, v%d offset:%d' % (64 + ddone, 80 * ddone))
                else: L.append('ds_read_b128 v[%d:%d], v57 offset:%d' % (108 + 4 * ((ddone - 16) % 2), 111 + 4 * ((ddone - 16) % 2), 1280 * (ddone - 16)))
                ddone += 1
        if vmem:
            want_g = 10 * (g + 1) // 78 if g < 40 else 10
            while gdone < want_g:
                if gdone < 4: L.append('buffer_load_dwordx4 v[%d:%d], v58, s[8:11], 0 offen offset:%d' % (108 + 4 * (gdone % 2), 111 + 4 * (gdone % 2), 1024 * gdone))
                else: L.append('buffer_load_dword v%d, v59, s[12:15], 0 offen offset:%d' % (116 + gdone % 4, 64 * gdone))
                gdone += 1
    if lds or vmem: L.append('s_waitcnt vmcnt(0) lgkmcnt(0)')
    return L

def body_clustered(K, nv, lds, vmem, rot, wreads=0, avgpr=True, waits=False):
    """K phases per tile, each = a run of 78/K MFMAs then a run of nv/K VALU (+ the phase's share of DS / VMEM); rot: start with the VALU run"""
    chains = []
    chains += ['v[%d:%d]' % (4 * (g % 2), 4 * (g % 2) + 3) for g in range(6)]
    chains += ['v[%d:%d]' % (8 + 4 * (g % 2), 8 + 4 * (g % 2) + 3) for g in range(32)]
    chains += ['v[%d:%d]' % (16 + 4 * (g % 2), 16 + 4 * (g % 2) + 3) for g in range(16)]
    chains += ['a[%d:%d]' % (4 * (g % 4), 4 * (g % 4) + 3) for g in range(16)]
    chains += ['a[%d:%d]' % (16 + 4 * (g % 2), 16 + 4 * (g % 2) + 3) for g in range(8)]
    L = []
    g = vdone = ddone = gdone = 0
    for k in range(K):
        M = []
        while g < 78 * (k + 1) // K:
            if wreads and g % 3 == 0 and g // 3 < wreads: M.append('ds_read_b64 v[%d:%d], v57 offset:%d' % (24 + 2 * ((g // 3) % 16), 25 + 2 * ((g // 3) % 16), 512 * (g // 3)))
            if wreads and waits and g % 4 == 2 and g // 3 < wreads: M.append('s_waitcnt lgkmcnt(0)')
            M.append(('v_mfma_f32_16x16x4_f32 %s, a%d, v%d, %s' % (chains[g], 32 + g % 54, 24 + g % 32, chains[g])) if avgpr else
                     ('v_mfma_f32_16x16x4_f32 %s, v%d, v%d, %s' % (chains[g], 24 + (g + 7) % 32, 24 + g % 32, chains[g]))); g += 1
        V = []
        while vdone < nv * (k + 1) // K:
            r = 64 + (vdone % 4) + 4 * ((vdone // 4) % 8)
            if (vdone // 4) % 3 == 0: V.append('v_fma_f32 v%d, -v%d, v%d, 1.0' % (r, 24 + vdone % 32, 24 + vdone % 32))
            elif (vdone // 4) % 3 == 1: V.append('v_mul_f32 v%d, v%d, v%d' % (r, r, 96 + vdone % 4))
            else: V.append('v_fmac_f32 v%d, v%d, v%d' % (104 + vdone % 4, r - 4 if r >= 68 else r, 100 + vdone % 4))
            vdone += 1
        if lds:
            while ddone < 20 * (k + 1) // K:
                if ddone < 16: V.append('ds_write_b32 v56, v%d offset:%d' % (64 + ddone, 80 * ddone))
                else: V.append('ds_read_b128 v[%d:%d], v57 offset:%d' % (108 + 4 * ((ddone - 16) % 2), 111 + 4 * ((ddone - 16) % 2), 1280 * (ddone - 16)))
                ddone += 1
        if vmem:
            while gdone < 10 * (k + 1) // K:
                if gdone < 4: V.append('buffer_load_dwordx4 v[%d:%d], v58, s[8:11], 0 offen offset:%d' % (108 + 4 * (gdone % 2), 111 + 4 * (gdone % 2), 1024 * gdone))
                else: V.append('buffer_load_dword v%d, v59, s[12:15], 0 offen offset:%d' % (116 + gdone % 4, 64 * gdone))
                gdone += 1
        L += (V + M) if rot else (M + V)
    if lds or vmem: L.append('s_waitcnt vmcnt(0) lgkmcnt(0)')
    return L

out = ['// GENERATED by gen_issue_model.py -- do not edit', '#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>', '#include <vector>',
       'typedef float f32x4 __attribute__((ext_vector_type(4)));']
clob = ', '.join('"v%d"' % i for i in range(0, 120)) + ', ' + ', '.join('"a%d"' % i for i in range(0, 96))
kernels = []          # (name, label, threads, body lines for the first-dispatched half, body lines for waves 4-7 or None)
for nv10, lds, vmem in [(0, 0, 0), (20, 0, 0), (30, 1, 1)]:
    kernels.append(('k_il_%d_%d_%d' % (nv10, lds, vmem), 'interleaved %.1f VALU/gap lds %d vmem %d, 1 wave/SIMD' % (nv10 / 10.0, lds, vmem), 256, body(nv10, lds, vmem), None))
kernels.append(('k_il2', 'interleaved 3.0 VALU/gap lds vmem, 2 waves/SIMD', 512, body(30, 1, 1), body(30, 1, 1)))
kernels.append(('k_cl2_w', 'clustered K=1, 2 waves/SIMD, 26 ds_read_b64 inside the MFMA run', 512, body_clustered(1, 230, 1, 1, False, 26), body_clustered(1, 230, 1, 1, False, 26)))
kernels.append(('k_cl2_w3', 'clustered K=3, 2 waves/SIMD, 26 ds_read_b64 inside the MFMA runs', 512, body_clustered(3, 230, 1, 1, False, 26), body_clustered(3, 230, 1, 1, False, 26)))
kernels.append(('k_cl2_v300', 'clustered K=1, 2 waves/SIMD, 300 VALU', 512, body_clustered(1, 300, 1, 1, False), body_clustered(1, 300, 1, 1, False)))
kernels.append(('k_cl2_v300w', 'clustered K=1, 2 waves/SIMD, 300 VALU, 26 ds_read_b64 in the MFMA run', 512, body_clustered(1, 300, 1, 1, False, 26), body_clustered(1, 300, 1, 1, False, 26)))
kernels.append(('k_cl2_va', 'clustered K=2, 2 waves/SIMD, A operands in VGPRs, 26 reads in runs', 512, body_clustered(2, 230, 1, 1, False, 26, False), body_clustered(2, 230, 1, 1, False, 26, False)))
kernels.append(('k_cl2_vaw', 'clustered K=2, 2 waves/SIMD, A in VGPRs, reads in runs + lgkmcnt(0) waits', 512, body_clustered(2, 230, 1, 1, False, 26, False, True), body_clustered(2, 230, 1, 1, False, 26, False, True)))
for K in (1, 3):
    kernels.append(('k_cl_%d' % K, 'clustered K=%d (230 VALU, lds, vmem), 1 wave/SIMD' % K, 256, body_clustered(K, 230, 1, 1, False), None))
    kernels.append(('k_cl2_%d' % K, 'clustered K=%d, 2 waves/SIMD same phase' % K, 512, body_clustered(K, 230, 1, 1, False), body_clustered(K, 230, 1, 1, False)))
    kernels.append(('k_cl2r_%d' % K, 'clustered K=%d, 2 waves/SIMD ANTI-phase' % K, 512, body_clustered(K, 230, 1, 1, False), body_clustered(K, 230, 1, 1, True)))
def emit_asm(lines):
    o = ['        asm volatile("s_mov_b32 s8, %0\\n s_mov_b32 s9, %1\\n s_mov_b32 s10, %2\\n s_mov_b32 s11, %3\\n s_mov_b32 s12, %4\\n s_mov_b32 s13, %5\\n s_mov_b32 s14, %6\\n s_mov_b32 s15, %7\\n v_mov_b32 v200, %8\\n v_mov_b32 v201, %8\\n v_mov_b32 v202, %9\\n v_mov_b32 v203, %9\\n"']
    o += ['            "%s\\n"' % l for l in lines]
    o.append('            :: "s"(d1[0]), "s"(d1[1]), "s"(d1[2]), "s"(d1[3]), "s"(d2[0]), "s"(d2[1]), "s"(d2[2]), "s"(d2[3]), "v"(la), "v"(off + it * 4096)')
    o.append('            : "memory", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", %s);' % clob)
    return o
for name, label, threads, b0, b1 in kernels:
    out.append('__global__ void __launch_bounds__(%d, 1) %s(const float* src, float* dst, int iters, unsigned long long* clk) {' % (threads, name))
    out.append('    __shared__ float lds[16384];')
    out.append('    unsigned d1[4], d2[4];')
    out.append('    { const unsigned long long b1 = (unsigned long long)(src + (size_t)blockIdx.x * 65536), b2 = (unsigned long long)(src + (size_t)blockIdx.x * 4096 + 8);')
    out.append('      d1[0] = __builtin_amdgcn_readfirstlane((unsigned)b1); d1[1] = __builtin_amdgcn_readfirstlane((unsigned)(b1 >> 32) & 0xffff); d1[2] = 1u << 20; d1[3] = 0x00020000;')
    out.append('      d2[0] = __builtin_amdgcn_readfirstlane((unsigned)b2); d2[1] = __builtin_amdgcn_readfirstlane((unsigned)(b2 >> 32) & 0xffff); d2[2] = 1u << 16; d2[3] = 0x00020000; }')
    out.append('    const unsigned la = (unsigned)(size_t)lds + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 8192, off = (threadIdx.x & 255) * 16;')
    out.append('    const int second = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));')
    out.append('    const unsigned long long t0 = __builtin_readcyclecounter();')
    out.append('    if (!second) { for (int it = 0; it < iters; ++it) {')
    out += emit_asm(b0)
    out.append('    } }')
    if b1 is not None:
        out.append('    else { for (int it = 0; it < iters; ++it) {')
        out += emit_asm(b1)
        out.append('    } }')
    out.append('    const unsigned long long t1 = __builtin_readcyclecounter();')
    out.append('    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;')
    out.append('    if (iters < 0) dst[threadIdx.x] = lds[threadIdx.x];')
    out.append('}')
out.append('int main() {')
out.append('    float *src, *dst; unsigned long long* clk; (void)hipMalloc(&src, 256ull * 65536 * 4 + (1 << 22)); (void)hipMalloc(&dst, 4096); (void)hipMalloc(&clk, 8);')
out.append('    (void)hipMemset(src, 0, 256ull * 65536 * 4 + (1 << 22));')
out.append('    const int iters = 200; unsigned long long h = 0;')
for name, label, threads, b0, b1 in kernels:
    out.append('    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(%s, dim3(256), dim3(%d), 0, 0, src, dst, iters, clk); (void)hipDeviceSynchronize(); }' % (name, threads))
    per = 2 if threads == 512 else 1
    out.append('    (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost); printf("%%-62s : %%6.0f cycles per tile-iteration of a wave = %%6.0f per tile per SIMD\\n", "%s", (double)h / iters, (double)h / iters / %d);' % (label, per))
out.append('    return 0;')
out.append('}')
open(sys.argv[1] if len(sys.argv) > 1 else 'issue_model.hip', 'w').write('\n'.join(out) + '\n')
