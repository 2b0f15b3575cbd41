// Micro-benchmarks of the primitives the ME kernel is built from (cycles per operation per wave, s_memtime).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_me tools/ubench_me.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 256
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
__global__ void k(unsigned long long *out, const uint32_t *g, int nwaves_note) {
    __shared__ uint32_t lds[8192];
    __shared__ unsigned long long key;
    int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = i * 2654435761u;
    if (tid == 0) key = ~0ull;
    __syncthreads();
    unsigned long long t0, t1;
    uint64_t acc = tid; uint32_t a = tid, b = tid * 3 + 1;
    // 1. dependent qsad chain
    t0 = now();
    for (int i = 0; i < N; i++) acc = __builtin_amdgcn_qsad_pk_u16_u8(acc ^ 0x0101010101010101ull, b, acc);
    t1 = now(); if (tid == 0) out[0] = t1 - t0;
    // 2. independent qsads (4 chains)
    uint64_t c0 = acc, c1 = acc + 1, c2 = acc + 2, c3 = acc + 3;
    t0 = now();
    for (int i = 0; i < N / 4; i++) {
        c0 = __builtin_amdgcn_qsad_pk_u16_u8(c0, b, c0); c1 = __builtin_amdgcn_qsad_pk_u16_u8(c1, b, c1);
        c2 = __builtin_amdgcn_qsad_pk_u16_u8(c2, b, c2); c3 = __builtin_amdgcn_qsad_pk_u16_u8(c3, b, c3);
    }
    t1 = now(); if (tid == 0) out[1] = t1 - t0;
    acc ^= c0 ^ c1 ^ c2 ^ c3;
    // 3. dependent v_sad_u8 chain
    t0 = now();
    for (int i = 0; i < N; i++) a = __builtin_amdgcn_sad_u8(a, b, a);
    t1 = now(); if (tid == 0) out[2] = t1 - t0;
    // 4. dependent LDS read chain (pointer chase)
    uint32_t p = tid;
    t0 = now();
    for (int i = 0; i < N; i++) p = lds[p & 8191];
    t1 = now(); if (tid == 0) out[3] = t1 - t0;
    a ^= p;
    // 5. independent LDS reads (8 per iteration, consecutive lanes consecutive dwords)
    uint32_t s = 0;
    t0 = now();
    for (int i = 0; i < N / 8; i++) {
        _Pragma("unroll") for (int u = 0; u < 8; u++) s += lds[(tid + 64 * u + 7 * i) & 8191];
    }
    t1 = now(); if (tid == 0) out[4] = t1 - t0;
    a ^= s;
    // 6. shuffle chain (ds_bpermute)
    t0 = now();
    for (int i = 0; i < N; i++) a += __shfl_xor(a, 1 + (i & 31));
    t1 = now(); if (tid == 0) out[5] = t1 - t0;
    // 7. barriers
    t0 = now();
    for (int i = 0; i < N; i++) __syncthreads();
    t1 = now(); if (tid == 0) out[6] = t1 - t0;
    // 8. same-address LDS 64-bit atomic min, all lanes
    t0 = now();
    for (int i = 0; i < 64; i++) atomicMin(&key, ((unsigned long long)a << 32) | (unsigned)(tid + i));
    t1 = now(); if (tid == 0) out[7] = t1 - t0;
    // 9. dependent global load chain (L2 / HBM)
    uint32_t q = (tid * 64 + blockIdx.x * 977) & ((1 << 22) - 1);
    t0 = now();
    for (int i = 0; i < 64; i++) q = g[q] & ((1 << 22) - 1);
    t1 = now(); if (tid == 0) out[8] = t1 - t0;
    // 10. v_readlane/v_writelane-free plain VALU chain for reference
    t0 = now();
    for (int i = 0; i < N; i++) a = a * 3 + b;
    t1 = now(); if (tid == 0) out[9] = t1 - t0;
    // 11. v_alignbyte chain
    t0 = now();
    for (int i = 0; i < N; i++) a = __builtin_amdgcn_alignbyte(a, b, a & 3);
    t1 = now(); if (tid == 0) out[10] = t1 - t0;
    if (a == 0x12345 && acc == 77 && q == 3) out[15] = key;
}
int main() {
    unsigned long long *d, h[16];
    uint32_t *g; size_t n = 1 << 22;
    hipMalloc(&d, sizeof h); hipMalloc(&g, n * 4);
    uint32_t *hg = (uint32_t *)malloc(n * 4);
    uint32_t x = 12345; for (size_t i = 0; i < n; i++) { x = x * 1664525u + 1013904223u; hg[i] = x >> 8; }
    hipMemcpy(g, hg, n * 4, hipMemcpyHostToDevice);
    const char *nm[11] = {"qsad dependent", "qsad 4 indep chains", "sad_u8 dependent", "lds pointer chase", "lds indep reads", "shfl_xor chain",
                          "__syncthreads (256 thr)", "lds atomicMin u64 same addr", "global load chase (16MB)", "valu mad chain", "alignbyte chain"};
    const int per[11] = {N, N, N, N, N, N, N, 64, 64, N, N};
    for (int blocks = 1; blocks <= 1; blocks++) {
        for (int threads = 64; threads <= 256; threads *= 4) {
            hipMemset(d, 0, sizeof h);
            k<<<1, threads>>>(d, g, 0);
            hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("== 1 block x %d threads\n", threads);
            for (int i = 0; i < 11; i++) printf("  %-32s %8.1f cycles/op\n", nm[i], (double)h[i] / per[i]);
        }
    }
    // loaded machine: 768 blocks x 256 threads (3 per CU)
    hipMemset(d, 0, sizeof h);
    k<<<768, 256>>>(d, g, 0);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("== 768 blocks x 256 threads (block 0's thread 0... last writer)\n");
    for (int i = 0; i < 11; i++) printf("  %-32s %8.1f cycles/op\n", nm[i], (double)h[i] / per[i]);
    return 0;
}
