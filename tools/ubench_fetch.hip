// ubench_fetch.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the transform kernels use
// (MI355X_MICROARCH.md, HBM: only the wide streaming read is calibrated -- "calibrate on a known byte count in your own access
// pattern").  Every kernel reads (and one writes) a 1 GiB plane exactly once; compare the counter with 2^30.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_fetch ubench_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- ./ubench_fetch      (and again with --pmc WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define STRIDE 4096
#define ROWS 262144
__device__ __forceinline__ int walk_group(int ngroups, int j) {   /* same order as tq_walk_of */
    const int per = (ngroups + 7) >> 3;
    return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) + j * (int)(gridDim.x >> 3);
}
__global__ void stream16(const uint4 *p, size_t n, uint32_t *sink) {
    uint32_t a = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (a == 0x12345678u) *sink = a;
}
/* one 4x4 block per lane: 4 dword loads, rows STRIDE apart.  ZORDER: consecutive blocks follow the z-order inside 64x64 SBs. */
template <bool ZORDER, bool WRITE> __global__ void lane4x4(uint8_t *p, uint32_t *sink) {
    const int nblk = (STRIDE / 4) * (ROWS / 4), ngroups = nblk / 256, per = (ngroups + 7) >> 3;
    uint32_t a = 0;
    for (int j = 0;; j++) {
        const int g = (int)(blockIdx.x >> 3) + j * (int)(gridDim.x >> 3);
        if (g >= per) break;
        const int blk = ((int)(blockIdx.x & 7) * per + g) * 256 + threadIdx.x;
        if (blk >= nblk) continue;
        int bx, by;
        if (ZORDER) {
            const int sb = blk >> 8, z = blk & 255;  /* 256 4x4 blocks per 64x64 SB */
            int zx = 0, zy = 0;
            for (int b = 0; b < 4; b++) { zx |= ((z >> (2 * b)) & 1) << b; zy |= ((z >> (2 * b + 1)) & 1) << b; }
            bx = (sb % (STRIDE / 64)) * 16 + zx; by = (sb / (STRIDE / 64)) * 16 + zy;
        } else { bx = blk % (STRIDE / 4); by = blk / (STRIDE / 4); }
        uint8_t *q = p + (size_t)by * 4 * STRIDE + bx * 4;
        for (int r = 0; r < 4; r++) {
            if (WRITE) *(uint32_t *)(q + (size_t)r * STRIDE) = (uint32_t)blk + r;
            else a ^= *(const uint32_t *)(q + (size_t)r * STRIDE);
        }
    }
    if (a == 0x12345678u) *sink = a;
}
/* N lanes per NxN block, lane i = row i of N bytes (8 -> dwordx2, 16 -> dwordx4); blocks raster inside 64x64 SBs */
template <int N> __global__ void rowsN(const uint8_t *p, uint32_t *sink) {
    constexpr int BPW = 256 / N, PER_SB = (64 / N) * (64 / N);
    const int nblk = (STRIDE / N) * (ROWS / N), ngroups = nblk / BPW, per = (ngroups + 7) >> 3;
    uint32_t a = 0;
    for (int j = 0;; j++) {
        const int g = (int)(blockIdx.x >> 3) + j * (int)(gridDim.x >> 3);
        if (g >= per) break;
        const int blk = ((int)(blockIdx.x & 7) * per + g) * BPW + threadIdx.x / N, i = threadIdx.x % N;
        if (blk >= nblk) continue;
        const int sb = blk / PER_SB, z = blk % PER_SB;
        const int bx = (sb % (STRIDE / 64)) * (64 / N) + z % (64 / N), by = (sb / (STRIDE / 64)) * (64 / N) + z / (64 / N);
        const uint8_t *q = p + ((size_t)by * N + i) * STRIDE + bx * N;
        if (N == 8) { uint2 v = *(const uint2 *)q; a ^= v.x ^ v.y; }
        else { for (int k = 0; k < N / 16; k++) { uint4 v = ((const uint4 *)q)[k]; a ^= v.x ^ v.y ^ v.z ^ v.w; } }
    }
    if (a == 0x12345678u) *sink = a;
}
int main() {
    uint8_t *p; uint32_t *sink;
    const size_t bytes = (size_t)STRIDE * ROWS;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(p, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 8;
    for (int rep = 0; rep < 2; rep++) {
        stream16<<<grid, 256>>>((const uint4 *)p, bytes / 16, sink);
        lane4x4<false, false><<<grid, 256>>>(p, sink);
        lane4x4<true, false><<<grid, 256>>>(p, sink);
        rowsN<8><<<grid, 256>>>(p, sink);
        rowsN<16><<<grid, 256>>>(p, sink);
        rowsN<32><<<grid, 256>>>(p, sink);
        lane4x4<true, true><<<grid, 256>>>(p, sink);
        lane4x4<false, true><<<grid, 256>>>(p, sink);
    }
    hipDeviceSynchronize();
    printf("plane bytes %zu\n", bytes);
    return 0;
}
