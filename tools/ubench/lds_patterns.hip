// tools/ubench/lds_patterns.hip -- LDS bank-conflict microbenchmark for the ME kernel's access patterns (gfx950).
// Sixteen waves on one CU (one 1024-thread workgroup): N back-to-back dependent-free ds_read2_b32 / ds_read_b32 / ds_read_b64 with a per-lane dword
// address table; cycles per instruction from s_memtime... (clock64).  Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_in/lds_patterns tools/ubench/lds_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND> // 0: ds_read_b32, 1: ds_read2_b32 (addr, addr + 1), 2: ds_read_b64, 3: ds_read2_b32 with second offset = +off1 dwords
__global__ void k(const int *addr, int off1, int iters, unsigned long long *out, unsigned *sink) {
    __shared__ unsigned lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const int a = addr[threadIdx.x & 63] + 32 * (threadIdx.x >> 6); /* 16 waves keep the CU's LDS pipe saturated: what is measured is its throughput */
    unsigned acc = 0;
    // warm
    acc += lds[a];
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int b = a + u * 0; // same address each time: the pattern is what is measured
            if (KIND == 0) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(b * 4)); acc += v; }
            if (KIND == 1) { unsigned long long v; asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(v) : "v"(b * 4)); acc += (unsigned)v + (unsigned)(v >> 32); }
            if (KIND == 2) { unsigned long long v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(b * 4)); acc += (unsigned)v + (unsigned)(v >> 32); }
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) *out = t2 - t0;
    sink[threadIdx.x & 63] = acc;
    (void)t1;
}

int main() {
    int *d_addr; unsigned long long *d_out; unsigned *d_sink;
    CHECK(hipMalloc(&d_addr, 64 * 4)); CHECK(hipMalloc(&d_out, 8)); CHECK(hipMalloc(&d_sink, 256));
    struct pat { const char *name; std::function<int(int)> f; };
    std::vector<pat> pats = {
        {"linear (lane)", [](int l) { return l; }},
        {"stride 2 (2*lane)", [](int l) { return 2 * l; }},
        {"HME now: 23*(l>>3) + 2*(l&7)", [](int l) { return 23 * (l >> 3) + 2 * (l & 7); }},
        {"HME rows {y,y+1,y+16,y+17} per half: ws 23", [](int l) { int run = l & 7, a = (l >> 3) & 1, b = (l >> 4) & 1, c = (l >> 5) & 1; return 23 * (a + 2 * c + 16 * b) + 2 * run; }},
        {"HME rows {y,y+16} x {y+1,y+17} per 16: ws 23", [](int l) { int run = l & 7, b = (l >> 3) & 1, a = (l >> 4) & 1, c = (l >> 5) & 1; return 23 * (a + 2 * c + 16 * b) + 2 * run; }},
        {"fullpel z-order 8x8 blocks, stride 25: (by*8)*25 + 2*bx", [](int l) { int bx = (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4), by = ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4); return by * 8 * 25 + 2 * bx; }},
        {"fullpel raster 8x8 blocks, stride 25", [](int l) { int bx = l & 7, by = l >> 3; return by * 8 * 25 + 2 * bx; }},
        {"fullpel z-order, stride 27", [](int l) { int bx = (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4), by = ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4); return by * 8 * 27 + 2 * bx; }},
        {"fullpel z-order, stride 29", [](int l) { int bx = (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4), by = ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4); return by * 8 * 29 + 2 * bx; }},
        {"fullpel z-order, stride 26", [](int l) { int bx = (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4), by = ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4); return by * 8 * 26 + 2 * bx; }},
        {"fullpel z-order, stride 18 (2 mod 16)", [](int l) { int bx = (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4), by = ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4); return by * 8 * 18 + 2 * bx; }},
        {"all same address", [](int l) { return 5; }},
        {"32-way: 32*lane", [](int l) { return 32 * l; }},
    };
    printf("%-62s %10s %10s %10s\n", "pattern (dword address per lane)", "b32", "read2_b32", "b64");
    for (auto &p : pats) {
        int h[64];
        for (int l = 0; l < 64; l++) h[l] = p.f(l);
        CHECK(hipMemcpy(d_addr, h, sizeof h, hipMemcpyHostToDevice));
        double cyc[3];
        for (int kind = 0; kind < 3; kind++) {
            const int iters = 2000;
            for (int rep = 0; rep < 2; rep++) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(1024), 0, 0, d_addr, 1, iters, d_out, d_sink);
                if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(1024), 0, 0, d_addr, 1, iters, d_out, d_sink);
                if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(1024), 0, 0, d_addr, 1, iters, d_out, d_sink);
                CHECK(hipDeviceSynchronize());
            }
            unsigned long long t;
            CHECK(hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost));
            cyc[kind] = (double)t / (iters * 16.0 * 16.0); /* LDS-pipe cycles per wave-instruction (16 waves x 16 instructions per iteration) */
        }
        printf("%-62s %10.2f %10.2f %10.2f\n", p.name, cyc[0], cyc[1], cyc[2]);
    }
    return 0;
}
