// tools/ubench/xcd_pingpong.hip -- what a producer -> consumer hand-over between two workgroups costs on MI355X, same XCD against different XCDs
// (workgroup id & 7 = XCD): A writes 64 dwords + a flag, B polls the flag, reads the dwords, answers with its own flag.  Variants of the data path:
//   0: write-through stores (sc1) + agent-scope loads (what svt_intra_kernel does)      1: plain stores + s_waitcnt, loads with glc (L2-coherent inside an XCD only)
// hipcc --offload-arch=gfx950 -O3 -o gpurun_in/xcd_pingpong tools/ubench/xcd_pingpong.hip && gpurun -- gpurun_in/xcd_pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ void pingpong(uint32_t *data, uint32_t *flags, int wg_a, int wg_b, int rounds, uint32_t *sink) {
    const int me = blockIdx.x == wg_a ? 0 : blockIdx.x == wg_b ? 1 : -1;
    if (me < 0) return;
    uint32_t *mine = data + me * 64, *theirs = data + (1 - me) * 64;
    uint32_t *fm = flags + me * 32, *ft = flags + (1 - me) * 32;
    uint32_t acc = 0;
    for (int r = 1; r <= rounds; r++) {
        if (me == 0) {
            if (MODE == 0) __hip_atomic_store(&mine[threadIdx.x], (uint32_t)r + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else mine[threadIdx.x] = (uint32_t)r + threadIdx.x;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(fm, (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x == 0) while (__hip_atomic_load(ft, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)r) __builtin_amdgcn_s_sleep(1);
            __syncthreads();
        } else {
            if (threadIdx.x == 0) while (__hip_atomic_load(ft, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)r) __builtin_amdgcn_s_sleep(1);
            __syncthreads();
            uint32_t v;
            if (MODE == 0) v = __hip_atomic_load(&theirs[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = __builtin_nontemporal_load(&theirs[threadIdx.x]);
            acc += v - ((uint32_t)r + threadIdx.x);   // 0 when the data was visible
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(fm, (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (me == 1) atomicAdd(sink, acc);
}
int main() {
    uint32_t *data, *flags, *sink;
    hipMalloc(&data, 4096); hipMalloc(&flags, 4096); hipMalloc(&sink, 4);
    const int rounds = 20000;
    for (int mode = 0; mode < 2; mode++)
        for (int pair = 0; pair < 3; pair++) {
            const int a = 0, b = pair == 0 ? 8 : pair == 1 ? 1 : 4;   // same XCD (0 and 8), neighbours (0, 1), across (0, 4)
            hipMemset(data, 0, 4096); hipMemset(flags, 0, 4096); hipMemset(sink, 0, 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(pingpong<0>, dim3(16), dim3(64), 0, 0, data, flags, a, b, rounds, sink);
            else hipLaunchKernelGGL(pingpong<1>, dim3(16), dim3(64), 0, 0, data, flags, a, b, rounds, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            uint32_t bad; hipMemcpy(&bad, sink, 4, hipMemcpyDeviceToHost);
            printf("mode %d (%s) workgroups %d and %d: %.2f us per round trip (two hand-overs); stale-data sum %u\n", mode, mode ? "plain stores, non-temporal loads" : "write-through stores, agent-scope loads",
                   a, b, 1000.0 * ms / rounds, bad);
        }
    return 0;
}
