// How fast are global atomics on ONE address (a work queue's counter) from 512 persistent workgroups?  hipcc --offload-arch=gfx950 -O3 -o tools/ubench_atomic tools/ubench_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int SCOPE, int NCTR>
__global__ void grab(unsigned *ctr, unsigned per_wg, unsigned *out)
{
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    unsigned *c = ctr + (NCTR == 1 ? 0 : (xcc & 7) * 64);
    unsigned last = 0;
    if (threadIdx.x == 0) {
        for (unsigned i = 0; i < per_wg; i++) {
            last += __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, SCOPE);       // dependent chain: latency-bound per workgroup, throughput-bound over 512 of them
            __builtin_amdgcn_s_sleep(8);
        }
        out[blockIdx.x] = last;
    }
}

template <int SCOPE, int NCTR>
static void run(const char *name, unsigned *d_ctr, unsigned *d_out)
{
    const unsigned G = 512, per = 400;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(d_ctr, 0, 8 * 64 * 4);
        hipEventRecord(a);
        hipLaunchKernelGGL((grab<SCOPE, NCTR>), dim3(G), dim3(64), 0, 0, d_ctr, per, d_out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned h[8 * 64]; hipMemcpy(h, d_ctr, sizeof h, hipMemcpyDeviceToHost);
        unsigned long long tot = 0; for (int i = 0; i < 8; i++) tot += h[i * 64];
        printf("%-44s %8.3f ms for %u grabs per workgroup x %u = %.1f ns per grab per workgroup, %.2f ns per grab overall; sum of counters %llu (want %u); per counter %u %u %u %u %u %u %u %u\n",
               name, ms, per, G, ms * 1e6 / per, ms * 1e6 / (per * G), tot, per * G, h[0], h[64], h[128], h[192], h[256], h[320], h[384], h[448]);
    }
}

int main()
{
    unsigned *d_ctr, *d_out;
    hipMalloc(&d_ctr, 8 * 64 * 4); hipMalloc(&d_out, 512 * 4);
    run<__HIP_MEMORY_SCOPE_AGENT, 1>("agent scope, one counter", d_ctr, d_out);
    run<__HIP_MEMORY_SCOPE_AGENT, 8>("agent scope, one counter per XCD", d_ctr, d_out);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, 8>("workgroup scope (at the L2), one per XCD", d_ctr, d_out);
    return 0;
}
