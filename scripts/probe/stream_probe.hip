// Development probe: how fast can the workgroups of a launch stream an L2-resident table into LDS with
// global_load_lds_dwordx4 (1 KiB per wave instruction), nothing else going on?  The level kernels' bank stream
//  is priced against this.  Build: hipcc -O3 --offload-arch=gfx950 stream_probe.hip -o stream_probe
// Run: ./stream_probe <table KiB> <workgroups per CU> <copies in flight per wave> <passes> [1 = an LDS read + lgkmcnt(0) per copy]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14))

template <int AHEAD, bool LDSREAD>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ table, int nblk, int passes,
                                                     unsigned long long* clocks) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* ring = lds + wave * (AHEAD * 1024);
    const unsigned long long t0 = __builtin_readcyclecounter();
    int slot = 0;
    float sink = 0.f;
    // every workgroup walks the whole table, starting at a block of its own (as the nodes of a launch do)
    int blk = (blockIdx.x * 37 + wave) % nblk;
    const long total = (long)passes * (nblk / 4);
    for (long i = 0; i < total; ++i) {
        __builtin_amdgcn_global_load_lds((gptr_t)(table + (size_t)blk * 1024 + lane * 16), (lptr_t)(ring + slot * 1024), 16, 0, 0);
        WAIT_VM(AHEAD - 1);
        if (LDSREAD) {
            // one LDS read of a slot that landed long ago, and the full LDS-counter wait a consumer would need for it:
            // does that wait also wait for the copies in flight?
            const float v = *reinterpret_cast<volatile float*>(lds + 4 * AHEAD * 1024 + threadIdx.x * 4);
            __builtin_amdgcn_s_waitcnt((63 & 15) | (7 << 4) | (0 << 8) | ((63 >> 4) << 14));   // lgkmcnt(0), vmcnt untouched
            sink += v;
        }
        slot = slot == AHEAD - 1 ? 0 : slot + 1;
        blk += 4;
        if (blk >= nblk) blk -= nblk;
    }
    WAIT_VM(0);
    if (threadIdx.x == 0) clocks[blockIdx.x] = __builtin_readcyclecounter() - t0;
    if (sink == 12345.f) clocks[0] = 0;
}

template <int AHEAD, bool LDSREAD>
static void run(const char* d_table, int nblk, int wg_per_cu, int passes, unsigned long long* d_clk, int ncu) {
    const int grid = ncu * wg_per_cu;
    // LDS per workgroup sized so that exactly wg_per_cu fit a CU (160 KiB)
    size_t lds = (160 * 1024 / wg_per_cu) & ~(size_t)1023;
    if (lds < (size_t)4 * AHEAD * 1024 + 1024) lds = (size_t)4 * AHEAD * 1024 + 1024;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)stream_kernel<AHEAD, LDSREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream_kernel<AHEAD, LDSREAD>), dim3(grid), dim3(256), lds, 0, d_table, nblk, 1, d_clk);   // warm L2
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<AHEAD, LDSREAD>), dim3(grid), dim3(256), lds, 0, d_table, nblk, passes, d_clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> clk(grid);
    hipMemcpy(clk.data(), d_clk, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : clk) mean += (double)c;
    mean /= grid;
    const double bytes = (double)grid * passes * (nblk / 4) * 4 * 1024.0;
    // bytes per clock and CU from the workgroups' own clocks (wg_per_cu of them share a CU)
    const double per_wg = (double)passes * (nblk / 4) * 4 * 1024.0 / mean;
    printf("{\"table_KiB\": %d, \"wg_per_cu\": %d, \"in_flight_per_wave\": %d, \"ms\": %.3f, \"TB_per_s\": %.2f, "
           "\"lds_read_and_wait\": %d, \"B_per_clk_per_CU\": %.1f, \"clocks_per_wg\": %.0f, \"lds_per_wg\": %zu, \"err\": \"%s\"}\n",
           nblk, wg_per_cu, AHEAD, ms, bytes / (ms * 1e-3) / 1e12, (int)LDSREAD, per_wg * wg_per_cu, mean, lds, hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int kib = argc > 1 ? atoi(argv[1]) : 144;
    const int wg_per_cu = argc > 2 ? atoi(argv[2]) : 3;
    const int ahead = argc > 3 ? atoi(argv[3]) : 6;
    const int passes = argc > 4 ? atoi(argv[4]) : 50;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    char* d_table;
    unsigned long long* d_clk;
    hipMalloc(&d_table, (size_t)kib * 1024);
    hipMemset(d_table, 0, (size_t)kib * 1024);
    hipMalloc(&d_clk, (size_t)ncu * 8 * sizeof(unsigned long long));
    const bool rd = argc > 5 && atoi(argv[5]) != 0;
    if (ahead <= 3) { if (rd) run<3, true>(d_table, kib, wg_per_cu, passes, d_clk, ncu); else run<3, false>(d_table, kib, wg_per_cu, passes, d_clk, ncu); }
    else if (ahead <= 6) { if (rd) run<6, true>(d_table, kib, wg_per_cu, passes, d_clk, ncu); else run<6, false>(d_table, kib, wg_per_cu, passes, d_clk, ncu); }
    else { if (rd) run<12, true>(d_table, kib, wg_per_cu, passes, d_clk, ncu); else run<12, false>(d_table, kib, wg_per_cu, passes, d_clk, ncu); }
    hipFree(d_table); hipFree(d_clk);
    return 0;
}
