// How many workgroups of a given shape a compute unit holds at once (MI355X): every workgroup raises a counter of its
// compute unit on entry, notes the highest value seen, idles a while, and lowers the counter again.
//   hipcc --offload-arch=gfx950 -O2 -o tools/census/census tools/census/census.hip && tools/census/census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LDS_BYTES, int THREADS, int VGPRS>
__global__ __launch_bounds__(THREADS) void k_census(int* cur, int* peak, int spin)
{
    __shared__ uint32_t lds[LDS_BYTES / 4];
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint32_t const cu = ((xcc & 15) << 8) | ((hw >> 8) & 0xFF);      // xcc | se, sh, cu
    // the register allocation of the shape under test: touching the highest register is what sets it
    if (VGPRS == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (VGPRS == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if (VGPRS == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (threadIdx.x == 0) {
        int const now = atomicAdd(&cur[cu], 1) + 1;
        atomicMax(&peak[cu], now);
    }
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(64);
    __syncthreads();
    if (threadIdx.x == 0) { if (lds[1] == 77777u) peak[4095] = 1; atomicAdd(&cur[cu], -1); }
}

template <int LDS_BYTES, int THREADS, int VGPRS>
void run(const char* what)
{
    int *cur, *peak;
    hipMalloc(&cur, 4096 * 4); hipMalloc(&peak, 4096 * 4);
    hipMemset(cur, 0, 4096 * 4); hipMemset(peak, 0, 4096 * 4);
    hipLaunchKernelGGL((k_census<LDS_BYTES, THREADS, VGPRS>), dim3(4096), dim3(THREADS), 0, 0, cur, peak, 200);
    hipDeviceSynchronize();
    std::vector<int> h(4096);
    hipMemcpy(h.data(), peak, 4096 * 4, hipMemcpyDeviceToHost);
    int hist[40] = {0}, cus = 0;
    for (int i = 0; i < 4095; i++) if (h[i]) { hist[h[i] < 39 ? h[i] : 39]++; cus++; }
    printf("%-44s CUs seen %3d; peak workgroups per CU:", what, cus);
    for (int i = 1; i < 40; i++) if (hist[i]) printf(" %dx%d", hist[i], i);
    printf("\n");
    hipFree(cur); hipFree(peak);
}

int main()
{
    run<71888, 320, 128>("72 KB LDS, 320 threads, 128 VGPRs");
    run<71888, 320, 96>("72 KB LDS, 320 threads, 96 VGPRs");
    run<71888, 320, 64>("72 KB LDS, 320 threads, 64 VGPRs");
    run<71888, 512, 128>("72 KB LDS, 512 threads, 128 VGPRs");
    run<71888, 256, 128>("72 KB LDS, 256 threads, 128 VGPRs");
    run<65536, 320, 128>("64 KB LDS, 320 threads, 128 VGPRs");
    run<49152, 320, 128>("48 KB LDS, 320 threads, 128 VGPRs");
    run<49152, 320, 64>("48 KB LDS, 320 threads, 64 VGPRs");
    run<69632, 512, 128>("68 KB LDS, 512 threads, 128 VGPRs");
    run<10240, 64, 128>("10 KB LDS, 64 threads, 128 VGPRs");
    run<10240, 64, 64>("10 KB LDS, 64 threads, 64 VGPRs");
    return 0;
}
