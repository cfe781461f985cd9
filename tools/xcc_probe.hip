// tools/xcc_probe.hip -- which XCD runs workgroup b?  (level A of the direct path writes group g as 8 sub-streams and lets workgroup b
// append to stream b % 8, counting on the observed round-robin deal of workgroups over the 8 XCDs: the partial cache lines of one
// stream then meet in ONE L2.  This probe checks that deal on the box at hand, with workgroups shaped like level A's: 1024 threads,
// 64 KB of LDS, two per CU, many more workgroups than fit at once.)   hipcc --offload-arch=gfx950 -O2 -o xcc_probe tools/xcc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(1024) probe(unsigned* xcc, unsigned long long* t, unsigned spin)
{
    __shared__ unsigned pad[16000];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    pad[threadIdx.x] = id;
    __syncthreads();
    unsigned long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { xcc[blockIdx.x] = pad[5] & 0xF; t[blockIdx.x] = t0; }
}
int main()
{
    const unsigned nb = 32768;
    unsigned* dx; unsigned long long* dt;
    hipMalloc(&dx, nb * 4); hipMalloc(&dt, nb * 8);
    for (unsigned spin : {2000u, 20000u}) {
        hipLaunchKernelGGL(probe, dim3(nb), dim3(1024), 0, 0, dx, dt, spin);
        hipDeviceSynchronize();
        std::vector<unsigned> x(nb);
        hipMemcpy(x.data(), dx, nb * 4, hipMemcpyDeviceToHost);
        unsigned map[8][8] = {};
        for (unsigned b = 0; b < nb; ++b) map[b % 8][x[b] % 8]++;
        unsigned off = 0;
        for (unsigned r = 0; r < 8; ++r) { unsigned best = 0; for (unsigned c = 0; c < 8; ++c) best = map[r][c] > best ? map[r][c] : best; off += nb / 8 - best; }
        std::printf("spin %u: workgroups not on the XCD most of their residue class (b %% 8) is on: %u of %u\n", spin, off, nb);
        for (unsigned r = 0; r < 8; ++r) { std::printf("  b%%8=%u:", r); for (unsigned c = 0; c < 8; ++c) std::printf(" %5u", map[r][c]); std::printf("\n"); }
    }
    return 0;
}
