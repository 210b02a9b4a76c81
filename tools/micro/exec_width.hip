// Microbenchmark: does a wave64 fp64 FMA with only 16 (or 32) active lanes issue faster than with all 64?
// (Decides whether running B = 4096 trajectories as 256 quarter-filled waves could beat 64 full ones.)
// Build: hipcc --offload-arch=gfx950 -O3 -o exec_width exec_width.hip ; run: ./exec_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int ILP>
__global__ void k(double* out, uint64_t* clk, int active, int n) {
    const int lane = threadIdx.x;
    if (lane >= active) return;
    double a[ILP];
    for (int i = 0; i < ILP; ++i) a[i] = 1.0 + 1e-9 * (lane + i);
    const double b = 1.0000001, c = 1e-7;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) a[i] = __builtin_fma(a[i], b, c);
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < ILP; ++i) s += a[i];
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
}

template <int ILP>
void run(int active) {
    double* out; uint64_t* clk;
    hipMalloc(&out, 64 * 8 * 1024); hipMalloc(&clk, 8 * 1024);
    const int n = 2000;
    for (int blocks : {1, 1024}) {
        hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(64), 0, 0, out, clk, active, n);
        hipDeviceSynchronize();
        uint64_t h[1024];
        hipMemcpy(h, clk, 8 * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += double(h[i]);
        avg /= blocks;
        printf("ILP %d  active %2d  blocks %4d : %.2f clocks per FMA instruction\n", ILP, active, blocks, avg / (double(n) * 16 * ILP));
    }
    hipFree(out); hipFree(clk);
}

int main() {
    for (int active : {64, 32, 16, 1}) { run<1>(active); run<4>(active); }
    return 0;
}
