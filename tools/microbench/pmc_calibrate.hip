// Developer microbenchmark (GPU box): known byte counts in the access patterns of the fused step, to calibrate rocprofv3's
// FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section: only 16-B/lane streams are calibrated there).
//   cal_read_rows   every lane reads ROWS dwords of a [ROWS][N] structure of arrays (a wavefront reads 256 contiguous bytes per row)
//   cal_write_rows  every lane writes ROWS dwords of the same layout
//   cal_write_tile  a [N][W] row-major tile leaves as dwordx4 stores (the obs / states tile of the post phase), W = 113
//   cal_rw_rows     reads and rewrites the rows in place (the state rows of a step)
// hipcc --offload-arch=gfx950 -O3 -o pmc_calibrate pmc_calibrate.hip ; rocprofv3 --pmc FETCH_SIZE -- ./pmc_calibrate (WRITE_SIZE likewise)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ROWS 64
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void cal_read_rows(const float* __restrict__ a, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float s = 0.0f;
#pragma unroll 8
    for (int r = 0; r < ROWS; ++r) s += a[(size_t)r * n + i];
    if (s == 123.456f) out[i] = s;                      // never true: the kernel writes nothing
}
__global__ void cal_write_rows(float* __restrict__ a, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll 8
    for (int r = 0; r < ROWS; ++r) a[(size_t)r * n + i] = v + (float)r;
}
__global__ void cal_rw_rows(float* __restrict__ a, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float x[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) x[r] = a[(size_t)r * n + i];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) a[(size_t)r * n + i] = x[r] * 1.0001f;
}
__global__ void cal_write_tile(f4* __restrict__ t, int total4, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total4) t[i] = f4{v, v + 1.0f, v + 2.0f, v + 3.0f};
}
int main() {
    const int n = 65536, W = 113;
    float *a, *out; f4* t;
    if (hipMalloc(&a, (size_t)ROWS * n * 4) != hipSuccess || hipMalloc(&out, (size_t)n * 4) != hipSuccess || hipMalloc(&t, (size_t)n * W * 4) != hipSuccess) return 1;
    (void)hipMemset(a, 0, (size_t)ROWS * n * 4);
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(cal_read_rows, dim3(n / 256), dim3(256), 0, 0, a, out, n);
        hipLaunchKernelGGL(cal_write_rows, dim3(n / 256), dim3(256), 0, 0, a, n, (float)rep);
        hipLaunchKernelGGL(cal_rw_rows, dim3(n / 256), dim3(256), 0, 0, a, n);
        hipLaunchKernelGGL(cal_write_tile, dim3((n * W / 4 + 255) / 256), dim3(256), 0, 0, t, n * W / 4, (float)rep);
    }
    (void)hipDeviceSynchronize();
    printf("expected bytes per dispatch: cal_read_rows fetch %d  cal_write_rows write %d  cal_rw_rows fetch %d write %d  cal_write_tile write %d\n",
           ROWS * n * 4, ROWS * n * 4, ROWS * n * 4, ROWS * n * 4, n * W * 4);
    return 0;
}
